"""-m "not gpu": hand-derived known-answer cases that pin the oracle's reading of the .cu semantics
(the reference ships no vectors for these kernels), plus oracle-vs-reference-Python golden checks."""
import numpy as np
import torch

from oracle import pointset as orc


def test_fps_line_known_answer():
    # points on a line at x = 0,1,2,3,10: start 0 -> farthest 10 (idx 4) -> then 3 is 7 from 10 and 3 from 0: min 3;
    # 2: min(2,8)=2; 1: 1 -> pick idx 3 (x=3)?  min-dist: x=3 -> min(3,7)=3 ; so order 0,4,3
    xyz = torch.tensor([[[0., 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0]]])
    assert orc.furthest_point_sample(xyz, 3).tolist() == [[0, 4, 3]]


def test_fps_tie_rule_is_bit_reversed_thread_order():
    # 4 points (block size 4): after picking 0, points 1,2,3 are all at distance 1 from point 0 (unit axes).
    # The LDS tree compares (0,2),(1,3) then (0,1): ties keep the lower slot -> thread 2 beats nothing, slot 0
    # holds max(t0,t2)=t2 (d=1 > 0), slot 1 holds t1 (tie with t3 -> keeps t1); final (slot0=t2, slot1=t1) tie -> t2.
    xyz = torch.tensor([[[0., 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]])
    assert orc.furthest_point_sample(xyz, 2).tolist() == [[0, 2]]


def test_fps_m_greater_than_n_repeats_index_zero():
    xyz = torch.tensor([[[0., 0, 0], [5, 0, 0], [0, 7, 0]]])
    out = orc.furthest_point_sample(xyz, 6).tolist()[0]
    assert out[:3] == [0, 2, 1] and out[3:] == [0, 0, 0]  # all temps are 0 afterwards: argmax -> slot 0


def test_ball_query_padding_and_empty():
    xyz = torch.tensor([[[0., 0, 0], [0.1, 0, 0], [5, 0, 0], [0.2, 0, 0]]])
    centers = torch.tensor([[[0., 0, 0], [100, 0, 0]]])
    idx = orc.ball_query(1.0, 4, xyz, centers)
    assert idx[0, 0].tolist() == [0, 1, 3, 0]  # hits 0,1,3 in index order, remaining slot = first hit
    assert idx[0, 1].tolist() == [0, 0, 0, 0]  # empty ball keeps the pre-zeroed row
    assert orc.ball_query(1.0, 2, xyz, centers)[0, 0].tolist() == [0, 1]  # stops at nsample


def test_three_nn_strict_less_keeps_earlier_index_and_inf_when_short():
    known = torch.tensor([[[1., 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]]])  # all at distance 1 from the origin
    d, i = orc.three_nn(torch.zeros(1, 1, 3), known)
    assert i[0, 0].tolist() == [0, 1, 2] and d[0, 0].tolist() == [1.0, 1.0, 1.0]
    d, i = orc.three_nn(torch.zeros(1, 1, 3), known[:, :2])
    assert i[0, 0].tolist() == [0, 1, 0] and torch.isinf(d[0, 0, 2])  # 1e40 -> inf in fp32, index stays 0


def test_three_interpolate_and_group_known_answer():
    feats = torch.tensor([[[1., 2, 3, 4]]])
    idx = torch.tensor([[[0, 1, 3]]], dtype=torch.int32)
    w = torch.tensor([[[0.5, 0.25, 0.25]]])
    assert orc.three_interpolate(feats, idx, w).tolist() == [[[2.0]]]
    g = orc.grouping_operation(feats, torch.tensor([[[3, 0], [1, 1]]], dtype=torch.int32))
    assert g.tolist() == [[[[4.0, 1.0], [2.0, 2.0]]]]
    assert orc.gather_operation(feats, torch.tensor([[2, 2, 0]], dtype=torch.int32)).tolist() == [[[3.0, 3.0, 1.0]]]


def test_opt_n_threads_matches_reference_rule():
    assert [orc.opt_n_threads(n) for n in (1, 2, 3, 64, 100, 1000, 1024, 8192, 65536)] == [1, 2, 2, 64, 64, 512, 1024, 1024, 1024]


def test_knn_definition_lexicographic_ties():
    ref = torch.tensor([[[1., 0, 0], [1, 0, 0], [0, 0, 0], [1, 0, 0]]])  # three exact duplicates
    idx, dist = orc.knn(torch.zeros(1, 1, 3), ref, 3, mode=1, return_dist=True)
    assert idx[0, 0].tolist() == [2, 0, 1] and dist[0, 0].tolist() == [0.0, 1.0, 1.0]
    assert orc.knn(torch.zeros(1, 1, 3), ref[:, :2], 4, mode=0)[0, 0].tolist() == [0, 1, 1, 1]  # N < K: tail repeats


def test_oracle_against_reference_python_goldens(golden_dir):
    """square_distance bitwise, knn / cosine-knn sets vs the reference's topk (stored by make_golden.py)."""
    import os
    from tests import golden_inputs as gi
    g = np.load(os.path.join(golden_dir, "layers_n256.npz"))
    a = gi.layer_inputs()
    lib = orc.lib()
    import ctypes
    qa, ra = a["xyz_a"][0, :64].contiguous().numpy(), a["xyz_b"][0, :64].contiguous().numpy()
    mine = np.array([[lib.orc_pair_dist(ctypes.c_void_p(qa[i].ctypes.data), ctypes.c_void_p(ra[j].ctypes.data), 0) for j in range(64)]
                     for i in range(64)], dtype=np.float32)
    assert np.array_equal(mine, g["square_distance"][0])  # the expansion canon IS torch-CPU's evaluation order
    for name, k in (("knn_point_k32", 32), ("knn_point_k3", 3)):
        got = np.sort(orc.knn(a["xyz_a"], a["xyz_b"], k).numpy(), -1)
        assert (got != g[name]).any(-1).mean() <= 0.02
    got = np.sort(orc.knn_cosine(a["f64_a"], a["f64_b"], 16).numpy(), -1)
    assert (got != g["knn_cosine_k16"]).any(-1).mean() <= 0.05  # bmm accumulation order differs from the fma chain


def test_chamfer_matches_definition():
    x = torch.tensor([[[0., 0, 0], [1, 0, 0]]])
    y = torch.tensor([[[0., 0, 0]]])
    assert abs(orc.chamfer(x, y) - (0.5 + 0.0)) < 1e-12  # mean_x min = (0+1)/2, mean_y min = 0


def test_oracle_grouping_modules_against_reference_classes(golden_dir):
    """QueryAndGroup / GroupAll (pointnet2/pointnet2_utils.py:231-290): the fixture holds the outputs of the REFERENCE'S classes;
    here the same composition is rebuilt from the oracle's ball_query / grouping_operation and must agree exactly."""
    import os
    from tests import golden_inputs as gi
    g = np.load(os.path.join(golden_dir, "pointnet2_modules.npz"))
    mi = gi.module_inputs()
    for r, ns, key in ((0.5, 16, "qg_r0.5_n16"), (2.0, 8, "qg_r2.0_n8")):
        idx = orc.ball_query(r, ns, mi["xyz"], mi["new_xyz"])
        gx = orc.grouping_operation(mi["xyz"].transpose(1, 2).contiguous(), idx) - mi["new_xyz"].transpose(1, 2).unsqueeze(-1)
        want = torch.cat([gx, orc.grouping_operation(mi["features"], idx)], dim=1)
        assert np.array_equal(want.numpy(), g[key])
    # the centres are cloud points: every ball holds at least its centre, and r=0.5 leaves some balls short of nsample (padding)
    idx = orc.ball_query(0.5, 16, mi["xyz"], mi["new_xyz"])
    assert bool((idx[0, :, 1:] == idx[0, :, :1]).all(dim=1).any()) or bool((idx[0, :, -1] == idx[0, :, 0]).any())
