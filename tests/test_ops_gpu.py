"""-m gpu: parity of every HIP entry point (called through the C ABI via mocopci_amd.pointnet2_utils /
mocopci_amd.ops) against the CPU oracle on the same seeded inputs.  Bit-exact for indices and for
values produced by the shared floating-point canon; property checks at BASELINE.json's full sizes."""
import numpy as np
import pytest
import torch

from mocopci_amd import _lib, ops, pointnet2_utils as pu
from oracle import pointset as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cloud(seed, b, n, dup=0.05, extent=(40.0, 40.0, 3.0)):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(b, n, 3, generator=g) * 2 - 1) * torch.tensor(extent)
    nd = int(n * dup)
    if nd:
        src = torch.randint(0, n - nd, (nd,), generator=g)
        x[:, n - nd:] = x[:, src]
        x = x[:, torch.randperm(n, generator=g)]
    return x.contiguous()


def test_library_loaded_in_process():
    lib = _lib.load()
    assert lib.mcp_abi_version() == 1
    assert torch.cuda.is_available()


@pytest.mark.parametrize("b,n,m", [(2, 1024, 256), (2, 8192, 2048), (3, 2048, 512), (2, 512, 256), (2, 256, 64),
                                   (1, 1000, 333), (2, 100, 40), (2, 40, 17), (1, 3, 3), (1, 1024, 2048), (1, 16384, 512),
                                   (1, 20000, 64), (1, 5000, 700), (2, 1025, 100), (1, 12000, 300), (1, 16383, 200),
                                   (1, 14000, 150), (1, 40000, 300), (1, 70000, 40)])
def test_fps_bit_exact(b, n, m):
    xyz = cloud(100 + n, b, n)
    want = orc.furthest_point_sample(xyz, m)
    got = pu.furthest_point_sample(xyz.to(DEV), m)
    assert got.dtype == torch.int32 and got.shape == (b, m)
    assert torch.equal(got.cpu(), want)


def test_fps_ties_all_equal_points():
    # every point identical: all distances tie at 0 -> the reference tree picks index 0 forever
    xyz = torch.ones(1, 777, 3)
    got = pu.furthest_point_sample(xyz.to(DEV), 9).cpu()
    assert torch.equal(got, orc.furthest_point_sample(xyz, 9)) and int(got.abs().sum()) == 0


def test_fps_resumes_from_given_temp():
    # the wrapper's temp buffer is an input (sampling_gpu.cu:100-112 reads it): a pre-filled temp must be honoured and the
    # final running distances written back in the original point order, by the resident and the spatially pruned kernel
    from mocopci_amd import pointnet2_cuda as pc
    for n, m in ((700, 50), (6000, 400)):
        xyz = cloud(9 + n, 2, n)
        g = torch.Generator().manual_seed(n)
        t0 = torch.rand(2, n, generator=g) * 30.0
        want_idx = torch.zeros(2, m, dtype=torch.int32)
        want_t = t0.clone()
        orc.lib().orc_fps(orc._f(xyz), orc._f(want_t), orc._i(want_idx), 2, n, m)
        got_idx = torch.zeros(2, m, dtype=torch.int32, device=DEV)
        got_t = t0.to(DEV)
        pc.furthest_point_sampling_wrapper(2, n, m, xyz.to(DEV), got_t, got_idx)
        assert torch.equal(got_idx.cpu(), want_idx) and torch.equal(got_t.cpu(), want_t)


def test_fps_prefix_property_full_size():
    # BASELINE config 5 shape (streaming kernel): FPS(m) is a prefix of FPS(2m); first index is 0; no repeats
    xyz = cloud(5, 1, 65536, dup=0.0).to(DEV)
    a = pu.furthest_point_sample(xyz, 64).cpu()
    b = pu.furthest_point_sample(xyz, 128).cpu()
    assert torch.equal(a[0], b[0, :64]) and int(a[0, 0]) == 0 and len(set(b[0].tolist())) == 128


def test_gather_group_and_grads():
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(2, 19, 500, generator=g)
    idx = torch.randint(0, 500, (2, 77), generator=g, dtype=torch.int32)
    assert torch.equal(pu.gather_operation(feats.to(DEV), idx.to(DEV)).cpu(), orc.gather_operation(feats, idx))
    gidx = torch.randint(0, 500, (2, 60, 9), generator=g, dtype=torch.int32)
    assert torch.equal(pu.grouping_operation(feats.to(DEV), gidx.to(DEV)).cpu(), orc.grouping_operation(feats, gidx))
    # backward = scatter-add as a segmented reduction in ascending position order (round 4): the oracle's sequential loop, bit for bit
    f = feats.to(DEV).requires_grad_(True)
    out = pu.grouping_operation(f, gidx.to(DEV))
    go = torch.randn(out.shape, generator=g)
    out.backward(go.to(DEV))
    assert torch.equal(f.grad.cpu(), orc.grouping_operation_grad(go, gidx, 500))
    f2 = feats.to(DEV).requires_grad_(True)
    out = pu.gather_operation(f2, idx.to(DEV))
    go = torch.randn(out.shape, generator=g)
    out.backward(go.to(DEV))
    assert torch.equal(f2.grad.cpu(), orc.gather_operation_grad(go, idx, 500))


def test_reference_api_backward_scatters_are_deterministic_and_exact():
    """K3 / K6 / K9 behind the reference's autograd functions: heavy collisions (every destination hit ~250 times), five runs give the
    same bits, and those bits are the oracle's sequential loop (pointnet2_oracle.c: positions ascending)."""
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(2, 33, 64, generator=g)
    gidx = torch.randint(0, 64, (2, 500, 32), generator=g, dtype=torch.int32)
    go = torch.randn(2, 33, 500, 32, generator=g)
    runs = []
    for _ in range(5):
        f = feats.to(DEV).requires_grad_(True)
        pu.grouping_operation(f, gidx.to(DEV)).backward(go.to(DEV))
        runs.append(f.grad.clone())
    assert all(torch.equal(runs[0], r) for r in runs[1:])
    assert torch.equal(runs[0].cpu(), orc.grouping_operation_grad(go, gidx, 64))
    # K9: three_interpolate backward, 3 weighted addends per unknown point
    idx3 = torch.randint(0, 64, (2, 4000, 3), generator=g, dtype=torch.int32)
    w3 = torch.rand(2, 4000, 3, generator=g)
    go3 = torch.randn(2, 33, 4000, generator=g)
    runs = []
    for _ in range(3):
        f = feats.to(DEV).requires_grad_(True)
        pu.three_interpolate(f, idx3.to(DEV), w3.to(DEV)).backward(go3.to(DEV))
        runs.append(f.grad.clone())
    assert all(torch.equal(runs[0], r) for r in runs[1:])
    assert torch.equal(runs[0].cpu(), orc.three_interpolate_grad(go3, idx3, w3, 64))


@pytest.mark.parametrize("B,T,n,hot", [(2, 5000, 64, 0), (3, 40000, 8192, 0), (2, 70000, 300, 5), (1, 17, 1000, 0), (2, 262144, 8192, 3), (1, 1048576, 2, 0)])
def test_scatter_segments_is_the_stable_sort(B, T, n, hot):
    """mcp_scatter_segments (counting sort: count / scan / fill / per-row rank) against torch's stable sort + searchsorted: the same
    permutation and the same CSR offsets, with short rows (all-pairs ranks), rows of 25..64 (bitonic network), rows longer than 64
    (four registers per lane), rows beyond 256 -- `hot` destinations that a third of the positions point at: sorted chunk share by
    chunk share in LDS; two destinations for a million positions: shares beyond the LDS buffer -- and empty rows; two runs give the
    same bits."""
    from mocopci_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    idx = torch.randint(0, n, (B, T), generator=g, dtype=torch.int32)
    if hot:
        mask = torch.rand(B, T, generator=g) < 0.33
        idx[mask] = torch.randint(0, hot, (int(mask.sum()),), generator=g, dtype=torch.int32)
    if n >= 1000:
        idx[idx == 7] = 8          # an empty row
    d = idx.to(DEV)
    order, seg = ops._scatter_segments(d, n)
    order2, seg2 = ops._scatter_segments(d, n)
    assert torch.equal(order, order2) and torch.equal(seg, seg2)
    keys, want = torch.sort(idx.long(), dim=1, stable=True)
    bounds = torch.arange(n + 1).expand(B, n + 1).contiguous()
    want_seg = torch.searchsorted(keys.contiguous(), bounds)
    assert torch.equal(seg.cpu().long(), want_seg)
    assert torch.equal(order.cpu().long(), want)


def test_reference_api_grad_wrappers_add_into_the_callers_buffer():
    """The reference's backward kernels atomicAdd into grad_points (sampling_gpu.cu:46-83, group_points_gpu.cu:8-44,
    interpolate_gpu.cu:120-161): a caller that passes a buffer that already holds something gets base + scatter."""
    from mocopci_amd import pointnet2_cuda as pn2
    g = torch.Generator().manual_seed(19)
    B, C, N = 2, 5, 64
    base = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, 40), generator=g, dtype=torch.int32)
    go = torch.randn(B, C, 40, generator=g)
    buf = base.to(DEV)
    pn2.gather_points_grad_wrapper(B, C, N, 40, go.to(DEV), idx.to(DEV), buf)
    assert torch.equal(buf.cpu(), base + orc.gather_operation_grad(go, idx, N))
    gidx = torch.randint(0, N, (B, 40, 8), generator=g, dtype=torch.int32)
    go = torch.randn(B, C, 40, 8, generator=g)
    buf = base.to(DEV)
    pn2.group_points_grad_wrapper(B, C, N, 40, 8, go.to(DEV), gidx.to(DEV), buf)
    assert torch.equal(buf.cpu(), base + orc.grouping_operation_grad(go, gidx, N))
    idx3 = torch.randint(0, N, (B, 90, 3), generator=g, dtype=torch.int32)
    w3 = torch.rand(B, 90, 3, generator=g)
    go = torch.randn(B, C, 90, generator=g)
    buf = base.to(DEV)
    pn2.three_interpolate_grad_wrapper(B, C, 90, N, go.to(DEV), idx3.to(DEV), w3.to(DEV), buf)
    assert torch.equal(buf.cpu(), base + orc.three_interpolate_grad(go, idx3, w3, N))


@pytest.mark.parametrize("radius,nsample", [(0.5, 16), (1.0, 16), (2.0, 8), (4.0, 8), (1e-4, 4)])
def test_ball_query_bit_exact(radius, nsample):
    xyz = cloud(11, 2, 16384, extent=(50.0, 50.0, 4.0))  # config 4 shape
    new_xyz = xyz[:, :2048].contiguous() + 0.01
    want = orc.ball_query(radius, nsample, xyz, new_xyz)
    got = pu.ball_query(radius, nsample, xyz.to(DEV), new_xyz.to(DEV)).cpu()
    assert torch.equal(got, want)
    if radius == 1e-4:
        assert int(want.abs().sum()) == 0  # empty balls stay zero (pre-zeroed idx)


def test_three_nn_ties_on_both_paths():
    """Coordinates on a coarse grid (many equal distances, duplicate points): the pruned search of large clouds and the exhaustive
    kernel of small ones both give the reference's answer -- strict < insertion, the lower index wins (interpolate_gpu.cu:26-49)."""
    from mocopci_amd import pointnet2_cuda
    for n, m in ((4096, 2048), (1024, 2048), (4096, 1024)):
        unknown, known = (cloud(31, 2, n) * 2).round() / 2, (cloud(32, 2, m) * 2).round() / 2
        wd, wi = orc.three_nn(unknown, known)
        d2 = torch.empty(2, n, 3, device=DEV)
        i2 = torch.empty(2, n, 3, dtype=torch.int32, device=DEV)
        pointnet2_cuda.three_nn_wrapper(2, n, m, unknown.to(DEV), known.to(DEV), d2, i2)
        assert torch.equal(i2.cpu(), wi), (n, m)
        assert torch.equal(torch.sqrt(d2.cpu()), wd), (n, m)


@pytest.mark.parametrize("n,m", [(8192, 2048), (1000, 300), (50, 2), (7, 1)])
def test_three_nn_and_interpolate(n, m):
    unknown, known = cloud(21, 2, n), cloud(22, 2, m)
    wd, wi = orc.three_nn(unknown, known)
    gd, gi_ = pu.three_nn(unknown.to(DEV), known.to(DEV))
    assert torch.equal(gi_.cpu(), wi)
    # the kernel's squared distances are bit-exact (inf where m < 3, like the reference's 1e40); the Python
    # wrapper's torch.sqrt runs on the device and may differ from the host sqrt by an ulp
    from mocopci_amd import pointnet2_cuda
    d2 = torch.empty(2, n, 3, device=DEV)
    i2 = torch.empty(2, n, 3, dtype=torch.int32, device=DEV)
    pointnet2_cuda.three_nn_wrapper(2, n, m, unknown.to(DEV), known.to(DEV), d2, i2)
    assert torch.equal(torch.sqrt(d2.cpu()), wd) and torch.equal(i2.cpu(), wi)
    torch.testing.assert_close(gd.cpu(), wd, rtol=2e-7, atol=0)
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(2, 13, m, generator=g)
    w = torch.rand(2, n, 3, generator=g)
    got = pu.three_interpolate(feats.to(DEV), wi.to(DEV), w.to(DEV)).cpu()
    assert torch.equal(got, orc.three_interpolate(feats, wi, w))


@pytest.mark.parametrize("q,n,k,mode", [(8192, 8192, 32, 0), (2048, 8192, 32, 0), (2048, 2048, 16, 0), (2048, 2048, 16, 1),
                                        (8192, 2048, 3, 0), (512, 256, 32, 0), (256, 64, 32, 0), (300, 1000, 7, 1), (64, 20, 32, 0),
                                        (1000, 3000, 1, 1), (70, 5000, 16, 0)])
def test_knn_bit_exact(q, n, k, mode):
    b = 2
    query, ref = cloud(31 + q, b, q), cloud(32 + n, b, n)
    if q == n:
        ref = query  # self search with exact duplicates: the lexicographic tie rule decides
    wi, wd = orc.knn(query, ref, k, mode=mode, return_dist=True)
    gi_, gd = ops.backend().knn(query.to(DEV), ref.to(DEV), k, mode=mode, return_dist=True)
    assert torch.equal(gi_.cpu(), wi)
    assert torch.equal(gd.cpu(), wd)


def test_knn_full_size_properties():
    # config 2 / 5 shapes: ascending distances, self is the nearest (distance of a point to itself is the
    # smallest expansion value up to rounding), indices in range and unique per row
    for b, n, k in ((8, 8192, 32), (1, 65536, 32)):
        x = cloud(41, b, n, dup=0.0).to(DEV)
        q = x[:, :2048].contiguous()
        idx, dist = ops.backend().knn(q, x, k, mode=1, return_dist=True)
        assert bool((dist[..., 1:] >= dist[..., :-1]).all())
        assert torch.equal(idx[..., 0].cpu(), torch.arange(2048, dtype=torch.int32).expand(b, -1))
        assert int(idx.min()) >= 0 and int(idx.max()) < n
        s = torch.sort(idx.long(), -1)[0]
        assert bool((s[..., 1:] != s[..., :-1]).all())


@pytest.mark.parametrize("c", [3, 35, 64, 256])
def test_group_rows(c):
    g = torch.Generator().manual_seed(c)
    pts = torch.randn(2, 700, c, generator=g)
    idx = torch.randint(0, 700, (2, 300, 8), generator=g, dtype=torch.int32)
    got = ops.backend().group_rows(pts.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(got, orc.group_rows(pts, idx))


def test_group_rows_add_leaky_is_exact():
    g = torch.Generator().manual_seed(8)
    pts, ctr = torch.randn(2, 300, 24, generator=g), torch.randn(2, 70, 24, generator=g)
    idx = torch.randint(0, 300, (2, 70, 9), generator=g, dtype=torch.int32)
    want = torch.nn.functional.leaky_relu(orc.group_rows(pts, idx) + ctr.unsqueeze(2), 0.1)
    got = ops.backend().group_rows_add_leaky(pts.to(DEV), idx.to(DEV), ctr.to(DEV), 0.1).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,s,c", [(8192, 2048, 3), (2048, 512, 128), (512, 256, 256), (256, 64, 5)])
def test_interp3_bit_exact(n, s, c):
    dense, sparse = cloud(51, 2, n), cloud(52, 2, s)
    feat = torch.randn(2, s, c, generator=torch.Generator().manual_seed(9))
    want = orc.interp3(dense, sparse, feat)
    be = ops.backend()
    got = be.interp3(dense.to(DEV), sparse.to(DEV), feat.to(DEV)).cpu()
    assert torch.equal(got, want)
    idx3, w3 = be.interp3_search(dense.to(DEV), sparse.to(DEV))
    assert torch.equal(be.interp3_apply(feat.to(DEV), idx3, w3).cpu(), want)


def test_channel_last_backward_kernels_match_autograd():
    # group_rows / interp3_apply backward (atomic scatter-adds: summation order differs, hence the tolerance) against
    # torch autograd through plain indexing of the same forward
    be = ops.backend()
    g = torch.Generator().manual_seed(12)
    pts = torch.randn(2, 300, 20, generator=g).to(DEV).requires_grad_(True)
    idx = torch.randint(0, 300, (2, 70, 9), generator=g, dtype=torch.int32).to(DEV)
    go = torch.randn(2, 70, 9, 20, generator=g).to(DEV)
    out = be.group_rows(pts, idx)
    out.backward(go)
    ref = pts.detach().clone().requires_grad_(True)
    torch.stack([ref[b][idx[b].long()] for b in range(2)]).backward(go)
    assert torch.equal(out.detach(), torch.stack([ref[b][idx[b].long()] for b in range(2)]).detach())
    torch.testing.assert_close(pts.grad, ref.grad, rtol=1e-5, atol=1e-5)
    dense, sparse = cloud(61, 2, 500).to(DEV), cloud(62, 2, 120).to(DEV)
    feat = torch.randn(2, 120, 12, generator=g).to(DEV).requires_grad_(True)
    i3, w3 = be.interp3_search(dense, sparse)
    go = torch.randn(2, 500, 12, generator=g).to(DEV)
    be.interp3_apply(feat, i3, w3).backward(go)
    ref = feat.detach().clone().requires_grad_(True)
    blend = torch.stack([(ref[b][i3[b].long()] * w3[b].unsqueeze(-1)).sum(dim=1) for b in range(2)])
    blend.backward(go)
    torch.testing.assert_close(feat.grad, ref.grad, rtol=1e-5, atol=1e-5)


def test_edge_shapes_and_error_codes():
    be = ops.backend()
    x = cloud(71, 1, 10).to(DEV)
    # fewer references than K: the list is padded by repeating its last valid entry (documented in mcp_knn)
    i, d = be.knn(x, x[:, :5].contiguous(), 16, return_dist=True)
    wi, wd = orc.knn(x.cpu(), x[:, :5].cpu().contiguous(), 16, return_dist=True)
    assert torch.equal(i.cpu(), wi) and torch.equal(d.cpu(), wd)
    # a single query / a single reference
    assert torch.equal(be.knn(x[:, :1].contiguous(), x, 3).cpu(), orc.knn(x[:, :1].cpu().contiguous(), x.cpu(), 3))
    assert int(be.knn(x, x[:, :1].contiguous(), 1).abs().sum()) == 0
    # npoint == 1 and npoint == n
    assert pu.furthest_point_sample(x, 1).tolist() == [[0]]
    assert sorted(pu.furthest_point_sample(x, 10)[0].tolist()) == list(range(10))
    # unsupported shapes fail loudly, never fall back
    with pytest.raises(RuntimeError):
        be.knn(x, x, 40)
    with pytest.raises(RuntimeError):
        be.knn_cosine(torch.randn(1, 8, 48, device=DEV), torch.randn(1, 8, 48, device=DEV), 4)
    with pytest.raises((RuntimeError, ValueError, TypeError)):
        be.knn(x.cpu(), x.cpu(), 3)


def test_chamfer_matches_oracle():
    x, y = cloud(61, 2, 4096), cloud(62, 2, 3000)
    got = float(ops.backend().chamfer(x.to(DEV), y.to(DEV)))
    want = orc.chamfer(x, y)
    assert abs(got - want) <= 1e-6 * abs(want)  # fp32 mean vs double mean


def test_fusion_mlp_matches_unfused_oracle():
    """mcp_fusion (fp32 MFMA chain) vs the unfused torch-CPU restatement: fp32 tolerance 1e-5 relative
    on coordinates of magnitude ~40 (sum order differs, arithmetic is exact fp32 on both sides)."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(77)
    B, N = 2, 1500
    p1 = cloud(71, B, N)
    p2 = (p1 + 0.2 * torch.randn(B, N, 3, generator=g)).contiguous()
    idx = torch.cat([orc.knn(p1, p1, 32), orc.knn(p1, p2, 32)], -1)
    ws = []
    for co, ci in ((64, 4), (64, 64), (128, 64)):
        ws += [torch.randn(co, ci, generator=g) / ci ** 0.5, 0.1 * torch.randn(co, generator=g)]
    want = OracleBackend().fusion_mlp(p1, p2, idx, *ws)
    got = ops.backend().fusion_mlp(p1.to(DEV), p2.to(DEV), idx.to(DEV), *[w.to(DEV) for w in ws]).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("d,n1,n2", [(64, 2048, 2048), (128, 512, 512), (64, 300, 777), (256, 256, 256), (256, 37, 500)])
def test_cross_volume_matches_unfused_oracle(d, n1, n2):
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(d + n1)
    B = 2
    xyz1, xyz2 = cloud(81, B, n1), cloud(82, B, n2)
    p1, p2 = torch.randn(B, n1, d, generator=g), torch.randn(B, n2, d, generator=g)
    idx = torch.randint(0, n2, (B, n1, 32), generator=g, dtype=torch.int32)
    wpos, bpos = torch.randn(d, 3, generator=g) * 0.3, torch.randn(d, generator=g) * 0.1
    wmlp, bmlp = torch.randn(d, d, generator=g) / d ** 0.5, torch.randn(d, generator=g) * 0.1
    ob = OracleBackend()
    want = ob.cross_volume(xyz1, xyz2, p1, p2, idx, ob.cross_pack(wpos, bpos, wmlp, bmlp))
    be = ops.backend()
    packed = be.cross_pack(*[t.to(DEV) for t in (wpos, bpos, wmlp, bmlp)])
    got = be.cross_volume(*[t.to(DEV) for t in (xyz1, xyz2, p1, p2, idx)], packed).cpu()
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("q,n,c,k", [(2048, 2048, 64, 16), (512, 512, 128, 16), (256, 256, 256, 16), (100, 333, 64, 5),
                                     (130, 20, 128, 16)])
def test_knn_cosine_bit_exact(q, n, c, k):
    g = torch.Generator().manual_seed(c + q)
    qf, rf = torch.randn(2, q, c, generator=g), torch.randn(2, n, c, generator=g)
    if q == n:
        rf[:, 7] = rf[:, 3]  # duplicated reference rows: exact ties
    wi, wd = orc.knn_cosine(qf, rf, k, return_dist=True)
    gi_, gd = ops.backend().knn_cosine(qf.to(DEV), rf.to(DEV), k, return_dist=True)
    assert torch.equal(gi_.cpu(), wi)
    assert torch.equal(gd.cpu(), wd)


@pytest.mark.parametrize("n,s,d", [(8192, 8192, 32), (2048, 512, 128), (256, 64, 512), (500, 77, 5)])
def test_pointconv_agg_matches_unfused_oracle(n, s, d):
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(n + d)
    B = 2
    xyz = cloud(91, B, n)
    new_xyz = xyz[:, :s].contiguous()
    pts = torch.randn(B, n, d, generator=g)
    idx = orc.knn(new_xyz, xyz, 32)
    wn = [torch.randn(8, 3, generator=g) * 0.3, torch.randn(8, generator=g) * 0.1, torch.randn(8, 8, generator=g) * 0.3,
          torch.randn(8, generator=g) * 0.1, torch.randn(8, 8, generator=g) * 0.3, torch.randn(8, generator=g) * 0.1]
    want = OracleBackend().pointconv_agg(xyz, new_xyz, pts, idx, *wn)
    got = ops.backend().pointconv_agg(*[t.to(DEV) for t in (xyz, new_xyz, pts, idx, *wn)]).cpu()
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("n,s,d,B", [(8192, 8192, 32, 2), (8192, 2048, 64, 8), (3000, 2731, 64, 6), (700, 333, 32, 1), (515, 77, 64, 2)])
def test_pointconv_linear_matches_the_two_kernel_form_and_the_oracle(n, s, d, B):
    """mcp_pointconv_linear (grouping + WeightNet + aggregation + Linear + LeakyReLU in one launch, mocopci.py:1330-1342) against
    the unfused oracle restatement and against mcp_pointconv_agg + mcp_linear: BIT-IDENTICAL from 16384 centres up (where the
    model uses it); ragged point counts (partial last workgroup)."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(n + d)
    xyz = cloud(93, B, n)
    new_xyz = xyz[:, :s].contiguous()
    pts = torch.randn(B, n, d, generator=g)
    idx = orc.knn(new_xyz, xyz, 32)
    wn = [torch.randn(8, 3, generator=g) * 0.3, torch.randn(8, generator=g) * 0.1, torch.randn(8, 8, generator=g) * 0.3,
          torch.randn(8, generator=g) * 0.1, torch.randn(8, 8, generator=g) * 0.3, torch.randn(8, generator=g) * 0.1]
    w = torch.randn(d, (d + 3) * 8, generator=g) * ((d + 3) * 8) ** -0.5
    b = torch.randn(d, generator=g) * 0.1
    be = ops.backend()
    assert be.pointconv_linear_supported(d, d) and not be.pointconv_linear_supported(128, 128) and not be.pointconv_linear_supported(32, 64)
    assert be.pointconv_linear_supported(d, d, rows=B * s) == (B * s >= 16384)
    want = OracleBackend().pointconv_linear(xyz, new_xyz, pts, idx, *wn, w, b, 0.1)
    dev = [t.to(DEV) for t in (xyz, new_xyz, pts, idx, *wn, w, b)]
    got = be.pointconv_linear(*dev, 0.1)
    assert got.shape == (B, s, d)
    two = be.linear(be.pointconv_agg(*dev[:-2]), dev[-2], dev[-1], 0.1)
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-4)
    if B * s >= 16384:
        assert torch.equal(got, two)
    else:
        torch.testing.assert_close(got, two, rtol=1e-5, atol=2e-5)   # same split-bf16 arithmetic, K summed in four parts there
    assert torch.equal(got, be.pointconv_linear(*dev, 0.1, packed=be.pointconv_linear_pack(dev[-2], dev[-1])))  # deterministic


def test_compat_helpers_keep_reference_signatures():
    from mocopci_amd import compat
    xyz, new_xyz = cloud(95, 2, 700).to(DEV), cloud(96, 2, 300).to(DEV)
    idx = compat.knn_point(16, xyz, new_xyz)
    assert idx.dtype == torch.int64 and idx.shape == (2, 300, 16)
    assert torch.equal(idx.cpu().int(), orc.knn(new_xyz.cpu(), xyz.cpu(), 16))
    g = compat.index_points_group(xyz, idx)
    assert g.shape == (2, 300, 16, 3) and torch.equal(g.cpu(), orc.group_rows(xyz.cpu(), idx.cpu().int()))
    d, i, _ = compat.knn_points(new_xyz, xyz, K=4)
    wi, wd = orc.knn(new_xyz.cpu(), xyz.cpu(), 4, mode=1, return_dist=True)
    assert torch.equal(i.cpu().int(), wi) and torch.equal(d.cpu(), wd)


@pytest.mark.parametrize("q,n,k,mode,b", [(8192, 8192, 32, 0, 2), (2048, 8192, 32, 0, 2), (4096, 5000, 16, 1, 2), (1500, 16384, 32, 0, 1),
                                          (8192, 8192, 7, 1, 1), (1024, 65536, 32, 0, 1), (8192, 2048, 3, 0, 2), (2048, 2048, 16, 0, 3),
                                          (3000, 2500, 1, 1, 1), (2048, 2048, 4, 0, 1)])
def test_knn_pruned_equals_bruteforce_and_oracle(q, n, k, mode, b):
    """The Morton/box-pruned search must return exactly what the exhaustive scan returns."""
    be = ops.backend()
    ref = cloud(131 + n, b, n)
    query = ref if q == n else cloud(132 + q, b, q)
    rd, qd = ref.to(DEV), (None if q == n else query.to(DEV))
    qd = rd if qd is None else qd
    gi_, gd = be.knn(qd, rd, k, mode=mode, return_dist=True)           # pruned path (n >= 2048, q >= 1024)
    bi, bd = be.knn_bruteforce(qd, rd, k, mode=mode, return_dist=True)
    assert torch.equal(gi_, bi) and torch.equal(gd, bd)
    if n <= 16384:
        wi, wd = orc.knn(query, ref, k, mode=mode, return_dist=True)
        assert torch.equal(gi_.cpu(), wi) and torch.equal(gd.cpu(), wd)


@pytest.mark.parametrize("n,q,k,mode", [(10, 40, 32, 0), (31, 100, 32, 1), (33, 64, 32, 0), (70, 17, 32, 0), (200, 300, 32, 1), (4096, 4096, 20, 0),
                                        (4096, 4096, 24, 1), (2500, 1100, 17, 0), (5000, 4096, 31, 0)])
def test_knn_walk_kernel_edges_through_the_c_abi(n, q, k, mode):
    """The K = 32 walk kernel straight through mcp_build_cloud + mcp_knn_pruned at the shapes the Python layer never sends it: fewer
    references than list entries (the missing ranks repeat the last valid one, as the exhaustive kernel does), a partial last tile,
    query counts that leave lanes dead, and row widths that are not whole 16-byte pieces (k = 17, 20, 24, 31)."""
    be = ops.backend()
    ref = cloud(401 + n, 2, n).to(DEV)
    query = cloud(402 + q, 2, q).to(DEV)
    B = 2
    rs, rperm, rboxes = be._build_cloud(ref)
    qs, qperm, _ = be._build_cloud(query)
    idx = torch.empty((B, q, k), dtype=torch.int32, device=DEV)
    dist = torch.empty((B, q, k), dtype=torch.float32, device=DEV)
    ops._call("mcp_knn_pruned", query, B, q, n, k, mode, _lib.fptr(qs), _lib.iptr(qperm), _lib.fptr(rs), _lib.iptr(rperm), _lib.fptr(rboxes),
              _lib.iptr(idx), _lib.fptr(dist))
    bi, bd = be.knn_bruteforce(query, ref, k, mode=mode, return_dist=True)
    assert torch.equal(idx, bi) and torch.equal(dist, bd)
    wi, wd = orc.knn(query.cpu(), ref.cpu(), k, mode=mode, return_dist=True)
    assert torch.equal(idx.cpu(), wi) and torch.equal(dist.cpu(), wd)


def test_interp3_search_pruned_route_matches_oracle():
    # dense 8192 / sparse 2048: the 3-NN search takes the pruned kernel (K <= 4 list), weights come from mcp_interp3_weights
    be = ops.backend()
    dense, sparse = cloud(201, 2, 8192), cloud(202, 2, 2048)
    feat = torch.randn(2, 2048, 9, generator=torch.Generator().manual_seed(4))
    i3, w3 = be.interp3_search(dense.to(DEV), sparse.to(DEV))
    wi = orc.knn(dense, sparse, 3, mode=0)
    assert torch.equal(i3.cpu(), wi)
    got = be.interp3_apply(feat.to(DEV), i3, w3).cpu()
    torch.testing.assert_close(got, orc.interp3(dense, sparse, feat), rtol=1e-6, atol=1e-7)


def test_sampled_neighbours_are_rows_of_the_self_search():
    # model.sampled_neighbours: the K nearest of an FPS-sampled point in its own cloud = the self search's row (dup points included)
    from mocopci_amd.model import MoCoPCI
    be = ops.backend()
    x = cloud(301, 2, 8192).to(DEV)
    sel = pu.furthest_point_sample(x, 2048)
    sub = be.group_rows(x, sel)
    direct = be.knn(sub, x, 32)
    assert torch.equal(MoCoPCI.sampled_neighbours(be.knn(x, x, 32), sel), direct)
    assert torch.equal(direct.cpu(), orc.knn(sub.cpu(), x.cpu(), 32))


def test_knn_pruned_clustered_and_degenerate_clouds():
    be = ops.backend()
    g = torch.Generator().manual_seed(5)
    # two tight clusters far apart + a flat (z = const) cloud + all points identical
    a = torch.cat([torch.randn(1, 3000, 3, generator=g) * 0.01, torch.randn(1, 3000, 3, generator=g) * 0.01 + 500.0], 1)
    flat = torch.rand(1, 6000, 3, generator=g) * torch.tensor([100.0, 100.0, 0.0])
    same = torch.ones(1, 5000, 3) * 3.25
    for c in (a, flat, same):
        x = c.contiguous().to(DEV)
        i1, d1 = be.knn(x, x, 32, return_dist=True)
        i2, d2 = be.knn_bruteforce(x, x, 32, return_dist=True)
        assert torch.equal(i1, i2) and torch.equal(d1, d2)


@pytest.mark.parametrize("bf,nq,nk,heads,hd", [(4, 2048, 2048, 8, 8), (6, 512, 512, 8, 16), (2, 300, 777, 4, 8), (1, 33, 70, 2, 16),
                                               (3, 256, 256, 8, 32), (4, 256, 256, 3, 256), (2, 100, 333, 2, 64), (1, 40, 50, 1, 256)])
def test_attention_small_matches_fp32_reference(bf, nq, nk, heads, hd):
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(nq + hd)
    C = heads * hd
    q = torch.randn(bf, nq, C, generator=g)
    kv = torch.randn(bf, nk, 2 * C, generator=g)
    want = OracleBackend().attention(q.double(), kv.double(), heads).float()   # float64 reference
    got = ops.backend().attention(q.to(DEV), kv.to(DEV), heads).cpu()
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6)


def test_full_size_hash_pins():
    """Bit-exactness at BASELINE sizes against SHA-256 pins of the oracle's outputs (oracle/make_hashes.py)."""
    import hashlib, json, os
    from tests.golden_inputs import big_cloud
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hashes.json")))
    h = lambda t: hashlib.sha256(t.cpu().contiguous().numpy().tobytes()).hexdigest()
    be = ops.backend()
    for n, m in ((8192, 2048), (16384, 2048), (65536, 512)):
        assert h(pu.furthest_point_sample(big_cloud(n).to(DEV), m)) == pins[f"fps_{n}_{m}"], (n, m)
    for n, s in ((8192, 2048), (16384, 2048), (65536, 2048)):
        _, i = pu.three_nn(big_cloud(n).to(DEV), big_cloud(s, seed=2).to(DEV))
        assert h(i) == pins[f"three_nn_{n}_{s}"], (n, s)
    x = big_cloud(8192).to(DEV)
    assert h(be.knn(x, x, 32)) == pins["knn32_8192"]
    x = big_cloud(16384).to(DEV)
    assert h(be.knn(x, x, 16, mode=1)) == pins["knn16_direct_16384"]
    assert h(pu.ball_query(1.0, 16, x, x[:, :2048].contiguous())) == pins["ball_query_16384_r1_16"]


@pytest.mark.parametrize("n", [2048, 333])
def test_ptblock_attention_matches_unfused_oracle(n):
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(n)
    B = 2
    xyz = cloud(141, B, n)
    q, k, v = (torch.randn(B, n, 64, generator=g) for _ in range(3))
    idx = orc.knn(xyz, xyz, 16, mode=1)
    ws = [torch.randn(64, 3, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1]
    for _ in range(3):
        ws += [torch.randn(64, 64, generator=g) / 8, torch.randn(64, generator=g) * 0.1]
    ob = OracleBackend()
    want = ob.ptblock_attention(xyz, q, k, v, idx, ob.ptblock_pack(*ws))
    be = ops.backend()
    packed = be.ptblock_pack(*[w.to(DEV) for w in ws])
    got = be.ptblock_attention(*[t.to(DEV) for t in (xyz, q, k, v, idx)], packed).cpu()
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)
    # q, k, v read in place as slices of one packed (B,N,192) projection (row stride 192): identical result
    qkv = torch.cat([q, k, v], dim=-1).to(DEV)
    got2 = be.ptblock_attention(xyz.to(DEV), qkv[..., :64], qkv[..., 64:128], qkv[..., 128:], idx.to(DEV), packed).cpu()
    assert torch.equal(got, got2)


def test_grouping_modules_match_reference_classes(golden_dir):
    """QueryAndGroup / GroupAll of mocopci_amd.pointnet2_utils on the HIP kernels against the outputs of the REFERENCE'S own
    classes (pointnet2/pointnet2_utils.py:231-290; fixture from oracle/make_golden.py): bit-exact -- ball query indices, two
    gathers and one exact subtraction."""
    import os
    from tests import golden_inputs as gi
    g = np.load(os.path.join(golden_dir, "pointnet2_modules.npz"))
    mi = {k: v.to(DEV) for k, v in gi.module_inputs().items()}
    for r, ns, key in ((0.5, 16, "qg_r0.5_n16"), (2.0, 8, "qg_r2.0_n8")):
        got = pu.QueryAndGroup(r, ns)(mi["xyz"], mi["new_xyz"], mi["features"])
        assert got.shape == g[key].shape and np.array_equal(got.cpu().numpy(), g[key])
    assert np.array_equal(pu.QueryAndGroup(1.0, 8)(mi["xyz"], mi["new_xyz"], None).cpu().numpy(), g["qg_xyz_only"])
    assert np.array_equal(pu.QueryAndGroup(1.0, 8, use_xyz=False)(mi["xyz"], mi["new_xyz"], mi["features"]).cpu().numpy(), g["qg_no_xyz"])
    assert np.array_equal(pu.GroupAll()(mi["xyz"], None, mi["features"]).cpu().numpy(), g["group_all"])
    with pytest.raises(AssertionError):
        pu.QueryAndGroup(1.0, 8, use_xyz=False)(mi["xyz"], mi["new_xyz"], None)


@pytest.mark.parametrize("r,ns", [(0.7, 12), (1.5, 24), (3.0, 64), (0.05, 8)])
def test_query_and_group_one_launch_equals_the_composed_functions(r, ns):
    """mcp_query_and_group (inference: one launch) against the differentiable composition ball_query + grouping_operation x 2 +
    subtraction + cat of the same module, bit for bit -- any nsample up to 64 (not only powers of two), empty balls included
    (r = 0.05: most centres group point 0 only through the no-hit rule) -- and the composed form carries the gradient."""
    xyz = cloud(61, 2, 3000).to(DEV)
    new_xyz = (xyz[:, ::7] + 0.01).contiguous()
    feats = torch.randn(2, 10, 3000, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    mod = pu.QueryAndGroup(r, ns)
    fused = mod(xyz, new_xyz, feats)
    f2 = feats.clone().requires_grad_(True)
    composed = mod(xyz, new_xyz, f2)
    assert composed.requires_grad and fused.shape == (2, 13, new_xyz.shape[1], ns)
    assert torch.equal(fused, composed.detach())
    (g,) = torch.autograd.grad(composed.sum(), f2)
    assert g.shape == feats.shape and torch.isfinite(g).all() and float(g.sum()) == float(2 * new_xyz.shape[1] * ns * 10)
    assert torch.equal(pu.QueryAndGroup(r, ns)(xyz, new_xyz, None), mod(xyz.clone().requires_grad_(True), new_xyz, None).detach())
    assert torch.equal(pu.QueryAndGroup(r, ns, use_xyz=False)(xyz, new_xyz, feats), composed.detach()[:, 3:])


def test_models_common_aliases_match_oracle():
    """The `models.common` names models/layers.py:15,35,37,62,64,162,170 imports (the reference does not ship that module):
    fps, gather_points, ball_query, three_nn, three_interpolate, group_points, bound by compat.install()."""
    import sys
    from mocopci_amd import compat
    saved = {k: sys.modules.get(k) for k in ("pointnet2_cuda", "pointnet2.pointnet2_utils", "models.pointnet2.pointnet2_utils", "models.common")}
    try:
        sys.modules.pop("models.common", None)
        compat.install()
        common = sys.modules["models.common"]
        xyz = cloud(41, 2, 1500)
        feats = torch.randn(2, 12, 1500, generator=torch.Generator().manual_seed(5))
        x, f = xyz.to(DEV), feats.to(DEV)
        sel = common.fps(x, 200)
        want_sel = orc.furthest_point_sample(xyz, 200)
        assert torch.equal(sel.cpu(), want_sel)
        new_f = common.gather_points(f, sel)
        assert torch.equal(new_f.cpu(), orc.gather_operation(feats, want_sel))
        new_xyz = common.gather_points(x.transpose(1, 2).contiguous(), sel).transpose(1, 2).contiguous()
        bq = common.ball_query(1.5, 12, x, new_xyz)
        want_bq = orc.ball_query(1.5, 12, xyz, new_xyz.cpu())
        assert torch.equal(bq.cpu(), want_bq)
        assert torch.equal(common.group_points(f, bq).cpu(), orc.grouping_operation(feats, want_bq))
        d, i3 = common.three_nn(x, new_xyz)
        wd, wi = orc.three_nn(xyz, new_xyz.cpu())
        assert torch.equal(i3.cpu(), wi)
        torch.testing.assert_close(d.cpu(), wd, rtol=2e-7, atol=0)  # device sqrt vs host sqrt: an ulp (the squared distances are exact)
        w = torch.softmax(-d, dim=-1).contiguous()
        assert torch.equal(common.three_interpolate(new_f, i3, w).cpu(), orc.three_interpolate(new_f.cpu(), wi, w.cpu()))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_config5_full_shape_pins():
    """BASELINE configs[4] at its full shape -- synthetic N=65536 dense scan, batch 8 (the FPS + KNN roofline run): FPS
    65536 -> 2048 of all eight clouds (tiled kernel, caller-provided workspace) against the oracle's SHA pin; the K=32
    self search with Q = N = 65536 (pruned kernel): a 4096-query slice against the oracle's pin, the same slice against
    the exhaustive kernel, and -- for every query of every cloud -- ascending distances with the query itself first."""
    import hashlib, json, os
    from tests.golden_inputs import big_cloud
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hashes.json")))
    h = lambda t: hashlib.sha256(t.cpu().contiguous().numpy().tobytes()).hexdigest()
    be = ops.backend()
    x = big_cloud(65536, seed=5, batch=8).to(DEV)
    assert _lib.load().mcp_fps_workspace_bytes(8, 65536, 2048) == 8 * 65536 * 20
    sel = pu.furthest_point_sample(x, 2048)
    assert h(sel) == pins["fps_c5_8x65536_2048"]
    # the reference wrapper's exact argument list (no workspace) takes the streaming kernel: same indices (two clouds: it is slow)
    t = torch.full((2, 65536), 1e10, device=DEV)
    o = torch.zeros((2, 2048), dtype=torch.int32, device=DEV)
    _lib.check(_lib.load().mcp_furthest_point_sampling(2, 65536, 2048, x.data_ptr(), t.data_ptr(), o.data_ptr(), _lib.stream()))
    assert torch.equal(o, sel[:2])
    idx, dist = be.knn(x, x, 32, return_dist=True)
    assert h(idx[:1, 30000:34096]) == pins["knn32_c5_65536_rows_30000_34096"]
    assert torch.equal(idx[:1, 30000:34096], be.knn_bruteforce(x[:1, 30000:34096].contiguous(), x[:1], 32))
    assert bool((dist[..., 1:] >= dist[..., :-1]).all())
    # every point finds itself among its 32 nearest (the expansion form's self distance is only ~0 up to cancellation, so it
    # need not be FIRST; the 32nd neighbour is ~1.6 m away at this density)
    me = torch.arange(65536, device=DEV, dtype=torch.int32).view(1, -1, 1)
    assert bool((idx == me).any(dim=-1).all())


@pytest.mark.parametrize("n,s,c", [(300, 100, 7), (1500, 700, 64), (4096, 2048, 16)])
def test_interp3_gradient_flows_at_every_size(n, s, c):
    """UpsampleFlow (mocopci.py:1485-1502) is differentiable w.r.t. the sparse features whichever kernel route the sizes
    select (fused small-level kernel / pruned search + blend): gradient against torch autograd of the same blend."""
    dense, sparse = cloud(n, 2, n).to(DEV), cloud(s + 1, 2, s).to(DEV)
    feat = torch.randn(2, s, c, device=DEV, generator=torch.Generator(device=DEV).manual_seed(n), requires_grad=True)
    be = ops.backend()
    out = be.interp3(dense, sparse, feat)
    assert out.requires_grad
    g = torch.randn_like(out)
    (grad,) = torch.autograd.grad(out, feat, g)
    with torch.no_grad():
        assert torch.equal(out, be.interp3(dense, sparse, feat.detach()))  # same values as the inference route
        idx3, w3 = be.interp3_search(dense, sparse)
    f2 = feat.detach().clone().requires_grad_(True)
    rows = torch.gather(f2.unsqueeze(1).expand(-1, n, -1, -1), 2, idx3.long().unsqueeze(-1).expand(-1, -1, -1, c))
    ref = (rows * w3.unsqueeze(-1)).sum(2)
    (want,) = torch.autograd.grad(ref, f2, g)
    torch.testing.assert_close(grad, want, rtol=1e-5, atol=1e-5)


def test_search_on_a_reused_buffer_sees_the_new_contents():
    """a buffer written through the library's raw-pointer entry points and then searched as a cloud must never be
    answered from a stale sorted copy.  Outside model.forward's cloud_scope nothing is cached."""
    a, b = cloud(71, 1, 4096), cloud(72, 1, 4096)
    buf = a.to(DEV).clone()
    be = ops.backend()
    assert torch.equal(be.knn(buf, buf, 16).cpu(), orc.knn(a, a, 16))
    src = b.to(DEV)
    ar = torch.arange(4096, dtype=torch.int32, device=DEV).unsqueeze(0).contiguous()
    ops._call("mcp_group_rows", src, 1, 4096, 3, 4096, src.data_ptr(), ar.data_ptr(), buf.data_ptr())  # raw write, no version bump
    assert torch.equal(buf.cpu(), b)
    assert torch.equal(be.knn(buf, buf, 16).cpu(), orc.knn(b, b, 16))


@pytest.mark.parametrize("cin,hidden,cout,rows,with_res", [(64, 256, 64, 5000, True), (64, 256, 3, 4096, False), (128, 512, 3, 1000, False),
                                                           (128, 512, 32, 3000, True), (64, 256, 64, 31, True)])
def test_mlp2_matches_unfused_oracle(cin, hidden, cout, rows, with_res):
    """Fused two-layer MLP (Mlp_T / flow heads, mocopci.py:1558-1565, :566-567, :510-511) against the oracle backend's unfused
    Linear - PReLU - Linear (+ residual); x read through a row stride (a column slice of a wider tensor) as the model does."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(cin + hidden + cout)
    wide = torch.randn(2, rows, cin + 8, generator=g)
    x = wide[..., :cin]
    res = torch.randn(2, rows, cout, generator=g) if with_res else None
    w1, b1 = torch.randn(hidden, cin, generator=g) / cin ** 0.5, torch.randn(hidden, generator=g) * 0.1
    w2, b2 = torch.randn(cout, hidden, generator=g) / hidden ** 0.5, torch.randn(cout, generator=g) * 0.1
    want = OracleBackend().mlp2(x, w1, b1, w2, b2, 0.25, res=res)
    be = ops.backend()
    dw = [t.to(DEV) for t in (w1, b1, w2, b2)]
    got = be.mlp2(wide.to(DEV)[..., :cin], *dw, 0.25, res=None if res is None else res.to(DEV))
    assert got.shape == want.shape
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5)
    got2 = be.mlp2(x.contiguous().to(DEV), *dw, 0.25, res=None if res is None else res.to(DEV), packed=be.mlp2_pack(*dw))
    assert torch.equal(got, got2)


@pytest.mark.parametrize("rows,ks,n,slope,with_res", [(20000, (64,), 64, 0.1, False), (16390, (280,), 32, 0.1, False), (17000, (64, 64, 64), 64, 1.0, True),
                                                      (16384, (32,), 192, 0.0, False), (33000, (128,), 100, 0.25, True),
                                                      (16500, (536,), 64, 0.1, False), (16384, (300, 200, 36), 40, 1.0, True)])
def test_linear_matches_unfused_oracle(rows, ks, n, slope, with_res):
    """Fused per-point Linear (Conv1d wrapper mocopci.py:1111-1127 + the concatenation in front of it) against the oracle backend's
    cat - Linear - activation - residual; one piece is read through a row stride (a column slice of a wider tensor)."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(rows + n)
    wide = torch.randn(rows, ks[0] + 12, generator=g)
    xs = [wide[:, 4:4 + ks[0]]] + [torch.randn(rows, k, generator=g) for k in ks[1:]]
    w, b = torch.randn(n, sum(ks), generator=g) / sum(ks) ** 0.5, torch.randn(n, generator=g) * 0.1
    res = torch.randn(rows, n, generator=g) if with_res else None
    want = OracleBackend().linear(xs, w, b, slope, res)
    be = ops.backend()
    wd = wide.to(DEV)
    xd = [wd[:, 4:4 + ks[0]]] + [t.to(DEV) for t in xs[1:]]
    assert be.linear_supported(xd, n)
    got = be.linear(xd if len(xd) > 1 else xd[0], w.to(DEV), b.to(DEV), slope, None if res is None else res.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5)
    # shapes outside the policy are declined (the caller then takes the BLAS chain)
    assert not be.linear_supported(xd[0][:1000], n) and not be.linear_supported(torch.empty(20000, 700, device=DEV), 64)
    assert not be.linear_supported(torch.empty(20000, 600, device=DEV), 128)


def test_cross_volume_batch_map_equals_replicated_inputs():
    """Replicated / selected batches (the three flow iterations of Multiframe_Attention share their features, mocopci.py:191-197)
    are read through a batch map instead of being copied: same bits as the call on the materialised copies."""
    g = torch.Generator().manual_seed(77)
    B0, n, d = 4, 300, 64
    members = [0, 1, 2, 3, 0, 2, 3, 1, 1]                       # a 9-element batch drawn from 4 sources
    p1, p2 = torch.randn(B0, n, d, generator=g), torch.randn(B0, n, d, generator=g)
    idx_c = torch.randint(0, n, (B0, n, 16), generator=g, dtype=torch.int32)
    xyz1, xyz2 = cloud(83, len(members), n), cloud(84, len(members), n)
    idx_p = torch.randint(0, n, (len(members), n, 16), generator=g, dtype=torch.int32)
    w = [torch.randn(d, 3, generator=g) * 0.3, torch.randn(d, generator=g) * 0.1, torch.randn(d, d, generator=g) / d ** 0.5, torch.randn(d, generator=g) * 0.1]
    be = ops.backend()
    packed = be.cross_pack(*[t.to(DEV) for t in w])
    m = torch.tensor(members)
    want = be.cross_volume(xyz1.to(DEV), xyz2.to(DEV), p1[m].contiguous().to(DEV), p2[m].contiguous().to(DEV),
                           (idx_c[m].contiguous().to(DEV), idx_p.to(DEV)), packed)
    got = be.cross_volume(xyz1.to(DEV), xyz2.to(DEV), p1.to(DEV), p2.to(DEV), (idx_c.to(DEV), idx_p.to(DEV)), packed,
                          bmap=m.int().to(DEV), shared=7)
    assert torch.equal(got, want)
    got4 = be.cross_volume(xyz1.to(DEV), xyz2.to(DEV), p1[m].contiguous().to(DEV), p2[m].contiguous().to(DEV), (idx_c.to(DEV), idx_p.to(DEV)),
                           packed, bmap=m.int().to(DEV), shared=4)
    assert torch.equal(got4, want)


@pytest.mark.parametrize("rows,k,n", [(12288, 1024, 3), (4099, 256, 4), (2048, 512, 1)])
def test_linear_narrow_matches_torch(rows, k, n):
    """PReLU followed by a (K -> n <= 4) Linear as one streaming kernel (the folded flow tail of Mlp_T, mocopci.py:1561-1565 with
    :566-567): against the oracle backend's two-step form; K-long fp32 sums in a different order, hence the tolerance."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(rows + k)
    wide = torch.randn(rows, k + 8, generator=g)
    x = wide[:, 4:4 + k]                                            # a column slice: rows 16-byte aligned, stride k + 8
    w, b = torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g) * 0.1
    want = OracleBackend().linear_narrow(x, w, b, 0.25)
    be = ops.backend()
    assert be.linear_narrow_supported(rows, k, n) and not be.linear_narrow_supported(rows, 300, n) and not be.linear_narrow_supported(rows, k, 5)
    got = be.linear_narrow(wide.to(DEV)[:, 4:4 + k], w.to(DEV), b.to(DEV), 0.25)
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5)
    got2 = be.linear_narrow(x.contiguous().to(DEV).reshape(1, rows, k), w.to(DEV), None, 0.0)
    torch.testing.assert_close(got2.cpu()[0], torch.nn.functional.linear(torch.relu(x), w), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("rows,k,n", [(4100, 256, 1536), (8192, 96, 500)])
def test_linear_wide_outputs_run_as_column_blocks(rows, k, n):
    """Outputs wider than 256 columns run as column blocks of 128 over grid.y (n <= 2048, ceil(n/32) a multiple of 4); the shape
    policy leaves these to the library (only 10-15 % faster), the entry point takes them."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(rows + n)
    x = torch.randn(rows, k, generator=g)
    w, b = torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g) * 0.1
    want = OracleBackend().linear(x, w, b, 0.25, None)
    be = ops.backend()
    assert not be.linear_supported(x.to(DEV), n)
    got = be.linear(x.to(DEV), w.to(DEV), b.to(DEV), 0.25, None)
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("bf,n,heads,hd,shift", [(6, 300, 8, 8, 3), (4, 200, 8, 16, 2), (4, 256, 8, 32, 2), (4, 96, 3, 256, 2), (5, 130, 2, 64, 1)])
def test_attention_with_rotated_key_value_batch_reads_packed_projections_in_place(bf, n, heads, hd, shift):
    """mcp_attention: q / k / v are column slices of ONE packed projection (row stride 3C), batch element b of the queries attends
    to keys / values of element (b + shift) mod BF -- what the folded EI cross-former and cross_block3 launch.  Against the
    float64 dense formulation on the rolled tensors."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(bf * n + hd)
    C = heads * hd
    y = torch.randn(bf, n, 3 * C, generator=g)
    want = OracleBackend().attention_rot(y[..., :C].double(), y[..., C:2 * C].double(), y[..., 2 * C:].double(), heads, shift).float()
    yd = y.to(DEV)
    got = ops.backend().attention_rot(yd[..., :C], yd[..., C:2 * C], yd[..., 2 * C:], heads, shift).cpu()
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6)
    # shift 0 on the same views = the plain entry points on contiguous copies
    plain = ops.backend().attention(yd[..., :C].contiguous(), yd[..., C:].contiguous(), heads).cpu()
    assert torch.equal(ops.backend().attention_rot(yd[..., :C], yd[..., C:2 * C], yd[..., 2 * C:], heads, 0).cpu(), plain)
    with pytest.raises(RuntimeError):
        ops.backend().attention_rot(yd[..., :C], yd[..., C:2 * C], yd[..., 2 * C:], heads, bf)   # shift out of range


@pytest.mark.parametrize("rows,c", [(1000, 64), (513, 128), (2048, 256), (7, 24), (300, 1000)])
def test_add_layernorm_matches_torch(rows, c):
    g = torch.Generator().manual_seed(rows + c)
    x, y, b = torch.randn(rows, c, generator=g) * 3 + 1, torch.randn(rows, c, generator=g), torch.randn(c, generator=g)
    be = ops.backend()
    want = torch.nn.functional.layer_norm((x + y + b).double(), (c,), None, None, 1e-6).float()
    torch.testing.assert_close(be.add_layernorm(x.to(DEV), y.to(DEV), b.to(DEV)).cpu(), want, rtol=2e-5, atol=2e-6)
    want = torch.nn.functional.layer_norm(x.double(), (c,), None, None, 1e-6).float()
    torch.testing.assert_close(be.add_layernorm(x.to(DEV)).cpu(), want, rtol=2e-5, atol=2e-6)
    # a row-strided view (second half of a stacked batch) is read in place
    xs = torch.randn(2, rows, c, generator=g).to(DEV)
    torch.testing.assert_close(be.add_layernorm(xs[1]).cpu(), torch.nn.functional.layer_norm(xs[1].cpu(), (c,), None, None, 1e-6), rtol=2e-5, atol=2e-6)


def test_mfa_prepare_matches_torch():
    """The three inputs of Multi_Frame_Att from a subset of computed members in one launch, against plain torch indexing."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(11)
    M, N, C, R = 40, 300, 64, 24
    fea = torch.randn(M, N, C, generator=g)
    ss = torch.randint(0, M, (R,), generator=g, dtype=torch.int32)
    sp = torch.randint(0, M, (R,), generator=g, dtype=torch.int32)
    ts, tp, sc, sh = (torch.randn(R, C, generator=g), torch.randn(R, C, generator=g), torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g))
    want = OracleBackend().mfa_prepare(fea, ss, sp, ts, tp, sc, sh)
    got = ops.backend().mfa_prepare(*[t.to(DEV) for t in (fea, ss, sp, ts, tp, sc, sh)])
    assert torch.equal(got[0].cpu(), want[0])
    for a, b in zip(got[1:], want[1:]):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("b,n,m", [(3, 8192, 2048), (2, 2048, 512), (2, 512, 256), (2, 256, 64), (1, 50, 7), (2, 20000, 300), (1, 70000, 64)])
def test_fps_fresh_start_with_points_matches_the_two_step_form(b, n, m):
    """mcp_furthest_point_sampling_fresh: no temp buffer, optional sampled coordinates from the same launch -- the same indices as
    the reference-signature entry point with temp = 1e10 (bit-exact, every kernel variant: resident, spatial, tiled, and the
    streaming fallback beyond 65536 points) and exactly the rows a gather of them returns."""
    xyz = cloud(7 + n, b, n).to(DEV)
    sel, pts = ops.backend().fps(xyz, m, with_points=True)
    assert torch.equal(sel, pu.furthest_point_sample(xyz, m))
    assert torch.equal(pts, ops.backend().group_rows(xyz, sel))
    assert torch.equal(sel.cpu(), orc.furthest_point_sample(xyz.cpu(), m))


@pytest.mark.parametrize("rows,ks,n,slope,with_res", [(4096, [1048], 256, 0.1, False), (8192, [536], 128, 0.1, False), (2048, [512], 256, 1.0, True),
                                                      (4100, [64, 64, 64], 96, 0.1, True), (3000, [256], 32, 0.0, False), (5000, [128, 32], 192, 0.25, True)])
def test_linear_few_rows_split_k_matches_unfused_oracle(rows, ks, n, slope, with_res):
    """Below 16384 rows mcp_linear runs the split-K kernel (four waves share a 32-row tile and split its K chunks; fixed-order sum of
    the partial accumulators): pieces, a row-strided piece, residual, odd tile counts (NT = 1 path) and ragged row counts, against
    the oracle backend's cat - Linear - activation - residual; bit-reproducible from call to call."""
    from oracle.backend import OracleBackend
    g = torch.Generator().manual_seed(rows + n)
    wide = torch.randn(rows, ks[0] + 12, generator=g)
    xs = [wide[:, 4:4 + ks[0]]] + [torch.randn(rows, k, generator=g) for k in ks[1:]]
    w, b = torch.randn(n, sum(ks), generator=g) / sum(ks) ** 0.5, torch.randn(n, generator=g) * 0.1
    res = torch.randn(rows, n, generator=g) if with_res else None
    want = OracleBackend().linear(xs, w, b, slope, res)
    be = ops.backend()
    wd = wide.to(DEV)
    xd = [wd[:, 4:4 + ks[0]]] + [t.to(DEV) for t in xs[1:]]
    assert be.linear_supported(xd, n)
    call = lambda: be.linear(xd if len(xd) > 1 else xd[0], w.to(DEV), b.to(DEV), slope, None if res is None else res.to(DEV))
    got = call()
    torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-5)
    assert torch.equal(call(), got)


def test_furthest_point_samples_are_prefixes_of_longer_samples():
    """What MoCoPCI.forward(train=True) relies on for the ground-truth pyramid (mocopci.py:1099-1104): the N/16 and N/32 samples of
    a cloud are the first points of its N/4 sample, and a cloud's sample does not depend on the batch it is sampled in."""
    import torch
    from mocopci_amd import ops
    be = ops.backend()
    g = torch.Generator().manual_seed(77)
    clouds = ((torch.rand(5, 8192, 3, generator=g) * 2 - 1) * 20).cuda()
    long = be.fps(clouds, 2048)
    for m in (512, 256):
        assert torch.equal(be.fps(clouds, m), long[:, :m])
    assert torch.equal(be.fps(clouds[1:3].contiguous(), 2048), long[1:3])
