"""oracle/pointset.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end for oracle/libpointset_oracle.so (the C restatement of the
reference's point-set operators, see pointset_oracle.c).  Works on CPU torch
tensors; signatures mirror the reference's public operator names
(pointnet2/pointnet2_utils.py) so parity tests read like the reference's call
sites.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpointset_oracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pointset_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_chamfer.restype = ctypes.c_double
        _lib.orc_pair_dist.restype = ctypes.c_float
    return _lib


class _Ptr:
    """Pointer argument that keeps its tensor alive for the duration of the call: `_f(x.contiguous())` may be handed a
    temporary, and a bare c_void_p would let it be freed before the C function reads it."""

    def __init__(self, t):
        self.t = t
        self._as_parameter_ = ctypes.c_void_p(t.data_ptr())


def _f(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return _Ptr(t)


def _i(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu"
    return _Ptr(t)


def opt_n_threads(n):
    return lib().orc_opt_n_threads(int(n))


def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) f32 -> (B,npoint) i32.  pointnet2_utils.py:10-29."""
    B, N, _ = xyz.shape
    xyz = xyz.contiguous()
    out = torch.zeros(B, npoint, dtype=torch.int32)
    temp = torch.full((B, N), 1e10, dtype=torch.float32)
    lib().orc_fps(_f(xyz), _f(temp), _i(out), B, N, npoint)
    return out


def gather_operation(features, idx):
    """features (B,C,N), idx (B,M) i32 -> (B,C,M).  pointnet2_utils.py:39-60."""
    B, C, N = features.shape
    M = idx.shape[1]
    out = torch.empty(B, C, M, dtype=torch.float32)
    lib().orc_gather(_f(features.contiguous()), _i(idx.contiguous()), _f(out), B, C, N, M)
    return out


def gather_operation_grad(grad_out, idx, N):
    B, C, M = grad_out.shape
    g = torch.zeros(B, C, N, dtype=torch.float32)
    lib().orc_gather_grad(_f(grad_out.contiguous()), _i(idx.contiguous()), _f(g), B, C, N, M)
    return g


def grouping_operation(features, idx):
    """features (B,C,N), idx (B,S,K) i32 -> (B,C,S,K).  pointnet2_utils.py:156-178."""
    B, C, N = features.shape
    _, S, K = idx.shape
    out = torch.empty(B, C, S, K, dtype=torch.float32)
    lib().orc_group(_f(features.contiguous()), _i(idx.contiguous()), _f(out), B, C, N, S, K)
    return out


def grouping_operation_grad(grad_out, idx, N):
    B, C, S, K = grad_out.shape
    g = torch.zeros(B, C, N, dtype=torch.float32)
    lib().orc_group_grad(_f(grad_out.contiguous()), _i(idx.contiguous()), _f(g), B, C, N, S, K)
    return g


def ball_query(radius, nsample, xyz, new_xyz):
    """pointnet2_utils.py:200-220: xyz (B,N,3), new_xyz (B,M,3) -> (B,M,nsample) i32."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=torch.int32)
    lib().orc_ball_query(_f(new_xyz.contiguous()), _f(xyz.contiguous()), _i(idx), B, N, M, ctypes.c_float(radius), nsample)
    return idx


def three_nn(unknown, known):
    """pointnet2_utils.py:76-98: returns (sqrt(dist2), idx)."""
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = torch.empty(B, n, 3, dtype=torch.float32)
    idx = torch.empty(B, n, 3, dtype=torch.int32)
    lib().orc_three_nn(_f(unknown.contiguous()), _f(known.contiguous()), _f(d2), _i(idx), B, n, m)
    return torch.sqrt(d2), idx


def three_interpolate(features, idx, weight):
    """pointnet2_utils.py:108-131: features (B,C,M), idx/weight (B,n,3) -> (B,C,n)."""
    B, C, M = features.shape
    n = idx.shape[1]
    out = torch.empty(B, C, n, dtype=torch.float32)
    lib().orc_three_interpolate(_f(features.contiguous()), _i(idx.contiguous()), _f(weight.contiguous()), _f(out), B, C, M, n)
    return out


def three_interpolate_grad(grad_out, idx, weight, M):
    B, C, n = grad_out.shape
    g = torch.zeros(B, C, M, dtype=torch.float32)
    lib().orc_three_interpolate_grad(_f(grad_out.contiguous()), _i(idx.contiguous()), _f(weight.contiguous()), _f(g), B, C, n, M)
    return g


def knn(query, ref, k, mode=0, return_dist=False):
    """Lexicographic (d, index) K nearest, ascending.  mode 0 = reference square_distance
    expansion (mocopci.py:1130-1169), mode 1 = direct differences (pytorch3d knn_points)."""
    B, Q, _ = query.shape
    N = ref.shape[1]
    idx = torch.empty(B, Q, k, dtype=torch.int32)
    dist = torch.empty(B, Q, k, dtype=torch.float32)
    rc = lib().orc_knn(_f(query.contiguous()), _f(ref.contiguous()), _i(idx), _f(dist), B, Q, N, k, mode)
    assert rc == 0
    return (idx, dist) if return_dist else idx


def knn_cosine(qfeat, rfeat, k, return_dist=False):
    """Feature cosine KNN on channel-last feats (pointconv_util.py:111-153)."""
    B, Q, C = qfeat.shape
    N = rfeat.shape[1]
    idx = torch.empty(B, Q, k, dtype=torch.int32)
    dist = torch.empty(B, Q, k, dtype=torch.float32)
    rc = lib().orc_knn_cosine(_f(qfeat.contiguous()), _f(rfeat.contiguous()), _i(idx), _f(dist), B, Q, N, C, k)
    assert rc == 0
    return (idx, dist) if return_dist else idx


def group_rows(points, idx):
    """points (B,N,C) channel-last, idx (B,...) i32 -> (B,...,C)."""
    B, N, C = points.shape
    idx = idx.contiguous()
    per = idx[0].numel()
    out = torch.empty(*idx.shape, C, dtype=torch.float32)
    lib().orc_group_rows(_f(points.contiguous()), _i(idx), _f(out), B, N, C, per)
    return out


def interp3(dense_xyz, sparse_xyz, sparse_feat, idx3=None):
    """UpsampleFlow (mocopci.py:1485-1502) on channel-last tensors:
    dense (B,N,3), sparse (B,S,3), feat (B,S,C) -> (B,N,C)."""
    B, N, _ = dense_xyz.shape
    S = sparse_xyz.shape[1]
    C = sparse_feat.shape[2]
    if idx3 is None:
        idx3 = knn(dense_xyz, sparse_xyz, 3, mode=0)
    w = torch.empty(B, N, 3, dtype=torch.float32)
    lib().orc_interp3_weights(_f(dense_xyz.contiguous()), _f(sparse_xyz.contiguous()), _i(idx3.contiguous()), _f(w), B, N, S)
    out = torch.empty(B, N, C, dtype=torch.float32)
    lib().orc_interp3_apply(_f(sparse_feat.contiguous()), _i(idx3.contiguous()), _f(w), _f(out), B, N, S, C)
    return out


def chamfer(x, y):
    """x (B,N,3), y (B,M,3) -> python float (models/utils.py:36-45 semantics)."""
    B, N, _ = x.shape
    M = y.shape[1]
    return float(lib().orc_chamfer(_f(x.contiguous()), _f(y.contiguous()), B, N, M))


def earth_mover_distance(xyz1, xyz2, return_match=False):
    """models/EMD/emd.py:26-45 with transpose=False: xyz1 (B,N,3), xyz2 (B,M,3) -> cost (B)."""
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    match = torch.empty(B, M, N, dtype=torch.float32)
    cost = torch.empty(B, dtype=torch.float32)
    lib().orc_emd(_f(xyz1.contiguous()), _f(xyz2.contiguous()), _f(match), _f(cost), B, N, M)
    return (cost, match) if return_match else cost


def EMD(pc1, pc2):
    """models/utils.py:223-235: pc (B,3,N) -> mean(cost)/N."""
    d = earth_mover_distance(pc1.permute(0, 2, 1).contiguous(), pc2.permute(0, 2, 1).contiguous())
    return torch.mean(d) / pc1.shape[2]
