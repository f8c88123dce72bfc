"""Operator API of the reference's pointnet2/pointnet2_utils.py (and its duplicate
models/pointnet2/pointnet2_utils.py) on top of the MI355X kernels: same public
names, argument order, dtypes (int32 indices), shapes and autograd contract
(index-producing ops return None grads; gather/group/interpolate backward is a
scatter-add).  Reference lines are cited per class."""
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_cuda as pointnet2


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _i32(shape, like):
    return torch.empty(shape, dtype=torch.int32, device=like.device)


class FurthestPointSampling(Function):
    """pointnet2_utils.py:10-33."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = _i32((B, npoint), xyz)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """pointnet2_utils.py:39-70."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = _f32((B, C, npoint), features)
        pointnet2.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features, accumulate=False)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """pointnet2_utils.py:76-102: returns (sqrt(dist2), idx)."""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = _f32((B, N, 3), unknown)
        idx = _i32((B, N, 3), unknown)
        pointnet2.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """pointnet2_utils.py:108-150."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = _f32((B, c, n), features)
        pointnet2.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.empty((B, c, m), dtype=torch.float32, device=grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features, accumulate=False)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """pointnet2_utils.py:156-194."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = _f32((B, C, nfeatures, nsample), features)
        pointnet2.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features, accumulate=False)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """pointnet2_utils.py:200-225."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class QueryAndGroup(nn.Module):
    """Ball query + grouping of coordinates (relative to the centre) and features: the module of pointnet2/pointnet2_utils.py:231-264.
    xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) or None -> (B, 3+C | C | 3, npoint, nsample).
    Inference takes ONE launch (mcp_query_and_group: the hits of a centre never leave the compute unit, no transposed copy of the
    cloud, no concatenation); when a gradient is wanted the result is composed from the differentiable functions above."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        if features is None and not self.use_xyz:
            raise AssertionError("QueryAndGroup without features needs use_xyz=True: there would be nothing to group")
        with_xyz = self.use_xyz or features is None
        if self.nsample <= 64 and not _wants_grad(xyz, new_xyz, features):
            from . import _lib
            xyz, new_xyz = xyz.detach().contiguous(), new_xyz.detach().contiguous()
            feats = None if features is None else features.detach().contiguous()
            B, N, _ = xyz.shape
            M = new_xyz.shape[1]
            C = 0 if feats is None else feats.shape[1]
            out = torch.empty((B, (3 if with_xyz else 0) + C, M, self.nsample), dtype=torch.float32, device=xyz.device)
            with torch.cuda.device(xyz.device):
                _lib.check(_lib.load().mcp_query_and_group(B, N, M, C, float(self.radius), int(self.nsample), int(bool(self.use_xyz)),
                                                           _lib.fptr(xyz), _lib.fptr(new_xyz), None if feats is None else _lib.fptr(feats),
                                                           _lib.fptr(out), torch.cuda.current_stream(xyz.device).cuda_stream))
            return out
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        parts = []
        if with_xyz:  # neighbour coordinates relative to their centre, channel-major as grouping_operation returns them
            parts.append(grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1))
        if features is not None:
            parts.append(grouping_operation(features, idx))
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


class GroupAll(nn.Module):
    """The whole cloud as one group (pointnet2/pointnet2_utils.py:267-290): (B, 3+C | C | 3, 1, N); new_xyz is ignored."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        parts = [xyz.transpose(1, 2).unsqueeze(2)] if (self.use_xyz or features is None) else []
        if features is not None:
            parts.append(features.unsqueeze(2))
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
