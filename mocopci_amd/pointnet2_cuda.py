"""Drop-in for the reference's pybind11 module `pointnet2_cuda`
(pointnet2/src/pointnet2_api.cpp:10-24): the same nine entry points with the same
argument lists (ints then tensors), backed by libmocopci_hip.so through its C ABI.
Each call is enqueued on torch's current stream of the tensors' device; outputs
are caller-allocated exactly as in the reference wrappers."""
import torch

from . import _lib


def _call(name, ref_tensor, *args):
    lib = _lib.load()
    with torch.cuda.device(ref_tensor.device):
        _lib.check(getattr(lib, name)(*args, _lib.stream()))
    return 1  # the reference wrappers return 1 (sampling.cpp:21)


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    # the library never allocates: sizes that can use scratch (16384 < N <= 65536) get it from torch's caching allocator
    need = _lib.load().mcp_fps_workspace_bytes(b, n, m)
    if need == 0:
        return _call("mcp_furthest_point_sampling", points, b, n, m, _lib.fptr(points), _lib.fptr(temp), _lib.iptr(idx))
    ws = torch.empty((need,), dtype=torch.uint8, device=points.device)
    return _call("mcp_furthest_point_sampling_ws", points, b, n, m, _lib.fptr(points), _lib.fptr(temp), _lib.iptr(idx), ws.data_ptr(), need)


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    return _call("mcp_gather_points", points, b, c, n, npoints, _lib.fptr(points), _lib.iptr(idx), _lib.fptr(out))


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    return _call("mcp_gather_points_grad", grad_out, b, c, n, npoints, _lib.fptr(grad_out), _lib.iptr(idx), _lib.fptr(grad_points))


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    return _call("mcp_group_points", points, b, c, n, npoints, nsample, _lib.fptr(points), _lib.iptr(idx), _lib.fptr(out))


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    return _call("mcp_group_points_grad", grad_out, b, c, n, npoints, nsample, _lib.fptr(grad_out), _lib.iptr(idx),
                 _lib.fptr(grad_points))


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    return _call("mcp_ball_query", xyz, b, n, m, float(radius), nsample, _lib.fptr(new_xyz), _lib.fptr(xyz), _lib.iptr(idx))


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _call("mcp_three_nn", unknown, b, n, m, _lib.fptr(unknown), _lib.fptr(known), _lib.fptr(dist2), _lib.iptr(idx))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _call("mcp_three_interpolate", points, b, c, m, n, _lib.fptr(points), _lib.iptr(idx), _lib.fptr(weight), _lib.fptr(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _call("mcp_three_interpolate_grad", grad_out, b, c, n, m, _lib.fptr(grad_out), _lib.iptr(idx), _lib.fptr(weight),
          _lib.fptr(grad_points))
