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


def _segments(idx, n):
    """Gather positions of every batch element sorted by (destination, position): (order (B,T) int32, seg (B,n+1) int32 CSR
    offsets) -- mcp_scatter_segments (a counting sort on the small keys).  Scratch comes from torch's allocator here, in the
    wrapper -- the library itself never allocates."""
    B = idx.shape[0]
    flat = idx.reshape(B, -1).int().contiguous()
    T = flat.shape[1]
    order = torch.empty((B, T), dtype=torch.int32, device=idx.device)
    seg = torch.empty((B, n + 1), dtype=torch.int32, device=idx.device)
    need = _lib.load().mcp_scatter_segments_workspace_bytes(B, T, n)
    if need == 0:   # more destinations than the kernel's LDS histograms hold: a general stable sort
        keys, order64 = torch.sort(flat, dim=1, stable=True)
        bounds = torch.arange(n + 1, device=idx.device, dtype=keys.dtype).expand(B, n + 1).contiguous()
        return order64.int().contiguous(), torch.searchsorted(keys.contiguous(), bounds).int().contiguous()
    ws = torch.empty((need,), dtype=torch.uint8, device=idx.device)
    _call("mcp_scatter_segments", flat, B, T, n, _lib.iptr(flat), _lib.iptr(order), _lib.iptr(seg), ws.data_ptr(), need)
    return order, seg


# The three backward scatters: the reference's kernels add with atomicAdd (sampling_gpu.cu:46-83, group_points_gpu.cu:8-44,
# interpolate_gpu.cu:120-161), so their sums depend on the order the hardware serves the atomics in.  These wrappers keep the
# reference's argument lists and its CONTRACT -- the scatter is ADDED to what grad_points holds (the reference's callers pass
# zeros, pointnet2_utils.py:67,146,190; a caller that accumulates into a live buffer gets the same sum here) -- but reduce each
# destination's addends in ascending position order first, so the result is bit-reproducible.
# accumulate=False (keyword only, used by this package's own autograd.Functions): grad_points is uninitialised scratch and is
# overwritten -- no zero fill, no second pass.
def _scatter(entry, grad_out, grad_points, accumulate, *args):
    dst = torch.empty_like(grad_points) if accumulate else grad_points
    rc = _call(entry, grad_out, *args, _lib.fptr(dst))
    if accumulate:
        grad_points.add_(dst)
    return rc


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points, *, accumulate=True):
    order, seg = _segments(idx, n)
    return _scatter("mcp_group_points_grad_sorted", grad_out, grad_points, accumulate, b, c, n, npoints, _lib.fptr(grad_out), _lib.iptr(order),
                    _lib.iptr(seg))


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    return _call("mcp_group_points", points, b, c, n, npoints, nsample, _lib.fptr(points), _lib.iptr(idx), _lib.fptr(out))


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points, *, accumulate=True):
    order, seg = _segments(idx, n)
    return _scatter("mcp_group_points_grad_sorted", grad_out, grad_points, accumulate, b, c, n, npoints * nsample, _lib.fptr(grad_out),
                    _lib.iptr(order), _lib.iptr(seg))


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    return _call("mcp_ball_query", xyz, b, n, m, float(radius), nsample, _lib.fptr(new_xyz), _lib.fptr(xyz), _lib.iptr(idx))


_searcher = None


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    """The three nearest known points are the K = 3 case of the package's neighbour search in its direct distance form (same
    fma chain, ties to the lower index: interpolate_gpu.cu:26-49 inserts on strict <).  Large clouds therefore go through the
    tile-pruned search over space-ordered copies (mcp_build_cloud + mcp_knn_pruned), which reads a few tiles of `known` per query
    instead of all m points; the copies are scratch from torch's allocator.  Small clouds run the exhaustive mcp_three_nn."""
    global _searcher
    from . import ops
    if m >= ops.HipBackend.PRUNE_MIN_REFS and n >= ops.HipBackend.PRUNE_MIN_QUERIES and max(n, m) <= 65536:
        if _searcher is None:
            _searcher = ops.HipBackend()
        _lib.fptr(dist2), _lib.iptr(idx)  # dtype / device / contiguity of the caller's outputs
        ks, kperm, boxes = _searcher._build_cloud(known)
        us, uperm, _ = _searcher._build_cloud(unknown)
        return _call("mcp_knn_pruned", unknown, b, n, m, 3, ops.MCP_DIST_DIRECT, _lib.fptr(us), _lib.iptr(uperm), _lib.fptr(ks),
                     _lib.iptr(kperm), _lib.fptr(boxes), _lib.iptr(idx), _lib.fptr(dist2))
    return _call("mcp_three_nn", unknown, b, n, m, _lib.fptr(unknown), _lib.fptr(known), _lib.fptr(dist2), _lib.iptr(idx))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _call("mcp_three_interpolate", points, b, c, m, n, _lib.fptr(points), _lib.iptr(idx), _lib.fptr(weight), _lib.fptr(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points, *, accumulate=True):
    order, seg = _segments(idx, m)
    _scatter("mcp_three_interpolate_grad_sorted", grad_out, grad_points, accumulate, b, c, n, m, _lib.fptr(grad_out), _lib.iptr(order),
             _lib.iptr(seg), _lib.fptr(weight))
