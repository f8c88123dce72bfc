"""oracle/ref_harness.py -- TEST INFRASTRUCTURE; runs ONLY in the build container.

Imports the reference's own Python layers from /root/reference (never copied,
never shipped) so that (a) the oracle restatements can be validated against
them and (b) golden fixtures can be generated (oracle/make_golden.py).
The reference needs three third-party modules that are absent here and whose
CUDA extension cannot be built; they are replaced by stand-ins BEFORE import:

  * pointnet2_cuda                 -> empty module; the six autograd wrappers in
                                      pointnet2_utils are rebound to the C oracle
                                      (oracle/pointset.py) in both import roots
  * pytorch3d.ops.knn_points       -> direct squared-L2, sorted, idx into 2nd arg
    pytorch3d.loss.chamfer_distance-> squared L2, point mean, batch mean, both dirs
  * timm.models.layers             -> DropPath (identity in eval), to_2tuple, trunc_normal_
  * sklearn is importable here (only imported, never used by the model)

torch.Tensor.cuda is patched to the identity because mocopci.py:199,205,518,571
hard-code .cuda().  Nothing here runs on the GPU box (/root/reference is absent there).

Two modes, chosen by the FIRST load() of a process (the reference binds the operator names at import time):
  * load()                 the six public operator names are rebound to forward-only oracle calls (inference goldens);
  * load(own_autograd=True) the reference's six autograd.Functions run UNCHANGED -- their forward and their BACKWARD
                           (pointnet2/pointnet2_utils.py:39-73, :108-153, :156-197) -- on a `pointnet2_cuda` stand-in that exports
                           the nine *_wrapper entry points of pointnet2/src/pointnet2_api.cpp:10-24 bound to the C oracle (same
                           argument lists, results written into the caller's tensors), with torch.cuda.FloatTensor / IntTensor
                           patched to the CPU constructors the Functions allocate through.  Used for the training-gradient
                           fixture (oracle/make_golden.py train_grad_*): loss.backward() then IS the reference's autograd.
"""
import importlib
import os
import sys
import types

import torch

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _wrapper_module():
    """`pointnet2_cuda` stand-in with the nine entry points of pointnet2/src/pointnet2_api.cpp:10-24 on the C oracle
    (sampling.cpp:11-49, group_points.cpp:11-37, ball_query.cpp:16-28, interpolate.cpp:14-56: same integer argument order; outputs
    and gradient buffers are the caller's tensors, written / accumulated in place as the CUDA kernels do)."""
    import ctypes
    from oracle import pointset as orc
    L, f, i = orc.lib(), orc._f, orc._i
    m = types.ModuleType("pointnet2_cuda")

    def furthest_point_sampling_wrapper(b, n, m_, points, temp, idx):
        L.orc_fps(f(points), f(temp), i(idx), b, n, m_)
        return 1

    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        L.orc_gather(f(points), i(idx), f(out), b, c, n, npoints)
        return 1

    def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
        L.orc_gather_grad(f(grad_out), i(idx), f(grad_points), b, c, n, npoints)
        return 1

    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        L.orc_group(f(points), i(idx), f(out), b, c, n, npoints, nsample)
        return 1

    def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
        L.orc_group_grad(f(grad_out), i(idx), f(grad_points), b, c, n, npoints, nsample)
        return 1

    def ball_query_wrapper(b, n, m_, radius, nsample, new_xyz, xyz, idx):
        L.orc_ball_query(f(new_xyz), f(xyz), i(idx), b, n, m_, ctypes.c_float(radius), nsample)
        return 1

    def three_nn_wrapper(b, n, m_, unknown, known, dist2, idx):
        L.orc_three_nn(f(unknown), f(known), f(dist2), i(idx), b, n, m_)

    def three_interpolate_wrapper(b, c, m_, n, points, idx, weight, out):
        L.orc_three_interpolate(f(points), i(idx), f(weight), f(out), b, c, m_, n)

    def three_interpolate_grad_wrapper(b, c, n, m_, grad_out, idx, weight, grad_points):
        L.orc_three_interpolate_grad(f(grad_out), i(idx), f(weight), f(grad_points), b, c, n, m_)

    for fn in (furthest_point_sampling_wrapper, gather_points_wrapper, gather_points_grad_wrapper, group_points_wrapper,
               group_points_grad_wrapper, ball_query_wrapper, three_nn_wrapper, three_interpolate_wrapper, three_interpolate_grad_wrapper):
        setattr(m, fn.__name__, fn)
    return m


def _install_standins(own_autograd=False):
    from oracle import pointset as orc

    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    # timm
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.training and self.drop_prob > 0:
                raise RuntimeError("stand-in DropPath supports eval() only")
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    timm_layers.DropPath, timm_layers.to_2tuple, timm_layers.trunc_normal_ = DropPath, to_2tuple, trunc_normal_
    timm.models, timm_models.layers = timm_models, timm_layers
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})

    # pytorch3d
    p3d = types.ModuleType("pytorch3d")
    p3d_ops = types.ModuleType("pytorch3d.ops")
    p3d_loss = types.ModuleType("pytorch3d.loss")

    def knn_points(p1, p2, K=1, **kw):
        idx, dist = orc.knn(p1.contiguous().float(), p2.contiguous().float(), K, mode=1, return_dist=True)
        return dist, idx.long(), None

    def knn_gather(x, idx):
        B, N, C = x.shape
        return torch.gather(x[:, :, None].expand(-1, -1, idx.shape[2], -1), 1, idx[..., None].expand(-1, -1, -1, C))

    def chamfer_distance(x, y, **kw):
        d = torch.cdist(x.double(), y.double()) ** 2
        cd = d.min(2)[0].mean(1) + d.min(1)[0].mean(1)
        return cd.mean().float(), None

    p3d_ops.knn_points, p3d_ops.knn_gather, p3d_loss.chamfer_distance = knn_points, knn_gather, chamfer_distance
    p3d.ops, p3d.loss = p3d_ops, p3d_loss
    sys.modules.update({"pytorch3d": p3d, "pytorch3d.ops": p3d_ops, "pytorch3d.loss": p3d_loss})

    sys.modules["pointnet2_cuda"] = _wrapper_module() if own_autograd else types.ModuleType("pointnet2_cuda")
    sys.modules.setdefault("emd_cuda", types.ModuleType("emd_cuda"))
    if own_autograd:
        # the Functions allocate through torch.cuda.FloatTensor(B, C, N) / torch.cuda.IntTensor(B, M) (pointnet2_utils.py:25-26, :52, ...)
        torch.cuda.FloatTensor, torch.cuda.IntTensor = torch.FloatTensor, torch.IntTensor

    if not getattr(torch.Tensor, "_mcp_cuda_patched", False):
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.Tensor._mcp_cuda_patched = True

    def _bind(mod):
        mod.furthest_point_sample = lambda xyz, npoint: orc.furthest_point_sample(xyz, npoint)
        mod.gather_operation = lambda f, idx: orc.gather_operation(f, idx.int())
        mod.grouping_operation = lambda f, idx: orc.grouping_operation(f, idx.int())
        mod.ball_query = lambda r, ns, xyz, new_xyz: orc.ball_query(r, ns, xyz, new_xyz)
        mod.three_nn = lambda u, k: orc.three_nn(u, k)
        mod.three_interpolate = lambda f, idx, w: orc.three_interpolate(f, idx, w)

    for name in ("pointnet2.pointnet2_utils", "models.pointnet2.pointnet2_utils"):
        mod = importlib.import_module(name)
        if not own_autograd:
            _bind(mod)


_loaded = {}


def load(own_autograd=False):
    """Returns a namespace with the reference modules: pointconv_util, pointT_layer2, mocopci, pointnet2_utils (+ utils, the
    reference's chamfer_loss, in own_autograd mode)."""
    if _loaded and _loaded["own_autograd"] != own_autograd:
        raise RuntimeError("the reference is already imported in the other mode (one mode per process)")
    if not _loaded:
        if not available():
            raise RuntimeError("reference tree not present (this harness only runs in the build container)")
        _install_standins(own_autograd)
        _loaded["own_autograd"] = own_autograd
        _loaded["pointconv_util"] = importlib.import_module("models.pointconv_util")
        _loaded["pointT_layer2"] = importlib.import_module("models.pointT_layer2")
        _loaded["mocopci"] = importlib.import_module("models.m_models.mocopci")
        _loaded["pointnet2_utils"] = importlib.import_module("pointnet2.pointnet2_utils")
        if own_autograd:
            _loaded["utils"] = importlib.import_module("models.utils")
    return types.SimpleNamespace(**_loaded)
