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
"""
import importlib
import os
import sys
import types

import torch

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _install_standins():
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

    sys.modules["pointnet2_cuda"] = types.ModuleType("pointnet2_cuda")
    sys.modules.setdefault("emd_cuda", types.ModuleType("emd_cuda"))

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
        _bind(importlib.import_module(name))


_loaded = {}


def load():
    """Returns a namespace with the reference modules: pointconv_util, pointT_layer2, mocopci."""
    if not _loaded:
        if not available():
            raise RuntimeError("reference tree not present (this harness only runs in the build container)")
        _install_standins()
        _loaded["pointconv_util"] = importlib.import_module("models.pointconv_util")
        _loaded["pointT_layer2"] = importlib.import_module("models.pointT_layer2")
        _loaded["mocopci"] = importlib.import_module("models.m_models.mocopci")
        _loaded["pointnet2_utils"] = importlib.import_module("pointnet2.pointnet2_utils")
    return types.SimpleNamespace(**_loaded)
