"""Drop-in installation for an unmodified checkout of the reference.

The reference reaches its extension through two import roots, both ending in `import pointnet2_cuda`
(pointnet2/pointnet2_utils.py:7; models/pointnet2/pointnet2_utils.py:7 via mocopci.py:8), and
models/layers.py:15 expects a `models.common` module that the repository does not ship.
`install()` registers this package's implementations under those names BEFORE the reference's
modules are imported:

    import mocopci_amd.compat as compat; compat.install()
    from models.m_models.mocopci import MoCoPCI      # the reference's own file, unmodified

`install(patch_helpers=True)` additionally rebinds the module-level helpers the reference defines in
Python (knn_point, knn_point_cosine, index_points_group, index_points_gather in pointconv_util and the
local copies in mocopci.py:1130-1215) to the fused kernels, keeping their signatures and int64 results.
"""
import sys
import types

import torch

from . import ops, pointnet2_cuda, pointnet2_utils


def _common_module():
    """`models.common` as implied by the call sites in models/layers.py:35,37,62,64,162,170."""
    m = types.ModuleType("models.common")
    m.fps = pointnet2_utils.furthest_point_sample                     # fps(xyz (B,N,3), npoint) -> (B,npoint) int32
    m.gather_points = pointnet2_utils.gather_operation                # gather_points(features (B,C,N), idx) -> (B,C,npoint)
    m.ball_query = pointnet2_utils.ball_query                         # ball_query(radius, nsample, xyz, new_xyz)
    m.three_nn = pointnet2_utils.three_nn
    m.three_interpolate = pointnet2_utils.three_interpolate
    m.group_points = pointnet2_utils.grouping_operation
    return m


# ---- reference-signature helpers on the fused kernels (channel-last in, like the reference's) ----
def knn_point(nsample, xyz, new_xyz):
    """mocopci.py:1158-1169: xyz (B,N,3) refs, new_xyz (B,S,3) queries -> (B,S,nsample) int64."""
    return ops.backend().knn(new_xyz.contiguous(), xyz.contiguous(), nsample).long()


def knn_point_cosine(nsample, xyz, new_xyz):
    """pointconv_util.py:142-153 on (B,N,C) features."""
    return ops.backend().knn_cosine(new_xyz.contiguous(), xyz.contiguous(), nsample).long()


def index_points_group(points, knn_idx):
    """mocopci.py:1204-1215: (B,N,C), (B,S,K) -> (B,S,K,C)."""
    return ops.backend().group_rows(points.contiguous(), knn_idx.int())


def index_points_gather(points, fps_idx):
    """mocopci.py:1190-1201: (B,N,C), (B,S) -> (B,S,C)."""
    return ops.backend().group_rows(points.contiguous(), fps_idx.int())


def knn_points(p1, p2, K=1, **_):
    """pytorch3d.ops.knn_points subset used at pointconv_util.py:910: returns (dists, idx, None)."""
    idx, dist = ops.backend().knn(p1.contiguous(), p2.contiguous(), K, mode=ops.MCP_DIST_DIRECT, return_dist=True)
    return dist, idx.long(), None


def install(patch_helpers=False):
    sys.modules["pointnet2_cuda"] = pointnet2_cuda
    for name in ("pointnet2.pointnet2_utils", "models.pointnet2.pointnet2_utils"):
        sys.modules[name] = pointnet2_utils
    for pkg in ("pointnet2", "models.pointnet2"):
        if pkg in sys.modules:
            setattr(sys.modules[pkg], "pointnet2_utils", pointnet2_utils)
    sys.modules.setdefault("models.common", _common_module())
    if patch_helpers:
        for modname in ("models.pointconv_util", "models.m_models.mocopci"):
            mod = sys.modules.get(modname)
            if mod is None:
                continue
            for fn in (knn_point, knn_point_cosine, index_points_group, index_points_gather):
                if hasattr(mod, fn.__name__):
                    setattr(mod, fn.__name__, fn)
            if hasattr(mod, "knn_points"):
                mod.knn_points = knn_points
