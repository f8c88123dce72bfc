"""EMD metric with the reference's names (models/EMD/emd.py:26-45, models/utils.py:223-235) on the HIP kernels
(forward only; the reference's matchcost backward is SURVEY 8(f) next #3)."""
import torch

from . import _lib


def _emd(xyz1, xyz2, want_match):
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    cost = torch.empty((B,), dtype=torch.float32, device=xyz1.device)
    ws = torch.empty((B * (3 * N + 2 * M),), dtype=torch.float32, device=xyz1.device)
    match = torch.empty((B, M, N), dtype=torch.float32, device=xyz1.device) if want_match else None
    lib = _lib.load()
    with torch.cuda.device(xyz1.device):
        _lib.check(lib.mcp_emd(B, N, M, _lib.fptr(xyz1), _lib.fptr(xyz2), _lib.fptr(match) if want_match else None, _lib.fptr(cost),
                               _lib.fptr(ws), _lib.stream()))
    return cost, match


def approxmatch_forward(xyz1, xyz2):
    """emd_cuda.approxmatch_forward: (B,N,3),(B,M,3) -> match (B,M,N)."""
    return _emd(xyz1, xyz2, True)[1]


def earth_mover_distance(xyz1, xyz2, transpose=True):
    """models/EMD/emd.py:26-45: (b,3,n) inputs when transpose=True, (b,n,3) otherwise -> cost (b)."""
    if xyz1.dim() == 2:
        xyz1 = xyz1.unsqueeze(0)
    if xyz2.dim() == 2:
        xyz2 = xyz2.unsqueeze(0)
    if transpose:
        xyz1, xyz2 = xyz1.transpose(1, 2), xyz2.transpose(1, 2)
    return _emd(xyz1, xyz2, False)[0]


def EMD(pc1, pc2):
    """models/utils.py:223-235: pc1, pc2 (B,3,M) -> mean(cost) / M."""
    d = earth_mover_distance(pc1.permute(0, 2, 1).contiguous(), pc2.permute(0, 2, 1).contiguous(), transpose=False)
    return torch.mean(d) / pc1.shape[2]
