"""Seeded inputs shared by oracle/make_golden.py (which stores the reference's outputs) and the
parity tests (which regenerate the same inputs).  Channel-last tensors, batch 1."""
import torch


def _r(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def layer_inputs():
    g = torch.Generator().manual_seed(7)
    xyz_a = torch.rand(1, 256, 3, generator=g) * torch.tensor([20.0, 20.0, 3.0])
    xyz_b = xyz_a[:, torch.randperm(256, generator=g)] + 0.3 * torch.randn(1, 256, 3, generator=g)
    xyz_a[:, 200:208] = xyz_a[:, 10:18]  # exact duplicates
    xyz_big = torch.rand(1, 1024, 3, generator=g) * torch.tensor([20.0, 20.0, 3.0])
    return {
        "xyz_a": xyz_a.contiguous(), "xyz_b": xyz_b.contiguous(), "xyz_big": xyz_big.contiguous(),
        "flow_a": _r(11, 1, 256, 3, scale=0.5),
        "f32_a": _r(12, 1, 256, 32), "f64_a": _r(13, 1, 256, 64), "f64_b": _r(14, 1, 256, 64),
        "f256_a": _r(15, 1, 256, 256), "f256_b": _r(16, 1, 256, 256), "f256_big": _r(17, 1, 1024, 256),
        "c576_a": _r(18, 1, 256, 576), "c576_b": _r(19, 1, 256, 576),
        "c192_a": _r(20, 1, 256, 192), "c192_b": _r(21, 1, 256, 192),
    }


def module_inputs():
    """Inputs of the QueryAndGroup / GroupAll fixtures (pointnet2_modules.npz): a 1024-point cloud, 64 centres taken from it
    (so some balls hold only their centre and some overflow nsample), 16-channel features in the reference's (B,C,N) layout."""
    xyz = layer_inputs()["xyz_big"]
    return {"xyz": xyz, "new_xyz": xyz[:, ::16].contiguous(), "features": _r(22, 1, 16, 1024)}


def big_cloud(n, seed=1, batch=1):
    """Seeded LiDAR-like box cloud with 5 % duplicated points (full-size hash pins)."""
    g = torch.Generator().manual_seed(1000 * seed + n)
    x = (torch.rand(batch, n, 3, generator=g) * 2 - 1) * torch.tensor([40.0, 40.0, 3.0])
    nd = n // 20
    src = torch.randint(0, n - nd, (nd,), generator=g)
    x[:, n - nd:] = x[:, src]
    return x[:, torch.randperm(n, generator=g)].contiguous()
