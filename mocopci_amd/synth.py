"""Synthetic NL-Drive-like inputs and deterministic by-name weights.

Neither the dataset nor checkpoints are available offline, so every test and benchmark uses
  * clouds from numpy PCG64(seed = 1000*config + sample): an outdoor-LiDAR-like box, a rigid
    motion between the two middle frames, independent permutations, 5 % duplicated points
    (the reference loader pads by sampling with replacement, data/no_norm_datasets.py:52-55);
  * weights generated from the state-dict entry NAME (crc32 seed), so the same tensors are
    produced here, on the GPU box, and inside the golden generator without shipping any file.
"""
import math
import zlib

import numpy as np
import torch

T_INTERP = (0.41666666666666663, 0.5, 0.5833333333333333)  # mocopci.py:824 == test.py:38-44


def _rigid(points, t, yaw_deg=1.0, trans=(1.0, 0.2, 0.0)):
    a = math.radians(yaw_deg) * t
    c, s = math.cos(a), math.sin(a)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=np.float64)
    return points.astype(np.float64) @ R.T + np.asarray(trans, dtype=np.float64) * t


def make_sequence(seed, n, extent=40.0, zlo=-3.0, zhi=3.0, noise=0.05, dup_frac=0.05):
    """One sample: (frame at t=0, frame at t=1, [3 GT frames]) as float32 (n,3) arrays."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = np.empty((n, 3), dtype=np.float64)
    base[:, 0] = rng.uniform(-extent, extent, n)
    base[:, 1] = rng.uniform(-extent, extent, n)
    base[:, 2] = rng.uniform(zlo, zhi, n)
    ndup = int(n * dup_frac)
    if ndup:
        src = rng.integers(0, n - ndup, ndup)
        base[n - ndup:] = base[src]  # exact duplicates: exercises the tie rules

    def frame(t):
        p = _rigid(base, t) + rng.normal(0.0, noise, (n, 3))
        if ndup:
            p[n - ndup:] = p[src]
        return p[rng.permutation(n)].astype(np.float32)

    f0, f1 = frame(0.0), frame(1.0)
    gts = [frame(t) for t in T_INTERP]
    return f0, f1, gts


def make_batch(config_id, batch, n, device="cpu", first_sample=0, **kw):
    """Returns xyz1, xyz2 as (B,3,N) tensors (the layout test.py:73-76 feeds the model) and gt: 3 x (B,N,3)."""
    f0s, f1s, gts = [], [], [[], [], []]
    for s in range(batch):
        f0, f1, g = make_sequence(1000 * config_id + first_sample + s, n, **kw)
        f0s.append(f0)
        f1s.append(f1)
        for j in range(3):
            gts[j].append(g[j])
    xyz1 = torch.from_numpy(np.stack(f0s)).permute(0, 2, 1).contiguous().to(device)
    xyz2 = torch.from_numpy(np.stack(f1s)).permute(0, 2, 1).contiguous().to(device)
    gt = [torch.from_numpy(np.stack(g)).to(device) for g in gts]
    return xyz1, xyz2, gt


def weights_by_name(spec_or_state_dict):
    """Deterministic tensors for every state-dict entry, keyed by name only."""
    out = {}
    for name, v in spec_or_state_dict.items():
        shape = tuple(v["shape"]) if isinstance(v, dict) else tuple(v.shape)
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) + 0.5
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "gamma":
            t = 0.5 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) >= 2 and math.prod(shape[1:]) > 1:
            fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g) * (0.5 / math.sqrt(fan_in))  # 0.5: keeps activations O(1) through the pyramid
        elif leaf == "weight" and name.endswith("act.weight"):
            t = torch.full(shape, 0.25)  # PReLU slope
        elif leaf == "weight" and "dwconv" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)  # depthwise k=1 conv: per-channel scale
        elif leaf == "weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)  # norm scales, (out,1,..) degenerate weights
        else:
            t = 0.05 * torch.randn(shape, generator=g)  # biases
        out[name] = t
    return out
