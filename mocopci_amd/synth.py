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


def weights_on_scan(spec_or_state_dict, flow_scale=0.1):
    """A second deterministic weight set under which the predicted frames stay ON THE SCAN, so that Chamfer / EMD against the
    ground truth are informative (with weights_by_name() the random `pred` head collapses every frame to a ~1-unit blob and
    Chamfer-vs-GT is E|gt|^2 whatever the kernels do).  No checkpoint exists offline; this is weights_by_name() with a
    hand-built coordinate-carrying path through the refinement branch (mocopci.py:1021-1053), everything else unchanged:
      * a signed value v survives Conv1d + LeakyReLU(0.1) / ReLU layers as the channel pair (act(v), act(-v)):
        act(v) - act(-v) = 1.1 v (LeakyReLU) or v (ReLU), so each Linear decodes the pair and re-encodes it;
      * encoder.level0_lift writes (+xyz, -xyz) into channels 0..5; the three PointConv layers on the path get a constant WeightNet
        (last conv: weight 0, bias 1), which turns their aggregation into a plain sum over the 32 neighbours, and a Linear that
        picks the neighbourhood MEAN of the decoded coordinates; rlevel0 passes the pair on; the TransformerBlock keeps it in its
        residual (fc2 rows zeroed); 3-NN upsampling is linear; `pred` decodes to xyz.
    The refined cloud is then a smoothed copy of the warped input frame (local means over the 32-neighbourhoods the KNN / FPS /
    grouping / interpolation kernels select -- a wrong neighbour list, sample or weight moves it), and the fusion stage averages
    it once more.  The flow heads (mapping_xyz) are scaled by `flow_scale` so the motion branch stays small but live."""
    w = weights_by_name(spec_or_state_dict)
    pair = 1.0 / 1.1

    def recode(t, bias, cols_plus, cols_minus, gain):
        """rows 0..2 <- +gain*(in[cols_plus] - in[cols_minus]), rows 3..5 <- the negative; their biases 0"""
        t[:6] = 0.0
        for i in range(3):
            t[i, cols_plus[i]], t[i, cols_minus[i]] = gain, -gain
            t[i + 3, cols_plus[i]], t[i + 3, cols_minus[i]] = -gain, gain
        bias[:6] = 0.0

    lift = "encoder.level0_lift.composed_module.0"
    w[lift + ".weight"][:6] = 0.0
    for i in range(3):
        w[lift + ".weight"][i, i, 0], w[lift + ".weight"][i + 3, i, 0] = 1.0, -1.0
    w[lift + ".bias"][:6] = 0.0
    for pc, d in (("encoder.level0", 32), ("multi_frame_inference.level1", 64)):
        w[pc + ".weightnet.mlp_convs.2.weight"].zero_()
        w[pc + ".weightnet.mlp_convs.2.bias"].fill_(1.0)
        # aggregate index (3 + channel) * 8 + slot: slot 0 of the pair's feature channels, mean over the 32 neighbours
        recode(w[pc + ".linear.weight"], w[pc + ".linear.bias"], [(3 + i) * 8 for i in range(3)], [(3 + i + 3) * 8 for i in range(3)], pair / 32.0)
    r0 = "multi_frame_inference.rlevel0.composed_module.0"
    recode(w[r0 + ".weight"][:, :, 0], w[r0 + ".bias"], [0, 1, 2], [3, 4, 5], pair)
    sh = "multi_frame_inference.shape1"
    w[sh + ".fc2.weight"][:6] = 0.0
    w[sh + ".fc2.bias"][:6] = 0.0
    recode(w["multi_frame_inference.pred.0.weight"], w["multi_frame_inference.pred.0.bias"], [0, 1, 2], [3, 4, 5], pair)
    p2 = "multi_frame_inference.pred.2"
    w[p2 + ".weight"].zero_()
    w[p2 + ".bias"].zero_()
    for i in range(3):
        w[p2 + ".weight"][i, i], w[p2 + ".weight"][i, i + 3] = 1.0, -1.0
    for name in w:
        if ".mapping_xyz." in name:
            w[name] *= flow_scale
    return w
