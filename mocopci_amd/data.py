"""NL-Drive data path, same on-disk format and sampling rule as the reference's
data/no_norm_datasets.py:8-91:

  * a frame is a flat little-endian float32 file of xyz triples (np.fromfile(...).reshape(-1, 3), :47);
  * a scene list has one sequence per line: 4 input frame names (frames 01,05,09,13) followed by the
    ground-truth frames (06,07,08), space separated (:26-33);
  * every frame is resampled to num_points: a random subset without replacement when it has enough points,
    otherwise all points followed by a random fill WITH replacement (:52-55) -- which is where the exact
    duplicate points in the inputs come from;
  * __getitem__ -> (input: num_frames x (N,3) tensors, gt: (interval-1) x (N,3) tensors).
np.random is used in the reference's call order, so a seeded run reproduces the reference's sample.

GPU-side resampling (NLDriveDataset(device=...)): the raw scan is uploaded once, whole, and the row selection runs on the GPU
(mcp_group_rows); the index list still comes from np.random in the reference's call order (O(N) host work, no point data
touched on the host), so the sample is bit-identical to the host path's and the duplicated rows are exact duplicates.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def read_frame(path):
    return np.fromfile(path, dtype=np.float32, count=-1).reshape([-1, 3])


def write_frame(path, xyz):
    np.asarray(xyz, dtype=np.float32).reshape(-1, 3).tofile(path)


def resample_indices(num, num_points):
    if num >= num_points:
        return np.random.choice(num, num_points, replace=False)
    return np.concatenate((np.arange(num), np.random.choice(num, num_points - num, replace=True)), axis=-1)


def resample_on_device(raw, pick, device):
    """raw (n,3) float32 numpy scan, pick (num_points,) indices -> (num_points,3) float32 tensor on `device`, gathered there."""
    from . import ops
    pts = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float32)).to(device, non_blocking=True).unsqueeze(0)
    idx = torch.from_numpy(np.ascontiguousarray(pick, dtype=np.int32)).to(device, non_blocking=True).unsqueeze(0)
    return ops.backend().group_rows(pts, idx)[0]


class NLDriveDataset(Dataset):
    def __init__(self, data_root, scene_list, num_points=8192, interval=4, num_frames=4, device=None):
        """device: None = the reference's host path (numpy fancy indexing); a CUDA device = GPU-side resampling (use with
        num_workers=0: the frames come back as tensors of that device)."""
        super().__init__()
        self.device = device
        self.data_root, self.scene_list = data_root, scene_list
        self.num_points, self.interval, self.num_frames = num_points, interval, num_frames
        with open(scene_list, "r") as fh:
            self.velodynes = [line.strip("\n").split(" ") for line in fh.readlines()]

    def __len__(self):
        return len(self.velodynes)

    def __getitem__(self, index):
        names = self.velodynes[index]
        frames, picks = [], []
        for i in range(self.num_frames):
            raw = read_frame(os.path.join(self.data_root, names[i]))
            frames.append(raw)
            picks.append(resample_indices(raw.shape[0], self.num_points))
        num_gt = len(names) - self.num_frames
        gt_intv = num_gt // (self.interval - 1)
        gts, gpicks = [], []
        for i in range(self.interval - 1):
            raw = read_frame(os.path.join(self.data_root, names[3 + (i + 1) * gt_intv]))
            gts.append(raw)
            gpicks.append(resample_indices(raw.shape[0], self.num_points))
        if self.device is not None:
            return ([resample_on_device(f, p, self.device) for f, p in zip(frames, picks)],
                    [resample_on_device(f, p, self.device) for f, p in zip(gts, gpicks)])
        inp = [torch.from_numpy(f[p, :].astype("float32")) for f, p in zip(frames, picks)]
        gt = [torch.from_numpy(f[p, :].astype("float32")) for f, p in zip(gts, gpicks)]
        return inp, gt


def evaluate(net, loader, device="cuda"):
    """The evaluation loop of test.py:71-135 in its intended form (one forward -> 3 frames; test.py:84 passes
    train=True by mistake): per-frame Chamfer distance and EMD means, forward time with device sync."""
    import time

    from . import emd as emd_mod, ops
    cd = [[], [], []]
    emd = [[], [], []]
    seconds = []
    with torch.no_grad():
        for inp, gt in loader:
            inp = [t.permute(0, 2, 1).to(device).contiguous().float() for t in inp]   # (B,3,N), test.py:73-74
            gt = [t.to(device).contiguous().float() for t in gt]                      # (B,N,3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = net(inp[1], inp[2])                                                  # the two middle frames, test.py:84
            torch.cuda.synchronize()
            seconds.append(time.perf_counter() - t0)
            for j in range(3):
                cd[j].append(float(ops.backend().chamfer(out[j].contiguous(), gt[j])))
                emd[j].append(float(emd_mod.EMD(out[j].permute(0, 2, 1).contiguous(), gt[j].permute(0, 2, 1).contiguous())))
    mean = lambda v: float(np.mean(v)) if v else float("nan")
    return {"chamfer": [mean(c) for c in cd], "emd": [mean(e) for e in emd], "seconds_per_forward": mean(seconds),
            "sequences": len(loader.dataset)}
