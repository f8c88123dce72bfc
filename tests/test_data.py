"""NL-Drive data path: format round trip, sampling rule, and (in the build container) identity with the
reference's own NLDriveDataset on the same files and the same np.random seed."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from mocopci_amd import data

REF = "/root/reference/data/no_norm_datasets.py"


def make_files(tmp_path, sizes=(9000, 8192, 5000, 12000, 8500, 300, 8192)):
    rng = np.random.default_rng(0)
    names = []
    for i, n in enumerate(sizes):
        name = f"scene00_seq0001_frame{i:02d}.bin"
        data.write_frame(tmp_path / name, rng.normal(size=(n, 3)).astype(np.float32) * 20)
        names.append(name)
    lst = tmp_path / "list.txt"
    lst.write_text(" ".join(names) + "\n")
    return str(tmp_path), str(lst), sizes


def test_format_and_sampling_rule(tmp_path):
    root, lst, sizes = make_files(tmp_path)
    ds = data.NLDriveDataset(root, lst, num_points=8192)
    assert len(ds) == 1
    np.random.seed(1)
    inp, gt = ds[0]
    assert len(inp) == 4 and len(gt) == 3 and all(t.shape == (8192, 3) and t.dtype == torch.float32 for t in inp + gt)
    raw = data.read_frame(os.path.join(root, "scene00_seq0001_frame02.bin"))       # 5000 points < 8192
    assert torch.equal(inp[2][:5000], torch.from_numpy(raw))                         # all points first, in order ...
    assert len(np.unique(inp[2].numpy(), axis=0)) == 5000                            # ... then a fill with replacement
    assert len(np.unique(inp[0].numpy(), axis=0)) == 8192                            # 9000 >= 8192: subset without replacement
    assert gt[1].shape == (8192, 3)                                                  # 300-point frame is padded too


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree only exists in the build container")
def test_identical_to_reference_dataset(tmp_path):
    spec = importlib.util.spec_from_file_location("ref_no_norm_datasets", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    root, lst, _ = make_files(tmp_path)
    np.random.seed(123)
    want_in, want_gt = mod.NLDriveDataset(root, lst, num_points=8192)[0]
    np.random.seed(123)
    got_in, got_gt = data.NLDriveDataset(root, lst, num_points=8192)[0]
    for a, b in zip(want_in + want_gt, got_in + got_gt):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_evaluate_loop_on_synthetic_files(tmp_path):
    from torch.utils.data import DataLoader
    from tests import harness_checks as hc
    root, lst, _ = make_files(tmp_path, sizes=(2500, 2048, 2048, 2100, 2048, 2048, 1500))
    ds = data.NLDriveDataset(root, lst, num_points=2048)
    np.random.seed(0)
    res = data.evaluate(hc.build_model("cuda:0"), DataLoader(ds, batch_size=1), device="cuda:0")
    assert res["sequences"] == 1 and all(np.isfinite(res["chamfer"])) and all(np.isfinite(res["emd"])) and res["seconds_per_forward"] > 0


@pytest.mark.gpu
def test_gpu_side_resampling_is_identical_to_the_host_path(tmp_path):
    """data/no_norm_datasets.py:52-55 with the row selection on the GPU: same np.random call order, so the same sample bit
    for bit (subset without replacement, and all-points-then-fill-with-replacement for short frames), and it feeds evaluate()."""
    from torch.utils.data import DataLoader
    from tests import harness_checks as hc
    root, lst, _ = make_files(tmp_path)
    np.random.seed(77)
    want_in, want_gt = data.NLDriveDataset(root, lst, num_points=8192)[0]
    np.random.seed(77)
    got_in, got_gt = data.NLDriveDataset(root, lst, num_points=8192, device="cuda:0")[0]
    for a, b in zip(want_in + want_gt, got_in + got_gt):
        assert b.is_cuda and b.dtype == torch.float32 and torch.equal(a, b.cpu())
    root2, lst2, _ = make_files(tmp_path / "small" if (tmp_path / "small").mkdir() is None else tmp_path, sizes=(2500, 2048, 2048, 2100, 2048, 2048, 1500))
    ds = data.NLDriveDataset(root2, lst2, num_points=2048, device="cuda:0")
    np.random.seed(0)
    res = data.evaluate(hc.build_model("cuda:0"), DataLoader(ds, batch_size=1), device="cuda:0")
    assert res["sequences"] == 1 and all(np.isfinite(res["chamfer"]))
