"""EMD metric.  The reference's only known-answer script for this repo is
models/EMD/test_emd_loss.py: two 2-point clouds whose optimal matching costs 0.30 + 0.41 = 0.71 per batch element."""
import pytest
import torch

from oracle import pointset as orc

P1 = [[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]   # models/EMD/test_emd_loss.py:7-10
P2 = [[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]


def test_oracle_reproduces_reference_known_answer():
    p1, p2 = torch.tensor([P1]).repeat(3, 1, 1), torch.tensor([P2]).repeat(3, 1, 1)
    cost, match = orc.earth_mover_distance(p1, p2, return_match=True)
    assert torch.allclose(cost, torch.full((3,), 0.71), atol=1e-5)                    # gt_dist terms, :16-18
    assert torch.allclose(match[0], torch.tensor([[0.0, 1.0], [1.0, 0.0]]), atol=1e-6)  # p1[0]<->p2[1], p1[1]<->p2[0]
    loss = cost[0] / 2 + cost[1] * 2 + cost[2] / 3                                     # :43
    assert abs(float(loss) - 0.71 * (0.5 + 2 + 1 / 3)) < 1e-4                          # 2.0117


@pytest.mark.gpu
def test_hip_emd_known_answer_and_oracle_parity():
    from mocopci_amd import emd
    dev = "cuda:0"
    p1, p2 = torch.tensor([P1]).repeat(3, 1, 1).to(dev), torch.tensor([P2]).repeat(3, 1, 1).to(dev)
    cost = emd.earth_mover_distance(p1, p2, transpose=False)
    assert torch.allclose(cost.cpu(), torch.full((3,), 0.71), atol=1e-5)
    g = torch.Generator().manual_seed(3)
    for n, m, b in ((512, 512, 2), (1000, 500, 2), (300, 900, 2), (2048, 2048, 2), (4096, 4096, 1)):
        x = torch.rand(b, n, 3, generator=g) * 4
        y = torch.rand(b, m, 3, generator=g) * 4
        want, wmatch = orc.earth_mover_distance(x, y, return_match=True)
        got = emd.earth_mover_distance(x.to(dev), y.to(dev), transpose=False).cpu()
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)                    # north_star: EMD within 1e-5 relative
        # the transport plan itself is only defined up to the exponential's rounding (the device uses the fast
        # __expf like the reference, the oracle expf): entries in [0,1] agree to 2e-3, row/column mass exactly enough
        gmatch = emd.approxmatch_forward(x.to(dev), y.to(dev)).cpu()
        assert float((gmatch - wmatch).abs().max()) < 2e-3
        torch.testing.assert_close(gmatch.sum(1), wmatch.sum(1), rtol=1e-4, atol=1e-5)
    # reference-signature wrappers: (B,3,N) layout and the per-point normalisation of models/utils.py:223-235
    x = (torch.rand(1, 3, 1024, generator=g) * 4).to(dev)
    y = (x + 0.01).contiguous()
    assert abs(float(emd.EMD(x, y)) - float(orc.EMD(x.cpu(), y.cpu()))) <= 1e-5 * float(orc.EMD(x.cpu(), y.cpu())) + 1e-9


@pytest.mark.gpu
def test_hip_emd_full_size_properties():
    # BASELINE N=8192: mass conservation of the transport plan is implied by cost(x,x) ~ 0 and cost scaling ~ shift^2
    from mocopci_amd import emd
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(1, 8192, 3, generator=g) * torch.tensor([80.0, 80.0, 6.0])).to("cuda:0")
    c0 = float(emd.earth_mover_distance(x, x, transpose=False))
    c1 = float(emd.earth_mover_distance(x, (x + 0.05).contiguous(), transpose=False))
    assert c0 / 8192 < 1e-3 and c1 > c0 and abs(c1 / 8192 - 3 * 0.05 ** 2) < 5e-3


@pytest.mark.gpu
def test_hip_emd_baseline_point_count_matches_oracle():
    """N=8192 (BASELINE configs[1]'s point count), one LiDAR-like pair: cost within the north-star 1e-5 relative of the
    oracle (about 2e9 exponentials on the host: ~20 s)."""
    from mocopci_amd import emd
    g = torch.Generator().manual_seed(8)
    x = torch.rand(1, 8192, 3, generator=g) * torch.tensor([80.0, 80.0, 6.0])
    y = (x[:, torch.randperm(8192, generator=g)] + 0.05 * torch.randn(1, 8192, 3, generator=g)).contiguous()
    want = orc.earth_mover_distance(x, y)
    got = emd.earth_mover_distance(x.to("cuda:0"), y.to("cuda:0"), transpose=False).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
