"""-m gpu: gradients of the fused layers.  Forward in training is the fused HIP kernel; the backward of the fusion layer,
the D = 64 / 128 cost volumes, the PointConv aggregation, the vector-attention block, the narrow-head attention, the per-point Linear and
the two-layer MLP is a hand-written kernel (or a composition of the streaming kernels) with the deterministic segmented-reduction
scatter -- so is the 3-neighbour blend's (gather-dot + weighted segmented reduction); the remaining layers (cross D = 256, the interpolation
weights, wide-head attention) differentiate their unfused twin (mocopci_amd/grad.py).
Each layer's gradients -- w.r.t. coordinates, features and weights -- are compared with float64 re-derivations or with torch autograd on
the CPU through the oracle backend's own restatement of the layer; one training iteration is compared with the gradients the REFERENCE
computes through its own autograd.Functions (tests/golden/train_grad_b1_n1024.npz) and, end to end, with the oracle backend."""
import pytest
import torch
import torch.nn.functional as F

from mocopci_amd import ops, synth, training
from oracle import pointset as orc
from oracle.backend import OracleBackend
from tests import harness_checks as hc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cloud(seed, b, n, scale=10.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(b, n, 3, generator=g) * 2 - 1) * scale


def rnd(seed, *shape, scale=1.0):
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(seed)) * scale


def compare_grads(fn_hip, fn_cpu, tensors, int_args=(), rtol=2e-4, names=None):
    """Both sides: out = fn(*leaves, *int_args); loss = <out, fixed random g>; gradients w.r.t. every leaf."""
    cpu = [t.clone().requires_grad_(True) for t in tensors]
    out_c = fn_cpu(*cpu, *int_args)
    g = rnd(99, *out_c.shape)
    gc = torch.autograd.grad(out_c, cpu, g, allow_unused=True)
    hip = [t.to(DEV).requires_grad_(True) for t in tensors]
    out_h = fn_hip(*hip, *[a.to(DEV) if isinstance(a, torch.Tensor) else a for a in int_args])
    assert out_h.requires_grad
    torch.testing.assert_close(out_h.detach().cpu(), out_c.detach(), rtol=5e-5, atol=5e-5)
    gh = torch.autograd.grad(out_h, hip, g.to(DEV), allow_unused=True)
    for k, (a, b) in enumerate(zip(gh, gc)):
        label = names[k] if names else k
        assert (a is None) == (b is None), label
        if b is None:
            continue
        # relative to the gradient's own scale, with an absolute floor: some gradients are identically zero in exact arithmetic
        # (a bias added before a softmax over neighbours) and only rounding noise on both sides
        scale = float(b.abs().max())
        err = float((a.cpu() - b).abs().max())
        assert err <= rtol * scale + 2e-5, f"grad {label}: max err {err:.2e}, gradient scale {scale:.2e}"


@pytest.mark.parametrize("C", [35, 64, 3, 1, 2, 4])
def test_group_rows_gradient_is_deterministic_and_exact(C):
    """Wide rows (one thread per channel or channel quad walks the segment) and narrow ones (C <= 4, the coordinate gradients: eight
    lanes share a destination); a third of the gathers hit three hot rows (segments of ~1600 entries), some rows are never gathered."""
    pts = rnd(1, 3, 700, C).to(DEV).requires_grad_(True)
    gen = torch.Generator().manual_seed(2)
    idx = torch.randint(0, 650, (3, 900, 16), generator=gen, dtype=torch.int32)
    hot = torch.rand(3, 900, 16, generator=gen) < 0.33
    idx[hot] = torch.randint(0, 3, (int(hot.sum()),), generator=gen, dtype=torch.int32) * 7
    idx = idx.to(DEV)
    g = rnd(3, 3, 900, 16, C).to(DEV)
    be = ops.backend()
    grads = [torch.autograd.grad(be.group_rows(pts, idx), pts, g)[0] for _ in range(3)]
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])          # segmented reduction: same bits every run
    want = torch.zeros(3, 700, C, dtype=torch.float64)
    want.index_put_((torch.arange(3).view(3, 1, 1).expand(3, 900, 16), idx.cpu().long()), g.cpu().double(), accumulate=True)
    torch.testing.assert_close(grads[0].cpu().double(), want, rtol=1e-5, atol=2e-4)   # hot rows: ~1600 addends of O(1) in fp32
    assert not grads[0][:, 650:].any()


@pytest.mark.parametrize("n,s,d", [(256, 256, 32), (1024, 256, 64)])
def test_pointconv_agg_gradients(n, s, d):
    s_xyz = cloud(10 + d, 2, n)
    new_xyz = s_xyz[:, :s].clone() if s <= n else cloud(11, 2, s)
    pts = rnd(12, 2, n, d)
    idx = orc.knn(new_xyz, s_xyz, 32)
    wn = [rnd(13, 8, 3, scale=0.5), rnd(14, 8, scale=0.1), rnd(15, 8, 8, scale=0.4), rnd(16, 8, scale=0.1), rnd(17, 8, 8, scale=0.4), rnd(18, 8, scale=0.1)]
    ob, be = OracleBackend(), ops.backend()
    f_h = lambda a, b, c, w0, b0, w1, b1, w2, b2: be.pointconv_agg(a, b, c, idx.to(DEV), w0, b0, w1, b1, w2, b2)
    f_c = lambda a, b, c, w0, b0, w1, b1, w2, b2: ob.pointconv_agg(a, b, c, idx, w0, b0, w1, b1, w2, b2)
    compare_grads(f_h, f_c, [s_xyz, new_xyz, pts, *wn], names=["s_xyz", "new_xyz", "points", "w0", "b0", "w1", "b1", "w2", "b2"])


@pytest.mark.parametrize("d,n", [(64, 256), (128, 200), (256, 64)])
def test_cross_layer_gradients(d, n):
    xyz1, xyz2 = cloud(20, 2, n), cloud(21, 2, n)
    p1, p2 = rnd(22, 2, n, d), rnd(23, 2, n, d)
    idx = torch.cat([orc.knn(xyz1, xyz2, 16), orc.knn(xyz2, xyz1, 16, mode=1)], dim=-1).contiguous()
    w = [rnd(24, d, 3, scale=0.3), rnd(25, d, scale=0.1), rnd(26, d, d, scale=d ** -0.5), rnd(27, d, scale=0.1)]
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda a, b, c, e, *ww: be.cross_layer(a, b, c, e, idx.to(DEV), *ww), lambda a, b, c, e, *ww: ob.cross_layer(a, b, c, e, idx, *ww),
                  [xyz1, xyz2, p1, p2, *w], names=["xyz1", "xyz2", "points1", "points2", "wpos", "bpos", "wmlp", "bmlp"])


def test_fusion_gradients():
    p1 = cloud(30, 2, 300)
    p2 = p1 + rnd(31, 2, 300, 3, scale=0.2)
    idx = torch.cat([orc.knn(p1, p1, 32), orc.knn(p1, p2, 32)], dim=-1).contiguous()
    ws = [rnd(32, 64, 4, scale=0.5), rnd(33, 64, scale=0.1), rnd(34, 64, 64, scale=0.125), rnd(35, 64, scale=0.1), rnd(36, 128, 64, scale=0.125),
          rnd(37, 128, scale=0.1)]
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda a, b, *w: be.fusion_mlp(a, b, idx.to(DEV), *w), lambda a, b, *w: ob.fusion_mlp(a, b, idx, *w), [p1, p2, *ws],
                  names=["p1", "p2", "w1", "b1", "w2", "b2", "w3", "b3"], rtol=5e-4)


def test_fusion_backward_kernel_matches_the_unfused_layer_and_repeats_bit_for_bit():
    """mcp_fusion_grad against autograd over the unfused layer (grad.fusion_twin) on the device, with the neighbour list given as the
    two searches' halves, a point count that leaves the last workgroup ragged, and duplicate neighbours (a point is its own
    neighbour: r = 0, where |r| takes the zero subgradient); two runs give identical bits (fixed-order weight-gradient sums)."""
    from mocopci_amd import grad
    be = ops.backend()
    p1 = cloud(130, 3, 1501).to(DEV)
    p2 = p1.clone()
    p2[1:] += rnd(131, 2, 1501, 3, scale=0.2).to(DEV)      # batch element 0: p2 == p1, so every point's first self-neighbour has r = 0
    halves = (be.knn(p1, p1, 32), be.knn(p1, p2, 32))
    ws = [rnd(132, 64, 4, scale=0.5), rnd(133, 64, scale=0.1), rnd(134, 64, 64, scale=0.125), rnd(135, 64, scale=0.1), rnd(136, 128, 64, scale=0.125),
          rnd(137, 128, scale=0.1)]
    g = rnd(138, 3, 1501, 3).to(DEV)
    names = ["p1", "p2", "w1", "b1", "w2", "b2", "w3", "b3"]
    # The channel that holds a neighbour's maximum is decided by last bits when two channels are within rounding of each other
    # (see the cross test): points with such a neighbour -- found in float64 -- get a zero upstream gradient.
    whole = torch.cat(halves, dim=-1).long()
    nb64 = p2.double()[torch.arange(3, device=DEV).view(3, 1, 1), whole]
    r64 = nb64 - p1.double().unsqueeze(2)
    x64 = torch.cat([r64, r64.norm(dim=-1, keepdim=True)], dim=-1)
    for wi, bi_ in ((ws[0], ws[1]), (ws[2], ws[3]), (ws[4], ws[5])):
        x64 = torch.relu(x64 @ wi.double().to(DEV).T + bi_.double().to(DEV))
    top2 = x64.topk(2, dim=-1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 1e-4 * (1.0 + top2[..., 0])).all(dim=-1)
    assert float(clear.float().mean()) > 0.9
    g = g * clear.unsqueeze(-1).float()

    def grads(fn):
        leaves = [t.detach().clone().to(DEV).requires_grad_(True) for t in (p1, p2, *ws)]
        return torch.autograd.grad(fn(*leaves), leaves, g)
    hip = grads(lambda a, b, *w: be.fusion_mlp(a, b, halves, *w))
    again = grads(lambda a, b, *w: be.fusion_mlp(a, b, halves, *w))
    want = grads(lambda a, b, *w: grad.fusion_twin(be.group_rows, a, b, halves, *w))
    for name, a, a2, b in zip(names, hip, again, want):
        assert torch.equal(a, a2), name
        assert torch.isfinite(a).all(), name
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 2e-4 * scale + 2e-5, f"grad {name}: max err {err:.2e}, gradient scale {scale:.2e}"


@pytest.mark.parametrize("d", [64, 128])
def test_cross_backward_kernel_matches_the_unfused_layer_and_repeats_bit_for_bit(d):
    """mcp_cross_grad (D = 64; D = 128 with its three workgroup roles) against autograd over the unfused layer (grad.cross_twin) on the device: the neighbour list as the two
    searches' 16 + 16 halves -- which overlap, so equal maxima between list positions are the common case and the kernel's
    lowest-position rule must give the same total as autograd's choice -- n1 != n2, a ragged last workgroup; two runs, same bits."""
    from mocopci_amd import grad
    be = ops.backend()
    n1, n2 = 1237, 1500
    xyz1, xyz2 = cloud(140, 3, n1).to(DEV), cloud(141, 3, n2).to(DEV)
    f1, f2 = rnd(142, 3, n1, d).to(DEV), rnd(143, 3, n2, d).to(DEV)
    halves = (be.knn(xyz1, xyz2, 16), be.knn(xyz1, xyz2, 16))           # identical halves: every maximum is tied between two positions
    w = [rnd(144, d, 3, scale=0.3), rnd(145, d, scale=0.1), rnd(146, d, d, scale=d ** -0.5), rnd(147, d, scale=0.1)]
    g0 = rnd(148, 3, n1, d).to(DEV)
    names = ["xyz1", "xyz2", "points1", "points2", "wpos", "bpos", "wmlp", "bmlp"]

    def grads(fn, idx):
        leaves = [t.detach().clone().to(DEV).requires_grad_(True) for t in (xyz1, xyz2, f1, f2, *w)]
        return torch.autograd.grad(fn(*leaves[:4], idx, *leaves[4:]), leaves, g)
    g = None
    bi = torch.arange(3, device=DEV).view(3, 1, 1)
    for idx in (halves, be.knn(xyz1, xyz2, 32)):                        # ... and one (B,N1,32) list of distinct neighbours
        # Which neighbour holds a channel's maximum is decided by z's last bits when two DISTINCT neighbours are within rounding of
        # each other, and the kernel's z (its forward's bits) and torch's differ there: either choice is a valid subgradient, but
        # the two gradients then differ by O(1) in that (point, channel); the same holds where a value sits on a LeakyReLU kink.
        # Those pairs -- found in float64 -- get a zero upstream gradient, so the comparison stays strict everywhere else.
        uniq = (idx[0] if isinstance(idx, tuple) else idx).long()
        wd = [t.double().to(DEV) for t in w]
        u64 = f2.double()[bi, uniq] + f1.double().unsqueeze(2) + (xyz2.double()[bi, uniq] - xyz1.double().unsqueeze(2)) @ wd[0].T + wd[1]
        top2 = (F.leaky_relu(u64, 0.1) @ wd[2].T + wd[3]).topk(2, dim=2).values
        clear = ((top2[:, :, 0] - top2[:, :, 1]) > 1e-4 * (1.0 + top2[:, :, 0].abs())) & (top2[:, :, 0].abs() > 1e-5)   # ... or sits on LeakyReLU's kink
        clear &= (u64.abs().amin(dim=(2, 3)) > 1e-5).unsqueeze(-1)        # a point with some u_j[k] on the kink: LeakyReLU'(u) is 1 or 0.1 by rounding
        g = g0 * clear.float()
        assert float(clear.float().mean()) > 0.9
        hip = grads(be.cross_layer, idx)
        again = grads(be.cross_layer, idx)
        want = grads(lambda a, b, c, e, i, *ww: grad.cross_twin(be.group_rows, a, b, c, e, i, *ww), idx)
        for name, a, a2, b in zip(names, hip, again, want):
            assert torch.equal(a, a2), name
            assert torch.isfinite(a).all(), name
            scale = float(b.abs().max())
            err = float((a - b).abs().max())
            assert err <= 2e-4 * scale + 2e-5, f"grad {name}: max err {err:.2e}, gradient scale {scale:.2e}"


@pytest.mark.parametrize("n,s,d", [(1500, 1500, 32), (2000, 501, 64), (600, 150, 128)])
def test_pointconv_backward_kernel_matches_the_unfused_layer_and_repeats_bit_for_bit(n, s, d):
    """mcp_pointconv_agg_grad against autograd over the unfused layer (grad.pointconv_agg_twin) on the device, odd centre counts (the
    last wave holds one centre), PointConv (s = n) and PointConvD (s < n) forms; two runs give identical bits."""
    from mocopci_amd import grad
    be = ops.backend()
    s_xyz = cloud(150 + d, 3, n).to(DEV)
    new_xyz = s_xyz[:, :s].clone()
    pts = rnd(151, 3, n, d).to(DEV)
    idx = be.knn(new_xyz, s_xyz, 32)
    wn = [rnd(152, 8, 3, scale=0.5), rnd(153, 8, scale=0.1), rnd(154, 8, 8, scale=0.4), rnd(155, 8, scale=0.1), rnd(156, 8, 8, scale=0.4), rnd(157, 8, scale=0.1)]
    g = rnd(158, 3, s, (d + 3) * 8).to(DEV)
    names = ["s_xyz", "new_xyz", "points", "w0", "b0", "w1", "b1", "w2", "b2"]

    def grads(fn):
        leaves = [t.detach().clone().to(DEV).requires_grad_(True) for t in (s_xyz, new_xyz, pts, *wn)]
        return torch.autograd.grad(fn(*leaves), leaves, g)
    hip = grads(lambda a, b, c, *w: be.pointconv_agg(a, b, c, idx, *w))
    again = grads(lambda a, b, c, *w: be.pointconv_agg(a, b, c, idx, *w))
    want = grads(lambda a, b, c, *w: grad.pointconv_agg_twin(be.group_rows, a, b, c, idx, *w))
    for name, a, a2, b in zip(names, hip, again, want):
        assert torch.equal(a, a2), name
        assert torch.isfinite(a).all(), name
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 2e-4 * scale + 2e-5, f"grad {name}: max err {err:.2e}, gradient scale {scale:.2e}"


def test_fusion_on_batch_statistics_matches_autograd_over_the_unfused_layer_and_repeats_bit_for_bit():
    """The fusion layer in net.train() mode (csrc/fusion_bn.hip: three statistics passes + the layer; four backward passes with the
    BatchNorm mean terms) against float64 autograd over the unfused layer with nn.BatchNorm semantics (batch statistics, biased
    variance, eps 1e-3): output, batch statistics, and the gradients w.r.t. coordinates, conv weights and BatchNorm weight / bias.
    Points with a neighbour whose arg-max channel or a ReLU is decided by rounding (found in float64) get a zero upstream gradient."""
    be = ops.backend()
    B, N = 2, 701
    p1 = cloud(160, B, N).to(DEV)
    p2 = p1 + rnd(161, B, N, 3, scale=0.2).to(DEV)
    halves = (be.knn(p1, p1, 32), be.knn(p1, p2, 32))
    conv = [t.to(DEV) for t in (rnd(162, 64, 4, scale=0.5), rnd(163, 64, scale=0.3), rnd(164, 64, 64, scale=0.125), rnd(165, 64, scale=0.3),
                                rnd(166, 128, 64, scale=0.125), rnd(167, 128, scale=0.3))]
    aff = [t.to(DEV) for t in (1 + rnd(168, 64, scale=0.2), rnd(169, 64, scale=0.2), 1 + rnd(170, 64, scale=0.2), rnd(171, 64, scale=0.2),
                               1 + rnd(172, 128, scale=0.2), rnd(173, 128, scale=0.2))]
    whole = torch.cat(halves, dim=-1).long()
    bi = torch.arange(B, device=DEV).view(B, 1, 1)

    def unfused(a, b, *wa):  # float64, mocopci.py:803-819 with BatchNorm2d in training mode
        w, af = wa[:6], wa[6:]
        nb = b[bi, whole]
        r = nb - a.unsqueeze(2)
        x = torch.cat([r, r.norm(dim=-1, keepdim=True)], dim=-1)
        pre = []
        for i in range(3):
            z = x @ w[2 * i].T + w[2 * i + 1]
            flat = z.reshape(-1, z.shape[-1])
            mean, var = flat.mean(0), flat.var(0, unbiased=False)
            v = (z - mean) * (af[2 * i] * torch.rsqrt(var + 1e-3)) + af[2 * i + 1]
            pre.append((v, mean, var))
            x = torch.relu(v)
        wgt = torch.softmax(x.max(dim=-1)[0], dim=-1)
        return torch.sum(wgt.unsqueeze(-1) * nb, dim=2), pre
    leaves64 = [t.detach().double().clone().requires_grad_(True) for t in (p1, p2, *conv, *aff)]
    want_out, pre = unfused(*leaves64)
    top2 = torch.relu(pre[2][0]).topk(2, dim=-1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 1e-4 * (1.0 + top2[..., 0])).all(dim=-1)
    clear &= (pre[0][0].abs().amin(dim=(2, 3)) > 1e-5) & (pre[1][0].abs().amin(dim=(2, 3)) > 1e-5)
    assert float(clear.float().mean()) > 0.8
    g = rnd(174, B, N, 3).to(DEV) * clear.unsqueeze(-1).float()
    want = torch.autograd.grad(want_out, leaves64, g.double())

    def run():
        leaves = [t.detach().clone().requires_grad_(True) for t in (p1, p2, *conv, *aff)]
        out, bn, var = be.fusion_bn(leaves[0], leaves[1], halves, leaves[2:8], leaves[8:], 1e-3)
        return out, bn, var, torch.autograd.grad(out, leaves, g)
    out, bn, var, got = run()
    out2, bn2, var2, got2 = run()
    assert torch.equal(out, out2) and torch.equal(bn, bn2) and torch.equal(var, var2)
    torch.testing.assert_close(out.double(), want_out.detach(), rtol=1e-4, atol=1e-4)
    off = 0
    for i, c in enumerate((64, 64, 128)):
        torch.testing.assert_close(bn[4 * off:4 * off + c].double(), pre[i][1].detach(), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(var[off:off + c].double(), pre[i][2].detach(), rtol=1e-4, atol=1e-6)
        off += c
    names = ["p1", "p2", "w1", "b1", "w2", "b2", "w3", "b3", "gamma1", "beta1", "gamma2", "beta2", "gamma3", "beta3"]
    for name, a, a2, b in zip(names, got, got2, want):
        assert torch.equal(a, a2), name
        assert torch.isfinite(a).all(), name
        scale = float(b.abs().max())
        err = float((a.double() - b).abs().max())
        floor = 2e-5 if not name.startswith("b") or name.startswith("beta") else 1e-3   # conv biases: exactly 0 here, rounding noise in autograd
        assert err <= 5e-4 * scale + floor, f"grad {name}: max err {err:.2e}, gradient scale {scale:.2e}"


def test_fusion_batch_statistics_of_a_channel_far_from_its_bias():
    """A layer-2 conv whose outputs sit ~100 standard deviations away from the conv bias (positive weights on inputs that layer 1's
    BatchNorm pins near 1): the variance comes from S2 / R - (S1 / R)^2 around the bias, so the cancellation eats ~1e-7 * 100^2 of
    it (with the workgroup partials added in fp32 it was the number of workgroups times more).  Against float64."""
    be = ops.backend()
    B, N = 4, 4096
    p1 = cloud(180, B, N).to(DEV)
    p2 = p1 + rnd(181, B, N, 3, scale=0.2).to(DEV)
    halves = (be.knn(p1, p1, 32), be.knn(p1, p2, 32))
    conv = [t.to(DEV) for t in (rnd(182, 64, 4, scale=0.5), rnd(183, 64, scale=0.3), rnd(184, 64, 64, scale=0.125).abs(), rnd(185, 64, scale=0.3),
                                rnd(186, 128, 64, scale=0.125), rnd(187, 128, scale=0.3))]
    aff = [t.to(DEV) for t in (torch.full((64,), 0.06), torch.ones(64), 1 + rnd(170, 64, scale=0.2), rnd(171, 64, scale=0.2),
                               1 + rnd(172, 128, scale=0.2), rnd(173, 128, scale=0.2))]
    out, bn, var = be.fusion_bn_forward(p1, p2, halves, conv, aff, 1e-3)
    whole = torch.cat(halves, dim=-1).long()
    bi = torch.arange(B, device=DEV).view(B, 1, 1)
    nb = p2.double()[bi, whole]
    r = nb - p1.double().unsqueeze(2)
    x = torch.cat([r, r.norm(dim=-1, keepdim=True)], dim=-1)
    z1 = (x @ conv[0].double().T + conv[1].double()).reshape(-1, 64)
    v1 = torch.relu((z1 - z1.mean(0)) * (aff[0].double() * torch.rsqrt(z1.var(0, unbiased=False) + 1e-3)) + aff[1].double())
    z2 = v1 @ conv[2].double().T + conv[3].double()
    mean2, var2 = z2.mean(0), z2.var(0, unbiased=False)
    ratio = ((mean2 - conv[3].double()).abs() / var2.sqrt())
    assert float(ratio.min()) > 30, float(ratio.min())     # the case is what it claims to be
    torch.testing.assert_close(bn[256:320].double(), mean2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(var[64:128].double(), var2, rtol=1e-2, atol=0)
    assert torch.isfinite(out).all()


def test_fusion_batch_statistics_forward_that_keeps_the_scores():
    """mcp_fusion_bn_forward_save / mcp_fusion_bn_backward_saved: the same out / bn / var as mcp_fusion_bn_forward and the same gradients
    as mcp_fusion_bn_backward, bit for bit -- the kept (score, arg-max channel, zhat3) rows are what the backward's first pass recomputes."""
    import ctypes
    from mocopci_amd import _lib
    lib, be = _lib.load(), ops.backend()
    B, N = 2, 700
    p1 = cloud(190, B, N).to(DEV)
    p2 = p1 + rnd(191, B, N, 3, scale=0.2).to(DEV)
    ia, ib = be.knn(p1, p1, 32), be.knn(p1, p2, 32)
    conv = [t.to(DEV) for t in (rnd(192, 64, 4, scale=0.5), rnd(193, 64, scale=0.1), rnd(194, 64, 64, scale=0.125), rnd(195, 64, scale=0.1),
                                rnd(196, 128, 64, scale=0.125), rnd(197, 128, scale=0.1))]
    aff = [t.to(DEV) for t in (1 + rnd(160, 64, scale=0.2), rnd(161, 64, scale=0.2), 1 + rnd(162, 64, scale=0.2), rnd(163, 64, scale=0.2),
                               1 + rnd(164, 128, scale=0.2), rnd(165, 128, scale=0.2))]
    rows = B * N * 64
    saved = (torch.empty(rows, dtype=torch.int32, device=DEV), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV))
    out0, bn0, var0 = be.fusion_bn_forward(p1, p2, (ia, ib), conv, aff, 1e-3)
    out1, bn1, var1 = be.fusion_bn_forward(p1, p2, (ia, ib), conv, aff, 1e-3, saved=saved)
    assert torch.equal(out0, out1) and torch.equal(bn0, bn1) and torch.equal(var0, var1)
    assert int(saved[0].min()) >= 0 and int(saved[0].max()) < 128 and bool((saved[2] >= 0).all())
    g = rnd(198, B, N, 3).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    need = lib.mcp_fusion_bn_grad_workspace_bytes(B, N)

    def run(entry, extra):
        scr = dict(row_c=torch.empty(rows, dtype=torch.int32, device=DEV), row_dy=torch.empty(rows, device=DEV), row_a=torch.empty(rows, device=DEV),
                   dy2=torch.empty(rows, 64, device=DEV), dy1=torch.empty(rows, 64, device=DEV), d_p1=torch.empty_like(p1),
                   d_nb=torch.empty(B, N, 64, 3, device=DEV), d_w=torch.empty(lib.mcp_fusion_grad_floats(), device=DEV), d_aff=torch.empty(512, device=DEV),
                   ws=torch.empty(need, dtype=torch.uint8, device=DEV))
        rc = getattr(lib, entry)(B, N, 64, P(p1), P(p2), P(ia), P(ib), *[P(t) for t in conv], P(bn0), P(g), *extra, P(scr["row_c"]), P(scr["row_dy"]),
                                 P(scr["row_a"]), P(scr["dy2"]), P(scr["dy1"]), P(scr["d_p1"]), P(scr["d_nb"]), P(scr["d_w"]), P(scr["d_aff"]), P(scr["ws"]), need, st)
        assert rc == 0
        torch.cuda.synchronize()
        return scr
    a = run("mcp_fusion_bn_backward", ())
    b = run("mcp_fusion_bn_backward_saved", tuple(P(t) for t in saved))
    for k in ("row_c", "row_dy", "row_a", "d_p1", "d_nb", "d_w", "d_aff"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("heads,hd,nq,nk,bf", [(8, 8, 333, 517, 3), (8, 8, 2048, 2048, 2), (4, 16, 200, 64, 2), (8, 16, 130, 1000, 1)])
def test_attention_small_backward_kernels_match_the_dense_formulation_and_repeat(heads, hd, nq, nk, bf):
    """mcp_attention_small_grad against float64 autograd over softmax(q k^T scale) v: ragged query / key counts (partial tiles on both
    sides), both head widths; two runs give identical bits."""
    be = ops.backend()
    C = heads * hd
    q, kv, g = rnd(200, bf, nq, C).to(DEV), rnd(201, bf, nk, 2 * C).to(DEV), rnd(202, bf, nq, C).to(DEV)

    def grads(fn, dt):
        leaves = [t.detach().to(dt).clone().requires_grad_(True) for t in (q, kv)]
        return torch.autograd.grad(fn(*leaves), leaves, g.to(dt))

    def dense(a, b):
        qh = a.reshape(bf, nq, heads, hd).permute(0, 2, 1, 3)
        kvh = b.reshape(bf, nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
        p = torch.softmax(qh @ kvh[0].transpose(-2, -1) * hd ** -0.5, dim=-1)
        return (p @ kvh[1]).permute(0, 2, 1, 3).reshape(bf, nq, C)
    hip = grads(lambda a, b: be.attention(a, b, heads), torch.float32)
    again = grads(lambda a, b: be.attention(a, b, heads), torch.float32)
    want = grads(dense, torch.float64)
    for name, a, a2, b in zip(("q", "kv"), hip, again, want):
        assert torch.equal(a, a2), name
        scale = float(b.abs().max())
        err = float((a.double() - b).abs().max())
        assert err <= 1e-4 * scale + 1e-6, f"grad {name}: max err {err:.2e}, gradient scale {scale:.2e}"


def test_attention_dropout_kernels_match_the_dense_formulation_under_the_same_mask():
    """mcp_attention_small_dropout and its backward: the mask is a counter-based hash of (seed, batch, head, query, key); the test
    rebuilds it with the same integer arithmetic in torch and compares output and gradients with float64 autograd over
    dropout(softmax(q k^T scale)) v under that mask; the kept fraction matches the rate; no gradient leaks through dropped entries."""
    be = ops.backend()
    bf, heads, hd, nq, nk, p, seed = 2, 8, 8, 301, 450, 0.25, 123457
    C = heads * hd
    q, kv, g = rnd(210, bf, nq, C).to(DEV), rnd(211, bf, nk, 2 * C).to(DEV), rnd(212, bf, nq, C).to(DEV)
    M = 0xFFFFFFFF
    row = (torch.arange(bf * heads * nq, device=DEV, dtype=torch.int64).view(bf, heads, nq, 1))
    key = torch.arange(nk, device=DEV, dtype=torch.int64).view(1, 1, 1, nk)
    x = (seed ^ ((row * 0x9E3779B1) & M) ^ ((key * 0x85EBCA77) & M)) & M
    x = x ^ (x >> 16); x = (x * 0x7FEB352D) & M; x = x ^ (x >> 15); x = (x * 0x846CA68B) & M; x = x ^ (x >> 16)
    keep = (x >= int(p * 4294967296.0)).double() / (1.0 - p)                      # (bf, heads, nq, nk)
    assert abs(float((keep > 0).double().mean()) - (1.0 - p)) < 5e-3

    def dense(a, b):
        qh = a.reshape(bf, nq, heads, hd).permute(0, 2, 1, 3)
        kvh = b.reshape(bf, nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
        pm = torch.softmax(qh @ kvh[0].transpose(-2, -1) * hd ** -0.5, dim=-1) * keep
        return (pm @ kvh[1]).permute(0, 2, 1, 3).reshape(bf, nq, C)
    l64 = [t.detach().double().clone().requires_grad_(True) for t in (q, kv)]
    want_out = dense(*l64)
    want = torch.autograd.grad(want_out, l64, g.double())
    l32 = [t.detach().clone().requires_grad_(True) for t in (q, kv)]
    out = ops._AttentionSmallFn.apply(be, l32[0], l32[1], heads, hd ** -0.5, p, seed)
    got = torch.autograd.grad(out, l32, g)
    torch.testing.assert_close(out.double(), want_out.detach(), rtol=1e-4, atol=1e-5)
    for name, a, b in zip(("q", "kv"), got, want):
        scale = float(b.abs().max())
        err = float((a.double() - b).abs().max())
        assert err <= 1e-4 * scale + 1e-6, f"grad {name}: max err {err:.2e}, gradient scale {scale:.2e}"
    # through the public entry: reproducible under torch.manual_seed, different across seeds
    torch.manual_seed(5); o1 = be.attention(q, kv, heads, dropout_p=p)
    torch.manual_seed(5); o2 = be.attention(q, kv, heads, dropout_p=p)
    torch.manual_seed(6); o3 = be.attention(q, kv, heads, dropout_p=p)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)


@pytest.mark.parametrize("shape", [(3, 1000, 256), (7, 33), (4099,)])
def test_prelu_dropout_kernels_match_torch_under_the_same_mask(shape):
    """mcp_prelu_dropout and its backward (Mlp_T's act + drop under net.train(), mocopci.py:1561-1562): the mask is a counter-based hash
    of (seed, element index), rebuilt here with the same integer arithmetic; output is exact, dz exact, the slope gradient within fp32
    summation error of float64 and the same bits on every call; kept fraction = 1 - p."""
    be = ops.backend()
    p, seed, M = 0.25, 987654321, 0xFFFFFFFF
    z, g = rnd(410, *shape).to(DEV), rnd(411, *shape).to(DEV)
    slope = torch.tensor([0.37], device=DEV)
    i = torch.arange(z.numel(), device=DEV, dtype=torch.int64)
    x = (seed ^ ((i * 0x9E3779B1) & M)) & M          # element indices below 2^32: the high half contributes 0
    x = x ^ (x >> 16); x = (x * 0x7FEB352D) & M; x = x ^ (x >> 15); x = (x * 0x846CA68B) & M; x = x ^ (x >> 16)
    keep = (x >= int(p * 4294967296.0)).reshape(shape)
    if z.numel() > 100000:
        assert abs(float(keep.double().mean()) - (1.0 - p)) < 5e-3
    inv = torch.tensor(1.0 / (1.0 - p), dtype=torch.float64).float().to(DEV)
    zl, sl = z.clone().requires_grad_(True), slope.clone().requires_grad_(True)
    out = ops._PreluDropFn.apply(zl, sl, p, seed)
    want_out = torch.where(z > 0, z, slope * z) * torch.where(keep, inv, torch.zeros_like(inv))
    assert torch.equal(out, want_out)
    dz, da = torch.autograd.grad(out, (zl, sl), g)
    gm = g * torch.where(keep, inv, torch.zeros_like(inv))
    assert torch.equal(dz, torch.where(z > 0, gm, slope * gm))
    want_da = float((gm.double() * z.double().clamp(max=0)).sum())
    assert da.shape == slope.shape and abs(float(da) - want_da) <= 1e-5 * float((gm.double() * z.double().clamp(max=0)).abs().sum()) + 1e-6
    out2 = ops._PreluDropFn.apply(zl, sl, p, seed)
    assert torch.equal(torch.autograd.grad(out2, (sl,), g)[0], da)                     # fixed summation order
    assert not (dz[~keep] != 0).any()                                                   # nothing leaks through dropped elements
    torch.manual_seed(5); o1 = be.prelu_dropout(z, slope, p)
    torch.manual_seed(5); o2 = be.prelu_dropout(z, slope, p)
    torch.manual_seed(6); o3 = be.prelu_dropout(z, slope, p)
    assert torch.equal(o1, o2) and (z.numel() < 64 or not torch.equal(o1, o3))
    assert torch.equal(ops._PreluDropFn.apply(z, slope, 0.0, 1), torch.where(z > 0, z, slope * z))   # p = 0: plain PReLU


@pytest.mark.parametrize("hd,nq,nk", [(8, 301, 450), (16, 128, 77), (8, 2048, 2048)])
def test_attention_forward_that_keeps_the_log_sum_exp(hd, nq, nk):
    """mcp_attention_small_lse (the training forward): at drop_p = 0 the output of mcp_attention_small bit for bit; its log-sum-exp is the
    statistics pass's (mcp_attention_small_grad recomputes it) -- the two backward entry points return the same gradients bit for bit,
    with and without dropout; lse against float64."""
    import ctypes
    from mocopci_amd import _lib
    lib, be = _lib.load(), ops.backend()
    bf, heads = 2, 8
    C = heads * hd
    q, kv, g = rnd(500, bf, nq, C).to(DEV), rnd(501, bf, nk, 2 * C).to(DEV), rnd(502, bf, nq, C).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    Pk, Pv = ctypes.c_void_p(kv.data_ptr()), ctypes.c_void_p(kv.data_ptr() + 4 * C)
    scale = hd ** -0.5
    for p_drop, seed in ((0.0, 0), (0.25, 4242)):
        out, lse = torch.empty_like(q), torch.empty(bf, heads, nq, device=DEV)
        assert lib.mcp_attention_small_lse(bf, nq, nk, heads, hd, P(q), C, Pk, 2 * C, Pv, 2 * C, scale, p_drop, seed, P(out), P(lse), st) == 0
        if p_drop == 0.0:
            assert torch.equal(out, be._attention(q, kv, heads, scale))
            qh = q.double().reshape(bf, nq, heads, hd).permute(0, 2, 1, 3)
            kh = kv.double().reshape(bf, nk, 2, heads, hd)[:, :, 0].permute(0, 2, 1, 3)
            want = torch.logsumexp(qh @ kh.transpose(-2, -1) * scale, dim=-1) / 0.6931471805599453      # log2 domain
            torch.testing.assert_close(lse.double(), want, rtol=1e-5, atol=1e-5)
        else:
            ref = torch.empty_like(q)
            assert lib.mcp_attention_small_dropout(bf, nq, nk, heads, hd, P(q), C, Pk, 2 * C, Pv, 2 * C, scale, p_drop, seed, P(ref), st) == 0
            assert torch.equal(out, ref)
        need = lib.mcp_attention_small_grad_workspace_bytes(bf, nq, heads)
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        dq1, dkv1, dq2, dkv2 = torch.empty_like(q), torch.empty_like(kv), torch.empty_like(q), torch.empty_like(kv)
        assert lib.mcp_attention_small_grad(bf, nq, nk, heads, hd, P(q), C, Pk, 2 * C, Pv, 2 * C, scale, p_drop, seed, P(out), P(g), P(dq1), P(dkv1), P(ws), need, st) == 0
        torch.cuda.synchronize()
        assert lib.mcp_attention_small_grad_lse(bf, nq, nk, heads, hd, P(q), C, Pk, 2 * C, Pv, 2 * C, scale, p_drop, seed, P(out), P(g), P(lse), P(dq2), P(dkv2),
                                                P(ws), need, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(dq1, dq2) and torch.equal(dkv1, dkv2), p_drop


def test_ptblock_backward_kernel_matches_the_unfused_block_and_repeats_bit_for_bit():
    """mcp_ptblock_grad against autograd over the unfused block (grad.ptblock_twin) on the device: q, k, v as slices of one packed
    projection (row stride 192), an odd point count (the last wave holds one point); two runs give identical bits.  Points with a
    ReLU input on the kink (found in float64) get a zero upstream gradient."""
    from mocopci_amd import grad
    be = ops.backend()
    B, n = 3, 1001
    xyz = cloud(180, B, n).to(DEV)
    qkv = rnd(181, B, n, 192).to(DEV)
    idx = be.knn(xyz, xyz, 16)
    ws = [rnd(182, 64, 3, scale=0.3), rnd(183, 64, scale=0.1)]
    for i in range(3):
        ws += [rnd(184 + i, 64, 64, scale=0.125), rnd(188 + i, 64, scale=0.1)]
    ws = [t.to(DEV) for t in ws]
    bi = torch.arange(B, device=DEV).view(B, 1, 1)
    x64, q64 = xyz.double(), qkv.double()
    d1 = (x64.unsqueeze(2) - x64[bi, idx.long()]) @ ws[0].double().T + ws[1].double()
    gpre = (q64[..., :64].unsqueeze(2) - q64[..., 64:128][bi, idx.long()]) + torch.relu(d1) @ ws[2].double().T + ws[3].double()
    a1 = gpre @ ws[4].double().T + ws[5].double()
    clear = (d1.abs().amin(dim=(2, 3)) > 1e-5) & (a1.abs().amin(dim=(2, 3)) > 1e-5)
    assert float(clear.float().mean()) > 0.8
    g = rnd(192, B, n, 64).to(DEV) * clear.unsqueeze(-1).float()
    names = ["xyz", "qkv", "wd1", "bd1", "wd2", "bd2", "wg1", "bg1", "wg2", "bg2"]

    def grads(fn):
        leaves = [t.detach().clone().requires_grad_(True) for t in (xyz, qkv, *ws)]
        x, p = leaves[0], leaves[1]
        return torch.autograd.grad(fn(x, p[..., :64], p[..., 64:128], p[..., 128:], leaves[2:]), leaves, g)
    hip = grads(lambda x, a, b, c, w: be.ptblock_layer(x, a, b, c, idx, w))
    again = grads(lambda x, a, b, c, w: be.ptblock_layer(x, a, b, c, idx, w))
    want = grads(lambda x, a, b, c, w: grad.ptblock_twin(be.group_rows, x, a, b, c, idx, *w))
    for name, a, a2, b in zip(names, hip, again, want):
        assert torch.equal(a, a2), name
        assert torch.isfinite(a).all(), name
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 2e-4 * scale + 2e-5, f"grad {name}: max err {err:.2e}, gradient scale {scale:.2e}"


def test_ptblock_gradients():
    n = 333
    xyz = cloud(40, 2, n)
    q, k, v = rnd(41, 2, n, 64), rnd(42, 2, n, 64), rnd(43, 2, n, 64)
    idx = orc.knn(xyz, xyz, 16, mode=1)
    ws = [rnd(44, 64, 3, scale=0.3), rnd(45, 64, scale=0.1)]
    for i in range(3):
        ws += [rnd(46 + i, 64, 64, scale=0.125), rnd(50 + i, 64, scale=0.1)]
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda x, a, b, c, *w: be.ptblock_layer(x, a, b, c, idx.to(DEV), list(w)), lambda x, a, b, c, *w: ob.ptblock_layer(x, a, b, c, idx, list(w)),
                  [xyz, q, k, v, *ws], names=["xyz", "q", "k", "v", "wd1", "bd1", "wd2", "bd2", "wg1", "bg1", "wg2", "bg2"])


@pytest.mark.parametrize("heads,hd,nq,nk", [(8, 8, 300, 500), (8, 32, 256, 256), (3, 256, 64, 100)])
def test_attention_gradients(heads, hd, nq, nk):
    C = heads * hd
    q, kv = rnd(60, 2, nq, C), rnd(61, 2, nk, 2 * C)
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda a, b: be.attention(a, b, heads), lambda a, b: ob.attention(a, b, heads), [q, kv], names=["q", "kv"])


@pytest.mark.parametrize("n,s,c", [(300, 100, 7), (4096, 2048, 16)])
def test_interp3_gradients_reach_features_and_coordinates(n, s, c):
    dense, sparse, feat = cloud(70, 2, n), cloud(71, 2, s), rnd(72, 2, s, c)
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda a, b, f: be.interp3(a, b, f), lambda a, b, f: ob.interp3(a, b, f), [dense, sparse, feat], names=["dense", "sparse", "feat"])


@pytest.mark.parametrize("n,s,c", [(500, 120, 3), (4096, 2048, 64), (900, 40, 35)])
def test_interp3_blend_backward_kernels_match_float64_and_repeat(n, s, c):
    """mcp_interp3_apply_grad_sorted (UpsampleFlow / PointWarping's blend, mocopci.py:1480-1481): both gradients against float64 autograd
    over the unfused blend, the same bits on every call, and against the unfused twin's autograd on the device; either output alone."""
    from mocopci_amd import grad as G
    be = ops.backend()
    gen = torch.Generator().manual_seed(n + c)
    feat, w3, g = rnd(300, 2, s, c).to(DEV), torch.rand(2, n, 3, generator=gen).to(DEV), rnd(301, 2, n, c).to(DEV)
    idx3 = torch.randint(0, s, (2, n, 3), generator=gen, dtype=torch.int32)
    idx3[:, : n // 3] = idx3[:, : n // 3] % 2            # two rows that a third of the points blend from
    idx3 = idx3.to(DEV)
    leaves = [feat.clone().requires_grad_(True), w3.clone().requires_grad_(True)]
    got = torch.autograd.grad(ops._Interp3ApplyFn.apply(leaves[0], idx3, leaves[1]), leaves, g)
    again = torch.autograd.grad(ops._Interp3ApplyFn.apply(leaves[0], idx3, leaves[1]), leaves, g)
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    l64 = [feat.double().clone().requires_grad_(True), w3.double().clone().requires_grad_(True)]
    bidx = torch.arange(2, device=DEV).view(2, 1, 1)
    want = torch.autograd.grad((l64[1].unsqueeze(-1) * l64[0][bidx, idx3.long()]).sum(2), l64, g.double())
    for name, a, b in zip(("feat", "w3"), got, want):
        scale = float(b.abs().max())
        assert float((a.double() - b).abs().max()) <= 2e-6 * scale * max(1.0, (n / s) ** 0.5), name
    tw = [feat.clone().requires_grad_(True), w3.clone().requires_grad_(True)]
    twin = torch.autograd.grad(G.interp3_apply_twin(be.group_rows, tw[0], idx3, tw[1]), tw, g)
    for a, b in zip(got, twin):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    only_w = torch.autograd.grad(ops._Interp3ApplyFn.apply(feat, idx3, leaves[1]), [leaves[1]], g)[0]
    only_f = torch.autograd.grad(ops._Interp3ApplyFn.apply(leaves[0], idx3, w3), [leaves[0]], g)[0]
    assert torch.equal(only_w, got[1]) and torch.equal(only_f, got[0])


@pytest.mark.parametrize("cin,hidden,cout", [(64, 256, 64), (128, 512, 3)])  # both heads of the two levels
def test_mlp2_gradients(cin, hidden, cout):
    x, res = rnd(90, 2, 300, cin), rnd(91, 2, 300, cout)
    w = [rnd(92, hidden, cin, scale=cin ** -0.5), rnd(93, hidden, scale=0.1), rnd(94, cout, hidden, scale=hidden ** -0.5), rnd(95, cout, scale=0.1)]
    slope = torch.tensor([0.25])
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda a, r, w1, b1, w2, b2, sl: be.mlp2(a, w1, b1, w2, b2, sl, res=r), lambda a, r, w1, b1, w2, b2, sl: ob.mlp2(a, w1, b1, w2, b2, sl, res=r),
                  [x, res, *w, slope], names=["x", "res", "w1", "b1", "w2", "b2", "slope"])


@pytest.mark.parametrize("slope,res_scale", [(0.1, 1.0), (0.0, 1e6)])
def test_linear_gradients(slope, res_scale):
    """(0.0, 1e6): ReLU with residuals so large that act(z) vanishes in y = act(z) + res for most entries -- the mask must still be
    the sign of z (the reference's autograd), not of y - res."""
    x, res = rnd(96, 20000, 64), rnd(97, 20000, 32, scale=res_scale)
    w, b = rnd(98, 32, 64, scale=0.125), rnd(99, 32, scale=0.1)
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda a, r, ww, bb: be.linear(a, ww, bb, slope, r), lambda a, r, ww, bb: ob.linear(a, ww, bb, slope, r), [x, res, w, b],
                  names=["x", "res", "w", "b"])


@pytest.mark.parametrize("rows,n,k", [(20000, 64, 32), (16384, 32, 64), (40001, 128, 192), (33000, 256, 64), (70, 64, 64), (50000, 3, 32), (30000, 32, 3), (9000, 40, 67)])
def test_linear_weight_gradient_kernel_matches_float64_and_repeats(rows, n, k):
    """mcp_linear_wgrad (csrc/linear_grad.hip): dW = gz^T x and db = column sums of gz with the rows on the MFMA's contraction axis,
    against float64; ragged row counts (partial last stage), every tiles-per-wave variant; two runs give the same bits."""
    from mocopci_amd import _lib
    lib = _lib.load()
    gz, x = rnd(300, rows, n).to(DEV), rnd(301, rows, k).to(DEV)
    need = lib.mcp_linear_wgrad_workspace_bytes(rows, n, k)
    assert need > 0
    outs = []
    for _ in range(2):
        dw, db = torch.empty((n, k), device=DEV), torch.empty((n,), device=DEV)
        ws = torch.empty((need,), dtype=torch.uint8, device=DEV)
        ops._call("mcp_linear_wgrad", gz, rows, n, k, _lib.fptr(gz), n, _lib.fptr(x), k, _lib.fptr(dw), _lib.fptr(db), ws.data_ptr(), need)
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    want_w, want_b = gz.double().t() @ x.double(), gz.double().sum(0)
    scale = float(want_w.abs().max())
    assert float((outs[0][0].double() - want_w).abs().max()) <= 2e-6 * scale * (rows ** 0.5) / 100 + 1e-4 * scale
    torch.testing.assert_close(outs[0][1].double(), want_b, rtol=1e-4, atol=1e-3)
    assert lib.mcp_linear_wgrad_workspace_bytes(1000, 512, 32) == 0     # shapes the kernel does not take report 0: the caller keeps the GEMM


@pytest.mark.parametrize("n,m", [(500, 700), (4096, 2048)])
def test_chamfer_gradients(n, m):
    """The explicit backward (ops._ChamferFn: search distances forward, direct term minus the other direction's scattered term backward)
    against torch autograd over the oracle backend's dense form; per-sample values; a ground truth that asks for no gradient; the
    same bits on every call; and against autograd over the unfused twin on the device."""
    x, y = cloud(80, 2, n), cloud(81, 2, m)
    ob, be = OracleBackend(), ops.backend()
    compare_grads(lambda a, b: be.chamfer(a, b), lambda a, b: ob.chamfer(a, b), [x, y], names=["x", "y"])
    w = torch.tensor([0.3, 1.7])
    compare_grads(lambda a, b: (be.chamfer(a, b, per_sample=True) * w.to(DEV)).sum(), lambda a, b: (ob.chamfer(a, b, per_sample=True) * w).sum(), [x, y],
                  names=["x", "y"])
    xd, yd = x.to(DEV).requires_grad_(True), y.to(DEV)
    g1 = torch.autograd.grad(be.chamfer(xd, yd), xd)[0]
    g2 = torch.autograd.grad(be.chamfer(xd, yd), xd)[0]
    assert torch.equal(g1, g2)
    be.EXPLICIT_CHAMFER_GRAD = False
    try:
        g3 = torch.autograd.grad(be.chamfer(xd, yd), xd)[0]
    finally:
        be.EXPLICIT_CHAMFER_GRAD = True
    torch.testing.assert_close(g1, g3, rtol=1e-4, atol=1e-6 * float(g3.abs().max()) + 1e-9)


@pytest.mark.parametrize("module_mode", ["eval", "train"])
def test_training_gradients_match_the_references_own_autograd_on_gpu(module_mode, capsys):
    """The HIP path's training iteration (forward(train=True), train.py:135-160's objective, backward through the hand-written
    backward kernels) against the gradients of the REFERENCE'S OWN autograd (tests/golden/train_grad_b1_n1024.npz: loss parts,
    per-parameter gradient norms, ~50 whole gradient tensors across encoder / cost volumes / fusion / attention)."""
    def report(msg):
        with capsys.disabled():
            print("\n" + msg)
    hc.run_train_grad_check(DEV, module_mode, report)


@pytest.mark.parametrize("module_mode", ["eval", "train"])
def test_one_training_step_matches_the_oracle_backend(module_mode):
    """module_mode "train": after net.train() (batch-statistics BatchNorm, dropout rates set to 0 so both runs are deterministic);
    "eval": the inference graph differentiated.
    MoCoPCI.forward(train=True) (mocopci.py:1076-1097) + the multi-scale Chamfer objective of train.py:135-160, N=1024, one
    sequence: loss within 1e-4 relative of the CPU oracle-backend run of the same graph, parameter gradients aligned (cosine
    >= 0.999 and norms within 1 %: single near-tie neighbour flips move individual gradient entries), every parameter the
    reference's forward uses receives a gradient, and an SGD step on the HIP side lowers the loss."""
    x1, x2, gt = synth.make_batch(1, 1, 1024)
    gtc = [g.transpose(1, 2).contiguous() for g in gt]
    def build(device):
        n = hc.build_model(device)
        if module_mode == "train":
            n.train()
            n.drop_rate = n.attn_drop_rate = n.drop_path_rate = 0.0
        return n
    cpu_net = build("cpu")
    prev = ops.set_backend(OracleBackend())
    try:
        out_c = cpu_net(x1, x2, gtc, None, True)
        loss_c, _ = training.multiscale_loss(*out_c, gtc)
        loss_c.backward()
    finally:
        ops.set_backend(prev)
    net = build(DEV)
    to = lambda ts: [t.to(DEV) for t in ts]
    out_h = net(x1.to(DEV), x2.to(DEV), to(gtc), None, True)
    assert len(out_h[0]) == 3 and [tuple(t.shape) for t in out_h[0][0]] == [(1, 1024, 3), (1, 1024, 3), (1, 2048, 3), (1, 512, 3), (1, 256, 3)]
    assert [tuple(t.shape) for t in out_h[2][0]] == [(1, 3, 1024), (1, 3, 256), (1, 3, 64), (1, 3, 32)]
    loss_h, parts = training.multiscale_loss(*out_h, to(gtc))
    loss_h.backward()
    assert abs(float(loss_h.detach()) - float(loss_c.detach())) <= 1e-4 * abs(float(loss_c.detach())), (float(loss_h.detach()), float(loss_c.detach()))
    gc = {n: p.grad for n, p in cpu_net.named_parameters()}
    gh = {n: p.grad for n, p in net.named_parameters()}
    assert {n for n, g in gc.items() if g is not None} == {n for n, g in gh.items() if g is not None}
    used = [n for n, g in gc.items() if g is not None]
    assert len(used) >= 280
    vc = torch.cat([gc[n].flatten() for n in used]).double()
    vh = torch.cat([gh[n].cpu().flatten() for n in used]).double()
    cos = float((vc @ vh) / (vc.norm() * vh.norm()))
    assert cos >= 0.999 and abs(float(vh.norm() / vc.norm()) - 1.0) <= 0.01, (cos, float(vh.norm() / vc.norm()))
    # a small step along the negative gradient lowers the objective
    with torch.no_grad():
        for p in net.parameters():
            if p.grad is not None:
                p -= 1e-6 * p.grad
    out2 = net(x1.to(DEV), x2.to(DEV), to(gtc), None, True)
    loss2, _ = training.multiscale_loss(*out2, to(gtc))
    assert float(loss2.detach()) < float(loss_h.detach())


def test_train_mode_batch_statistics_match_the_reference_on_the_gpu(capsys):
    """net.train() + forward(train=True) on the HIP backend vs the reference's own net.train() forward (dropout rates 0):
    frames, update counters and running statistics (tests/golden/train_b2_n1024.npz)."""
    def report(msg):
        with capsys.disabled():
            print("\n" + msg)
    hc.run_train_mode_check(DEV, report)


def test_train_mode_dropout_is_drawn_from_the_torch_generator():
    """With the reference's rates (0.05 / 0.05 / 0.04) a net.train() forward is stochastic, reproducible under the same device
    seed, back-propagates finite gradients, and leaves net.eval() forwards untouched (no dropout there)."""
    x1, x2, gt = synth.make_batch(1, 1, 1024, device=DEV)
    gtc = [g.transpose(1, 2).contiguous() for g in gt]
    net = hc.build_model(DEV)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    net.train()
    assert (net.drop_rate, net.attn_drop_rate, net.drop_path_rate) == (0.05, 0.05, 0.04)

    def run(seed):
        net.load_state_dict(state)  # same running statistics going in
        net.train()
        torch.manual_seed(seed)
        return net(x1, x2, gtc, None, True)
    a, b, c = run(1), run(1), run(2)
    assert all(torch.equal(u, v) for u, v in zip(a[3], b[3]))
    assert any(not torch.equal(u, v) for u, v in zip(a[3], c[3]))
    loss, _ = training.multiscale_loss(*c, gtc)
    loss.backward()
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    assert len(grads) >= 280 and all(torch.isfinite(g).all() for g in grads)
    net.load_state_dict(state)
    net.eval()
    with torch.no_grad():
        e1, e2 = net(x1, x2), net(x1, x2)
    assert all(torch.equal(u, v) for u, v in zip(e1, e2))


def test_backward_entry_points_reject_bad_arguments_and_handle_tiny_batches():
    """The C ABI of the backward kernels: wrong neighbour counts / head dims / feature widths are MCP_ERR_UNSUPPORTED, a short
    workspace or a null pointer MCP_ERR_BAD_ARG (no launch); and a batch smaller than one workgroup's worth of points (a single
    cloud of 40 points: 32 of its 40 points are everybody's neighbours) still gives the unfused layer's gradients."""
    import ctypes
    from mocopci_amd import _lib, grad
    lib, be = _lib.load(), ops.backend()
    f = lambda *s: torch.zeros(*s, device=DEV)
    i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    p1, idx, w = f(1, 64, 3), i32(1, 64, 64), [f(64, 4), f(64), f(64, 64), f(64), f(128, 64), f(128)]
    ws = torch.zeros(lib.mcp_fusion_grad_workspace_bytes(1, 64), dtype=torch.uint8, device=DEV)
    args = lambda nb, wsb: (1, 64, nb, P(p1), P(p1), P(idx), None, *[P(t) for t in w], P(p1), P(f(1, 64, 3)), P(f(1, 64, 64, 3)), P(f(12800)), P(ws), wsb, st)
    assert lib.mcp_fusion_grad(*args(32, ws.numel())) == 10002          # MCP_ERR_UNSUPPORTED: the layer's k is 32 + 32
    assert lib.mcp_fusion_grad(*args(64, 16)) == 10001                  # MCP_ERR_BAD_ARG: workspace too small
    assert lib.mcp_cross_grad_floats(256) == 0 and lib.mcp_cross_grad_floats(64) == 64 * 64 + 5 * 64
    assert lib.mcp_attention_small_grad_workspace_bytes(2, 100, 8) == 2 * 2 * 8 * 100 * 4
    q = f(1, 64, 96)
    assert lib.mcp_attention_small_grad(1, 64, 64, 8, 12, P(q), 96, P(q), 96, P(q), 96, 1.0, 0.0, 0, P(q), P(q), P(q), P(q), P(ws), ws.numel(), st) == 10002
    assert lib.mcp_attention_small_dropout(1, 64, 64, 8, 8, P(q), 64, P(q), 64, P(q), 64, 1.0, 1.5, 0, P(q), st) == 10001   # drop_p outside [0, 1)
    assert lib.mcp_prelu_dropout(64, P(q), P(q), 1.0, 0, P(q), st) == 10001                                                # likewise
    assert lib.mcp_prelu_dropout_grad(64, P(q), P(q), P(q), 0.5, 0, P(q), P(q), P(q), 0, st) == 10001                      # workspace too small
    assert lib.mcp_prelu_dropout(0, P(q), P(q), 0.5, 0, P(q), st) == 10001
    torch.cuda.synchronize()
    # tiny batches through the public operators
    n = 40
    a = cloud(220, 1, n).to(DEV)
    b = a + rnd(221, 1, n, 3, scale=0.2).to(DEV)
    halves = (be.knn(a, a, 32), be.knn(a, b, 32))
    wf = [rnd(222 + k, *sh, scale=sc).to(DEV) for k, (sh, sc) in enumerate((((64, 4), 0.5), ((64,), 0.1), ((64, 64), 0.125), ((64,), 0.1), ((128, 64), 0.125), ((128,), 0.1)))]
    g = rnd(230, 1, n, 3).to(DEV)

    def grads(fn):
        leaves = [t.detach().clone().requires_grad_(True) for t in (a, b, *wf)]
        return torch.autograd.grad(fn(*leaves), leaves, g)
    got = grads(lambda x, y, *ww: be.fusion_mlp(x, y, halves, *ww))
    want = grads(lambda x, y, *ww: grad.fusion_twin(be.group_rows, x, y, halves, *ww))
    for u, v in zip(got, want):
        assert float((u - v).abs().max()) <= 5e-4 * float(v.abs().max()) + 2e-5
    idx16 = be.knn(a, a, 16)
    qkv = rnd(231, 1, n, 192).to(DEV)
    wp = [rnd(232, 64, 3, scale=0.3).to(DEV), rnd(233, 64, scale=0.1).to(DEV)] + [t for k in range(3) for t in (rnd(234 + k, 64, 64, scale=0.125).to(DEV), rnd(238 + k, 64, scale=0.1).to(DEV))]
    gp = rnd(242, 1, n, 64).to(DEV)

    def pgrads(fn):
        leaves = [t.detach().clone().requires_grad_(True) for t in (a, qkv, *wp)]
        x, pq = leaves[0], leaves[1]
        return torch.autograd.grad(fn(x, pq[..., :64], pq[..., 64:128], pq[..., 128:], leaves[2:]), leaves, gp)
    got = pgrads(lambda x, q_, k_, v_, ww: be.ptblock_layer(x, q_, k_, v_, idx16, ww))
    want = pgrads(lambda x, q_, k_, v_, ww: grad.ptblock_twin(be.group_rows, x, q_, k_, v_, idx16, *ww))
    for u, v in zip(got, want):
        assert float((u - v).abs().max()) <= 5e-4 * float(v.abs().max()) + 2e-5
