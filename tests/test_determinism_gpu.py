"""-m gpu: every fused MFMA kernel run repeatedly on the same inputs must give bit-identical outputs (no atomics, no races).
Sized like the N=8192 pipeline on purpose: a software-pipelined variant of the fusion kernel was wrong on ~5 of 196608 points
per launch, only with two waves per SIMD -- invisible at unit-test sizes, caught by the forward determinism test and by this one."""
import pytest
import torch

from mocopci_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*s, scale=1.0):
    return torch.randn(*s, device=DEV) * scale


def repeatable(fn, reps=6):
    ref = fn()
    torch.cuda.synchronize()
    for _ in range(reps):
        out = fn()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            return False
    return True


def test_fusion_kernel_is_deterministic():
    torch.manual_seed(0)
    be = ops.backend()
    B, N = 24, 8192
    p1 = rnd(B, N, 3, scale=20.0)
    p2 = p1 + rnd(B, N, 3, scale=0.05)
    idx = torch.randint(0, N, (B, N, 64), device=DEV, dtype=torch.int32)
    ws = [rnd(64, 4, scale=0.5), rnd(64, scale=0.1), rnd(64, 64, scale=0.125), rnd(64, scale=0.1), rnd(128, 64, scale=0.125), rnd(128, scale=0.1)]
    assert repeatable(lambda: be.fusion_mlp(p1, p2, idx, *ws))


@pytest.mark.parametrize("d,n", [(64, 2048), (128, 512), (256, 256)])
def test_cross_kernel_is_deterministic(d, n):
    torch.manual_seed(d)
    be = ops.backend()
    B = 48
    x1, x2, f1, f2 = rnd(B, n, 3, scale=10.0), rnd(B, n, 3, scale=10.0), rnd(B, n, d), rnd(B, n, d)
    ix = torch.randint(0, n, (B, n, 32), device=DEV, dtype=torch.int32)
    pk = be.cross_pack(rnd(d, 3, scale=0.3), rnd(d, scale=0.1), rnd(d, d, scale=d ** -0.5), rnd(d, scale=0.1))
    assert repeatable(lambda: be.cross_volume(x1, x2, f1, f2, ix, pk))


def test_ptblock_and_wide_attention_are_deterministic():
    torch.manual_seed(1)
    be = ops.backend()
    n = 2048
    xyz, qkv = rnd(24, n, 3, scale=10.0), rnd(24, n, 192)
    ix = torch.randint(0, n, (24, n, 16), device=DEV, dtype=torch.int32)
    pk = be.ptblock_pack(rnd(64, 3, scale=0.3), rnd(64, scale=0.1), *[t for _ in range(3) for t in (rnd(64, 64, scale=0.125), rnd(64, scale=0.1))])
    assert repeatable(lambda: be.ptblock_attention(xyz, qkv[..., :64], qkv[..., 64:128], qkv[..., 128:], ix, pk))
    q, kv = rnd(32, 256, 768), rnd(32, 256, 1536)
    assert repeatable(lambda: be.attention(q, kv, 3, scale=1 / 16))
    q, kv = rnd(8, 256, 256), rnd(8, 256, 512)
    assert repeatable(lambda: be.attention(q, kv, 8))
    q, kv = rnd(48, 2048, 64), rnd(48, 2048, 128)
    assert repeatable(lambda: be.attention(q, kv, 8))


def test_mlp2_is_deterministic():
    torch.manual_seed(2)
    be = ops.backend()
    x, res = rnd(49152, 64), rnd(49152, 64)
    pk = be.mlp2_pack(rnd(256, 64, scale=0.125), rnd(256, scale=0.1), rnd(64, 256, scale=0.06), rnd(64, scale=0.1))
    w = (rnd(256, 64), rnd(256), rnd(64, 256), rnd(64))  # shapes only: the packed image carries the values
    assert repeatable(lambda: be.mlp2(x, *w, 0.25, res=res, packed=pk))


def test_cross_kernel_with_batch_map_is_deterministic():
    """The replicated-batch form the inference graph uses (features and the first index list read through a batch map)."""
    torch.manual_seed(5)
    be = ops.backend()
    B0, R, n, d = 16, 3, 2048, 64
    B = B0 * R
    x1, x2 = rnd(B, n, 3, scale=10.0), rnd(B, n, 3, scale=10.0)
    f1, f2 = rnd(B0, n, d), rnd(B0, n, d)
    ic = torch.randint(0, n, (B0, n, 16), device=DEV, dtype=torch.int32)
    ip = torch.randint(0, n, (B, n, 16), device=DEV, dtype=torch.int32)
    bmap = torch.tensor([i % B0 for i in range(B)], device=DEV, dtype=torch.int32)
    pk = be.cross_pack(rnd(d, 3, scale=0.3), rnd(d, scale=0.1), rnd(d, d, scale=d ** -0.5), rnd(d, scale=0.1))
    assert repeatable(lambda: be.cross_volume(x1, x2, f1, f2, (ic, ip), pk, bmap=bmap, shared=7))


def test_linear_pointconv_and_small_attention_are_deterministic():
    torch.manual_seed(6)
    be = ops.backend()
    x = rnd(196608, 536)
    w, b = rnd(64, 536, scale=536 ** -0.5), rnd(64, scale=0.1)
    pk = be.linear_pack(w, b, [536])
    assert repeatable(lambda: be.linear(x, w, b, 0.1, None, packed=pk))
    xs = [rnd(32768, 64), rnd(32768, 64), rnd(32768, 64)]
    w3, res = rnd(64, 192, scale=192 ** -0.5), rnd(32768, 64)
    assert repeatable(lambda: be.linear(xs, w3, b, 1.0, res))
    # PointConv aggregation at level 0 of the pipeline
    B, n = 16, 8192
    xyz, feat = rnd(B, n, 3, scale=10.0), rnd(B, n, 32)
    idx = torch.randint(0, n, (B, n, 32), device=DEV, dtype=torch.int32)
    wn = [rnd(8, 3, scale=0.5), rnd(8, scale=0.1), rnd(8, 8, scale=0.4), rnd(8, scale=0.1), rnd(8, 8, scale=0.4), rnd(8, scale=0.1)]
    assert repeatable(lambda: be.pointconv_agg(xyz, xyz, feat, idx, *wn))
    # head dims 8 / 16 (Multi_Frame_Att at the two pyramid levels)
    q, kv = rnd(48, 2048, 64), rnd(48, 2048, 128)
    assert repeatable(lambda: be.attention(q, kv, 8))
    q, kv = rnd(48, 512, 128), rnd(48, 512, 256)
    assert repeatable(lambda: be.attention(q, kv, 8))


def test_fps_is_independent_of_what_another_stream_runs():
    """Furthest point sampling on a side stream while the main stream runs the fusion kernel (what the steady state of
    forward(inputs_ready=...) does): the sampled indices must be those of a quiet chip.  With packed-fp32 instructions in the
    scan ~8 % of the level-2 launches sampled different points (mocopci_amd/csrc/common.h: mcp_f2)."""
    from mocopci_amd import synth
    be = ops.backend()
    x1, x2, _ = synth.make_batch(2, 8, 8192, device=DEV)
    cur = torch.cat([x1, x2]).transpose(1, 2).contiguous()
    levels = []
    for m in (2048, 512, 256, 64):
        sel = be.fps(cur, m)
        levels.append((cur, m, sel))
        cur = be.group_rows(cur, sel)
    torch.cuda.synchronize()
    torch.manual_seed(3)
    p1 = rnd(24, 8192, 3, scale=20.0)
    idx = torch.randint(0, 8192, (24, 8192, 64), device=DEV, dtype=torch.int32)
    ws = [rnd(64, 4, scale=0.5), rnd(64, scale=0.1), rnd(64, 64, scale=0.125), rnd(64, scale=0.1), rnd(128, 64, scale=0.125), rnd(128, scale=0.1)]
    side = torch.cuda.Stream()
    for _ in range(40):
        be.fusion_mlp(p1, p1, idx, *ws)
        be.fusion_mlp(p1, p1, idx, *ws)
        with torch.cuda.stream(side):
            outs = [(m, be.fps(c, m), want) for c, m, want in levels]
        torch.cuda.synchronize()
        for m, got, want in outs:
            assert torch.equal(got, want), f"level {m} sampled different points beside the fusion kernel"


def test_back_to_back_pipelined_forwards_are_bit_identical():
    """forward(inputs_ready=...) issued back to back: the sampling pyramid of call k+1 runs under the tail of call k.  Every
    call must return exactly what an isolated call returns."""
    from mocopci_amd import synth
    from tests import harness_checks as hc
    net = hc.build_model(DEV)
    x1, x2, _ = synth.make_batch(2, 8, 8192, device=DEV)
    ev = torch.cuda.Event()
    ev.record()
    alone = net(x1, x2)
    torch.cuda.synchronize()
    outs = [net(x1, x2, inputs_ready=ev) for _ in range(5)]
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert all(torch.equal(a, b) for a, b in zip(o, alone)), f"pipelined call {k} differs from the isolated call"


def test_prefetched_forwards_are_bit_identical_to_an_isolated_call():
    """The serving loop of bench.py: prefetch(batch k+1) is issued right after batch k's encoder is enqueued and consumed by the
    next forward.  Every call must return exactly what an isolated call returns; a handle for other inputs is refused."""
    from mocopci_amd import synth
    from tests import harness_checks as hc
    net = hc.build_model(DEV)
    x1, x2, _ = synth.make_batch(2, 8, 8192, device=DEV)
    y1, y2, _ = synth.make_batch(2, 8, 8192, device=DEV, first_sample=8)
    alone_x = net(x1, x2)
    alone_y = net(y1, y2)
    torch.cuda.synchronize()
    batches = [(x1, x2, alone_x), (y1, y2, alone_y), (x1, x2, alone_x), (x1, x2, alone_x)]
    h = net.prefetch(*batches[0][:2])
    outs = []
    for k, (a, b, _) in enumerate(batches):
        nxt = batches[k + 1][:2] if k + 1 < len(batches) else None
        outs.append(net(a, b, prefetched=h, then_prefetch=nxt))
        h = net.take_prefetched()
    assert h is None
    torch.cuda.synchronize()
    for k, (o, (_, _, want)) in enumerate(zip(outs, batches)):
        assert all(torch.equal(p, q) for p, q in zip(o, want)), f"prefetched call {k} differs from the isolated call"
    h = net.prefetch(x1, x2)
    with pytest.raises(RuntimeError):
        net(y1, y2, prefetched=h)
    torch.cuda.synchronize()
    # a loader that refills the SAME buffers in place between prefetch() and the forward: the handle holds the old contents
    z1, z2 = x1.clone(), x2.clone()
    h = net.prefetch(z1, z2)
    z1.copy_(y1)
    with pytest.raises(RuntimeError):
        net(z1, z2, prefetched=h)
    torch.cuda.synchronize()


@pytest.mark.parametrize("tail_on_own_stream", [False, True])
def test_two_batches_in_flight_give_the_isolated_results(tail_on_own_stream):
    """begin(batch k+1) is enqueued BEFORE finish(batch k) (bench.py's serving loop): the refinement-stage sampling of batch k runs
    beside batch k+1's encoder and the deferred tail takes the non-speculative PointConvD path.  Every batch must come out
    exactly as an isolated forward() returns it."""
    from mocopci_amd import synth
    from tests import harness_checks as hc
    net = hc.build_model(DEV)
    x1, x2, _ = synth.make_batch(2, 8, 8192, device=DEV)
    y1, y2, _ = synth.make_batch(2, 8, 8192, device=DEV, first_sample=8)
    want = {0: net(x1, x2), 1: net(y1, y2)}
    torch.cuda.synchronize()
    order = [0, 1, 1, 0, 0]
    tail = torch.cuda.Stream() if tail_on_own_stream else None   # finish() on a stream of its own, beside the next batch's first part
    batches = {0: (x1, x2), 1: (y1, y2)}
    h = net.prefetch(*batches[order[0]])
    pend, outs = None, []
    for k, which in enumerate(order):
        nxt = batches[order[k + 1]] if k + 1 < len(order) else None
        cur = net.begin(*batches[which], prefetched=h, then_prefetch=nxt)
        if pend is not None:
            outs.append(net.finish(pend, tail_stream=tail))
        pend = cur
        h = net.take_prefetched()
    outs.append(net.finish(pend, tail_stream=tail))
    torch.cuda.synchronize()
    assert len(outs) == len(order)
    for k, (o, which) in enumerate(zip(outs, order)):
        assert all(torch.equal(a, b) for a, b in zip(o, want[which])), f"pipelined batch {k} differs from its isolated forward"


@pytest.mark.parametrize("b", [1, 2])
def test_pipelined_forms_match_the_isolated_forward_at_small_batches(b):
    """below 3 sequences the sampled rows of the two PointConvD stages number fewer than 16384, where a Linear on them
    alone would take the few-row split-K kernel (another K summation order) while an isolated forward() computes every candidate
    row with the full-K kernel and gathers.  The prefetched / begin() + finish() forms ask for the tall product's kernels
    (MoCoPCI.lin(like_rows=), mcp_linear_as), so all three forms return the same bits at every batch size."""
    from mocopci_amd import synth
    from tests import harness_checks as hc
    net = hc.build_model(DEV)
    x1, x2, _ = synth.make_batch(2, b, 8192, device=DEV)
    want = net(x1, x2)
    torch.cuda.synchronize()
    h = net.prefetch(x1, x2)
    got = net(x1, x2, prefetched=h)
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(got, want)), "prefetched forward differs from the isolated one"
    pend = net.begin(x1, x2)
    other = net.begin(x1, x2)
    outs = [net.finish(pend), net.finish(other)]
    torch.cuda.synchronize()
    for o in outs:
        assert all(torch.equal(p, q) for p, q in zip(o, want)), "begin() / finish() differs from the isolated forward"
