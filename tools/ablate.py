"""What each block of the step is WORTH in the real (untraced, pipelined) serving loop: ms per step with that block's launches
removed.  A removed call returns the tensors it produced in a recorded steady-state step (same shapes, same values: everything
downstream runs unchanged), so the difference to the full step is the block's marginal cost -- which, beside side lanes that keep the
chip busy, is NOT its kernel time in a trace.  Diagnostic only (results of an ablated step are stale by construction).
    python tools/ablate.py [group ...]     groups: see GROUPS below; default = every group"""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
from mocopci_amd.model import MoCoPCI

net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()
be = ops.backend()


class Replay:
    """Wraps obj.name: 'off' passes through; 'record' stores the results of the calls of one loop iteration by call order;
    'replay' returns them without calling."""
    def __init__(self, obj, name, pred=None):
        self.obj, self.name, self.orig, self.pred = obj, name, getattr(obj, name), pred
        self.mode, self.store, self.i = "off", [], 0
        setattr(obj, name, self)

    def __call__(self, *a, **k):
        if self.mode == "off" or (self.pred is not None and not self.pred(*a, **k)):
            return self.orig(*a, **k)
        if self.mode == "record":
            r = self.orig(*a, **k)
            self.store.append(r)
            return r
        r = self.store[self.i]
        self.i += 1
        return r

    def start(self, mode):
        self.mode, self.i = mode, 0
        if mode == "record":
            self.store = []


def run(n, wraps=(), mode="off"):
    h = net.prefetch(x1, x2, ev)
    pend = out = None
    for i in range(n):
        nxt = None if i == n - 1 else (x1, x2, ev)
        # steady-state iterations only: the first iteration has no tail to finish, the last prefetches nothing
        m = mode if 2 <= i < n - 1 else "off"
        if mode == "record":
            m = "record" if i == 3 else "off"
        for w in wraps:
            w.start(m)
        cur = net.begin(x1, x2, prefetched=h, then_prefetch=nxt)
        if pend is not None:
            out = net.finish(pend)
        pend = cur
        h = net.take_prefetched()
    for w in wraps:
        w.start("off")
    if pend is not None:
        out = net.finish(pend)
    return out


def timed(wraps, steps=30):
    run(6, wraps, "record")
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(steps, wraps, "replay" if wraps else "off")
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / steps * 1e3)
    return min(best)


rows = lambda x: x.numel() // x.shape[-1] if isinstance(x, torch.Tensor) else sum(t.numel() // t.shape[-1] for t in x[:1])
GROUPS = {
    "fusion": lambda: [Replay(be, "fusion_mlp")],
    "cross (all 7)": lambda: [Replay(be, "cross_layer")],
    "knn K=32": lambda: [Replay(be, "knn", lambda q, r, k, **kw: k == 32)],
    "knn K=16": lambda: [Replay(be, "knn", lambda q, r, k, **kw: k == 16)],
    "knn_cosine": lambda: [Replay(be, "knn_cosine")],
    "interp3 search+apply": lambda: [Replay(be, "interp3_search"), Replay(be, "interp3_apply")],
    "attention (frame blocks)": lambda: [Replay(be, "attention")],
    "attention_rot (EI + cross_block3)": lambda: [Replay(be, "attention_rot")],
    "ei cross-formers whole": lambda: [Replay(net, "ei_crossformer_folded")],
    "pointconv fused (L0, L1, refine)": lambda: [Replay(be, "pointconv_linear")],
    "pointconv agg (L2-4)": lambda: [Replay(be, "pointconv_agg")],
    "encoder L2-4 whole (pointconv + lin)": lambda: [Replay(net, "pointconv", lambda prefix, *a, **k: prefix in ("encoder.level2", "encoder.level3", "encoder.level4"))],
    "mlp2": lambda: [Replay(be, "mlp2")],
    "linear (fused kernel)": lambda: [Replay(be, "linear")],
    "F.linear (library)": lambda: [Replay(F, "linear")],
    "ptblock": lambda: [Replay(be, "ptblock_layer")],
    "cross_block3 whole": lambda: [Replay(net, "cross_frame_att_pair")],
    "mfa core (q/kv, attention, mlps) both levels": lambda: [Replay(net, "_mfa_core")],
    "fps (pyramid + refine)": lambda: [Replay(be, "fps")],
    "torch.cat": lambda: [Replay(torch, "cat")],
}
want = [a for a in sys.argv[1:] if not a.startswith("--")] or list(GROUPS)
run(6); torch.cuda.synchronize()
base = timed(())
print(f"full step {base:.3f} ms", flush=True)
for g in want:
    wraps = GROUPS[g]()
    t = timed(wraps)
    for w in wraps:
        setattr(w.obj, w.name, w.orig)
    print(f"  without {g:48s} {t:.3f} ms   ({(base - t) * 1e3:+7.0f} us)", flush=True)
print(f"full step again {timed(()):.3f} ms")
