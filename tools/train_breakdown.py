"""Where one eval-graph training step goes, per fused layer: forward and backward device time of every grad.RecomputeFn node,
grouped by the twin it differentiates (HIP events around each node; the events serialise nothing, the training forward runs on
one stream).  usage: python tools/train_breakdown.py [batch] [npoints] [eval|train]"""
import collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import grad, synth, training
from mocopci_amd.model import MoCoPCI

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
MODE = sys.argv[3] if len(sys.argv) > 3 else "eval"
spans = []  # (label, phase, start event, end event, rows)


def label_of(twin):
    names = [n for n in twin.__code__.co_names if n.endswith("_twin")]
    return names[0] if names else twin.__name__


orig_fwd, orig_bwd = grad.RecomputeFn.forward, grad.RecomputeFn.backward


def timed(phase, fn, label_from):
    def wrapper(ctx, *args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn(ctx, *args)
        b.record()
        spans.append((label_from(ctx, args), phase, a, b))
        return out
    return wrapper


def shape_tag(args):
    t = next((x for x in args if isinstance(x, torch.Tensor) and x.dim() == 3), None)
    return "" if t is None else f" {t.shape[0]}x{t.shape[1]}x{t.shape[2]}"


grad.RecomputeFn.forward = staticmethod(timed("fwd", orig_fwd, lambda ctx, args: label_of(args[1]) + shape_tag(args[2:])))
grad.RecomputeFn.backward = staticmethod(timed("bwd", orig_bwd, lambda ctx, args: label_of(ctx.twin) + shape_tag(ctx.saved_tensors)))

net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
net.train(MODE == "train")
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
x1, x2, gt = synth.make_batch(2, B, N, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
for it in range(3):
    spans.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    frames_f, frames_b, gt_frame, out_lst = net(x1, x2, gtc, None, True)
    loss, parts = training.multiscale_loss(frames_f, frames_b, gt_frame, out_lst, gtc)
    ev[1].record()
    opt.zero_grad()
    loss.backward()
    ev[2].record()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 2.0)
    opt.step()
    ev[3].record()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"[{MODE}] B={B} N={N}: step {dt * 1e3:.1f} ms wall; device: forward+loss {ev[0].elapsed_time(ev[1]):.1f}  backward {ev[1].elapsed_time(ev[2]):.1f}  "
      f"clip+Adam {ev[2].elapsed_time(ev[3]):.1f} ms")
tot = collections.defaultdict(lambda: [0.0, 0])
for label, phase, a, b in spans:
    e = tot[(label, phase)]
    e[0] += a.elapsed_time(b); e[1] += 1
family = collections.defaultdict(lambda: [0.0, 0.0, 0])
for (label, phase), (ms, cnt) in tot.items():
    f = family[label.split(" ")[0]]
    f[0 if phase == "fwd" else 1] += ms
    if phase == "bwd":
        f[2] += cnt
print(f"{'layer':24s} {'calls':>5s} {'forward ms':>11s} {'backward ms':>12s}")
for name, (fw, bw, cnt) in sorted(family.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:24s} {cnt:5d} {fw:11.2f} {bw:12.2f}")
print(f"{'all RecomputeFn nodes':24s} {sum(v[2] for v in family.values()):5d} {sum(v[0] for v in family.values()):11.2f} {sum(v[1] for v in family.values()):12.2f}")
print("\nper shape (backward, >= 1 ms):")
for (label, phase), (ms, cnt) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    if phase == "bwd" and ms >= 1.0:
        print(f"  {label:44s} x{cnt:3d} {ms:8.2f} ms")
