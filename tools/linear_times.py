"""Every F.linear of one forward ranked by time (CUDA events around each call serialise the streams: for ranking only)."""
import collections, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI
import mocopci_amd.model as M
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
for _ in range(2): net(x1, x2)
rec = []
orig = F.linear
def timed(x, w, b=None):
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record(); y = orig(x, w, b); e.record(); torch.cuda.synchronize()
    rec.append((s.elapsed_time(e) * 1e3, tuple(x.shape), tuple(w.shape)))
    return y
F.linear = timed; M.F.linear = timed
net(x1, x2)
F.linear = orig; M.F.linear = orig
print(f"{len(rec)} linears, {sum(r[0] for r in rec) / 1e3:.2f} ms")
agg = collections.defaultdict(lambda: [0.0, 0])
for t, xs, ws in rec:
    rows = 1
    for d in xs[:-1]: rows *= d
    agg[(rows, ws[1], ws[0])][0] += t; agg[(rows, ws[1], ws[0])][1] += 1
for (rows, k, n), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {t:8.1f} us  {c:2d}x  rows {rows:8d}  {k:4d} -> {n:4d}   {2.0 * rows * k * n * c / t / 1e6:6.1f} TFLOP/s  {4.0 * rows * (k + n) * c / t / 1e6:6.2f} TB/s")
