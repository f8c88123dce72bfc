"""One-off: time the harness sections on the GPU by wrapping MoCoPCI methods with CUDA events."""
import collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI

net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
acc = collections.defaultdict(float); cnt = collections.Counter(); depth = [0]
def wrap(name):
    fn = getattr(net, name)
    def w(*a, **k):
        if depth[0] > 0: return fn(*a, **k)   # only outermost wrapped call is timed
        depth[0] += 1
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); r = fn(*a, **k); e.record(); torch.cuda.synchronize()
        depth[0] -= 1
        acc[name] += s.elapsed_time(e); cnt[name] += 1
        return r
    setattr(net, name, w)
for n in ["pointconv", "fps_gather", "cross", "interp", "warp", "ei_crossformer", "cross_frame_att", "multi_frame_att",
          "transformer_block", "fusion"]:
    wrap(n)
for _ in range(2): net(x1, x2)
acc.clear(); cnt.clear()
torch.cuda.synchronize(); t = time.time(); net(x1, x2); torch.cuda.synchronize(); tot = (time.time() - t) * 1e3
print("total ms (with per-section syncs)", tot)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:20s} {v:8.2f} ms  calls {cnt[k]}")
print("sum", sum(acc.values()))
