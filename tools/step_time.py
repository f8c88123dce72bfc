"""ms per step of the N=8192, B=8 inference pipeline (the bench's prefetch loop without its instrumentation), for A/B runs:
    python tools/step_time.py [HipBackend attribute=value ...] [--steps N]
e.g. `python tools/step_time.py _LIN_MIN_ROWS=4096` (policy attributes of mocopci_amd.ops.HipBackend are set on the class)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
from mocopci_amd.model import MoCoPCI

steps = 40
for a in sys.argv[1:]:
    if a.startswith("--steps"):
        steps = int(a.split("=")[1])
    elif a.startswith("--move="):   # --move=node:lane, e.g. --move="('fus', 3):3" or --move=wf:none -- an entry of schedule.NODE_LANES
        from mocopci_amd import schedule
        node, lane = a[7:].rsplit(":", 1)
        schedule.NODE_LANES[eval(node) if node.startswith("(") else node] = None if lane == "none" else int(lane)
    elif a.startswith("net.LANE_MAP="):   # e.g. net.LANE_MAP=0,1,2,3,0,5: side lanes folded onto fewer HIP streams
        MoCoPCI.LANE_MAP = tuple(int(t) for t in a.split("=")[1].split(","))
    elif a.startswith("net."):
        k, v = a[4:].split("=")
        setattr(MoCoPCI, k, type(getattr(MoCoPCI, k))(int(v)))
    elif "=" in a and not a.startswith("--"):
        k, v = a.split("=")
        setattr(ops.HipBackend, k, type(getattr(ops.HipBackend, k))(int(v)) if not isinstance(getattr(ops.HipBackend, k), bool) else v == "1")
net = MoCoPCI(); net.load_state_dict((synth.weights_on_scan if "--scan-weights" in sys.argv else synth.weights_by_name)(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()


TAIL = torch.cuda.Stream() if "--tail-stream" in sys.argv else None


def run(n):
    """the bench's serving loop: two batches in flight (begin / finish) unless --prefetch-only"""
    h = net.prefetch(x1, x2, ev)
    pend = out = None
    for i in range(n):
        nxt = None if i == n - 1 else (x1, x2, ev)
        if "--prefetch-only" in sys.argv:
            out = net(x1, x2, prefetched=h, then_prefetch=nxt)
        else:
            cur = net.begin(x1, x2, prefetched=h, then_prefetch=nxt)
            if pend is not None:
                out = net.finish(pend, tail_stream=TAIL)
            pend = cur
        h = net.take_prefetched()
    if pend is not None:
        out = net.finish(pend, tail_stream=TAIL)
    return out


MAIN = torch.cuda.Stream(priority=-1) if "--high-prio" in sys.argv else torch.cuda.current_stream()   # the main stream above the side lanes
MAIN.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(MAIN):
    run(6)
torch.cuda.synchronize()
best = []
for rep in range(3):
    t0 = time.perf_counter()
    with torch.cuda.stream(MAIN):
        run(steps)
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / steps * 1e3)
print(" ".join(sys.argv[1:]) or "default", "ms/step:", " ".join(f"{b:.3f}" for b in best), flush=True)
