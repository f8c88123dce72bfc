"""Where the main stream's time goes in the REAL (untraced, pipelined) step: hipEvent markers between the sections of
MoCoPCI.forward (model._mark), averaged over steps of the bench's prefetch loop.  About 17 event records per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI
for a in sys.argv[1:]:
    if a.startswith("net."):
        k, v = a[4:].split("=")
        setattr(MoCoPCI, k, type(getattr(MoCoPCI, k))(int(v)))
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()


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
                out = net.finish(pend)
            pend = cur
        h = net.take_prefetched()
    if pend is not None:
        out = net.finish(pend)
    return out


run(6)
torch.cuda.synchronize()
steps = 20
net._marks = []
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
marks = net._marks
net._marks = None
starts = [i for i, m in enumerate(marks) if m[0] == "enc start"]
per = starts[2] - starts[1]  # markers of one steady-state step (its first part, then the previous batch's deferred tail)
names = [m[0] for m in marks[starts[1]:starts[2]]]
acc = [0.0] * per
used = 0
for s in range(1, len(starts) - 2):
    if starts[s + 1] - starts[s] != per:
        continue
    used += 1
    for j in range(per):
        acc[j] += marks[starts[s] + j][1].elapsed_time(marks[starts[s] + j + 1][1])
steps = used + 2
print(f"{wall:.3f} ms/step (with markers); main-stream time from each marker to the next, averaged over {steps - 2} steps:")
tot = 0.0
for nme, v in zip(names, acc):
    v /= steps - 2
    tot += v
    print(f"  {v * 1e3:8.1f} us  after '{nme}'")
print(f"  sum {tot:.3f} ms")
