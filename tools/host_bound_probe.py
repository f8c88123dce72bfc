"""Is the step host-bound?  The serving loop of bench.py (two batches in flight) with a busy-wait of d microseconds added to every
library call (ops._call: ~125 of the ~213 launches of a step).  If the step time grows by about 125 * d the enqueue thread is on the
critical path; if it does not move, the device is.  usage: python tools/host_bound_probe.py [d ...]   (default 0 2 5 10)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
from mocopci_amd.model import MoCoPCI

delays = [float(a) for a in sys.argv[1:]] or [0.0, 2.0, 5.0, 10.0]
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()
orig = ops._call
calls = [0]
def make(d):
    def slow(name, ref, *a):
        calls[0] += 1
        t = time.perf_counter() + d * 1e-6
        while time.perf_counter() < t:
            pass
        return orig(name, ref, *a)
    return slow
def run(n):
    h = net.prefetch(x1, x2, ev)
    pend = None
    for i in range(n):
        cur = net.begin(x1, x2, prefetched=h, then_prefetch=None if i == n - 1 else (x1, x2, ev))
        if pend is not None:
            net.finish(pend)
        pend = cur
        h = net.take_prefetched()
    return net.finish(pend)
run(5); torch.cuda.synchronize()
for d in delays:
    ops._call = make(d) if d > 0 else orig
    run(3); torch.cuda.synchronize()
    calls[0] = 0
    t0 = time.perf_counter(); run(30); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"+{d:4.1f} us per library call ({calls[0] // 30 if d > 0 else '~125'} calls per step): {ms:.3f} ms per step", flush=True)
ops._call = orig
