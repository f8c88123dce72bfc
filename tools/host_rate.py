"""Host enqueue time per step vs device time per step: if the two are close the forward is launch-bound on the host."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()
def run(n):
    """the bench's serving loop: two batches in flight (begin / finish)"""
    h = net.prefetch(x1, x2, ev)
    pend = None
    for i in range(n):
        cur = net.begin(x1, x2, prefetched=h, then_prefetch=None if i == n - 1 else (x1, x2, ev))
        if pend is not None:
            net.finish(pend)
        pend = cur
        h = net.take_prefetched()
    net.finish(pend)
run(5)
torch.cuda.synchronize()
steps = 30
t0 = time.perf_counter()
run(steps)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_enq / steps * 1e3:.2f} ms/step, device {t_all / steps * 1e3:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
run(5)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
