"""32-NN search time against the geometry of the reference cloud: the fusion stage searches the REFINED cloud, which the synthetic `pred` head of the stress weights collapses to a blob."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
be = ops.backend()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
c = torch.cat([x1, x2, x1]).transpose(1, 2).contiguous()   # (24,8192,3)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
near = (c + 0.3 * torch.randn_like(c)).contiguous()
blob = (torch.randn_like(c) * 0.5).contiguous()
far = (c * 0.02).contiguous()
print("self search            %.0f us" % t(lambda: be.knn(c, c, 32)))
print("ref = cloud + 0.3 noise %.0f us" % t(lambda: be.knn(c, near, 32)))
print("ref = 0.5-unit blob     %.0f us" % t(lambda: be.knn(c, blob, 32)))
print("ref = cloud scaled 0.02 %.0f us" % t(lambda: be.knn(c, far, 32)))
