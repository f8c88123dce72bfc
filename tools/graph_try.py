import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
for _ in range(3): ref = net(x1, x2)
torch.cuda.synchronize()
def bench(fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("eager ms/step", bench(lambda: net(x1, x2)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): net(x1, x2)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = net(x1, x2)
g.replay(); torch.cuda.synchronize()
print("graph ms/step", bench(g.replay))
print("match", all(torch.equal(a, b) for a, b in zip(out, ref)))
