"""One inference step captured into a HIP graph (torch.cuda.graph) and replayed: ms per step against eager enqueue."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI

dev = torch.device("cuda", 0)
net = MoCoPCI()
net.load_state_dict(synth.weights_by_name(net._spec), strict=True)
net = net.to(dev)
x1, x2, _ = synth.make_batch(2, 8, 8192, device=dev)


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    ref = net(x1, x2)
print("eager: %.3f ms/step" % timed(lambda: net(x1, x2)), flush=True)
ev = torch.cuda.Event(); ev.record()
print("eager, inputs_ready: %.3f ms/step" % timed(lambda: net(x1, x2, inputs_ready=ev)), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        net(x1, x2)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = net(x1, x2)
torch.cuda.synchronize()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("max |graph - eager| per frame:", [float((a - b).abs().max()) for a, b in zip(out, ref)], flush=True)
print("graph replay: %.3f ms/step" % timed(g.replay), flush=True)
