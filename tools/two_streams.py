"""Throughput of back-to-back forwards issued on one stream vs alternating over two streams (independent batches)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
def run(nstreams, steps=12):
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for i in range(4):
        with torch.cuda.stream(streams[i % nstreams]): net(x1, x2)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]): out = net(x1, x2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]): out = net(x1, x2)
    cpu = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"{nstreams} stream(s): {dt/steps*1e3:.2f} ms/step   (CPU issue time {cpu/steps*1e3:.2f} ms/step)")
    return out
a = run(1); b = run(2); c = run(3); run(1)
print("same results:", all(torch.equal(p, q) for p, q in zip(a, b)))
