import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
for _ in range(3): net(x1, x2)
torch.cuda.synchronize()
# CPU issue time with an empty GPU queue (one forward, then sync)
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t = time.perf_counter(); net(x1, x2); ts.append((time.perf_counter() - t) * 1e3); torch.cuda.synchronize()
print("CPU issue ms (GPU idle at start):", [round(v, 2) for v in ts])
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): net(x1, x2)
torch.cuda.synchronize(); print("steady ms/step:", (time.perf_counter() - t) / 10 * 1e3)
