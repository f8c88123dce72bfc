import os, sys, statistics, torch
sys.path.insert(0, '/root/repo')
from mocopci_amd import ops, synth
be = ops.backend()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
a = torch.cat([x1, x2]).transpose(1, 2).contiguous()
def t(fn, reps=9):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e) * 1e3)
    return statistics.median(v)
for n in (8192, 2048):
    c = a[:, :n].contiguous()
    print(n, "m=2", t(lambda: be.fps(c, 2)), "m=34", t(lambda: be.fps(c, 34)), "m=66", t(lambda: be.fps(c, 66)))
g = torch.Generator().manual_seed(0)
big = ((torch.rand(8, 65536, 3, generator=g) * 2 - 1) * torch.tensor([80.0, 80.0, 6.0])).cuda().contiguous()
x1b, x2b, _ = synth.make_batch(2, 4, 65536, device="cuda")
lidar = torch.cat([x1b, x2b]).transpose(1, 2).contiguous()
for name, c in (("uniform box", big), ("lidar-like", lidar)):
    print(f"65536 {name}: m=2 {t(lambda: be.fps(c, 2), 3):.0f} us  m=258 {t(lambda: be.fps(c, 258), 3):.0f} us  m=2048 {t(lambda: be.fps(c, 2048), 3):.0f} us")
