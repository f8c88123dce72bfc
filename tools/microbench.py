"""Per-kernel timings on the GPU (median of interleaved rounds, CUDA events)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
be = ops.backend()
dev = "cuda"
x1, x2, _ = synth.make_batch(2, 8, 8192, device=dev)
xyz16 = torch.cat([x1, x2]).transpose(1, 2).contiguous()
xyz24 = torch.cat([xyz16, xyz16[:8] + 0.01]).contiguous()
q2048 = xyz16[:, :2048].contiguous()
f64 = torch.randn(16, 2048, 64, device=dev); f64b = torch.randn(16, 2048, 64, device=dev)
f128 = torch.randn(16, 512, 128, device=dev); f128b = torch.randn(16, 512, 128, device=dev)
idx32 = torch.randint(0, 2048, (16, 2048, 32), device=dev, dtype=torch.int32)
idx64 = torch.randint(0, 8192, (24, 8192, 64), device=dev, dtype=torch.int32)
w = lambda *s: torch.randn(*s, device=dev) * 0.1
qa = torch.randn(80, 2048, 64, device=dev); kva = torch.randn(80, 2048, 128, device=dev)
qb = torch.randn(80, 512, 128, device=dev); kvb = torch.randn(80, 512, 256, device=dev)
pk64 = be.cross_pack(w(64, 3), w(64), w(64, 64), w(64)); pk128 = be.cross_pack(w(128, 3), w(128), w(128, 128), w(128))
q512 = xyz16[:, :512].contiguous(); idx32b = torch.randint(0, 512, (16, 512, 32), device=dev, dtype=torch.int32)
cases = {
    "fps 16x8192->2048": lambda: be.fps(xyz16, 2048),
    "fps 24x8192->2048": lambda: be.fps(xyz24, 2048),
    "fps 16x2048->512": lambda: be.fps(q2048, 512),
    "knn 16x8192x8192 k32": lambda: be.knn(xyz16, xyz16, 32),
    "knn 24x8192x8192 k32": lambda: be.knn(xyz24, xyz24, 32),
    "knn 16x2048qx8192 k32": lambda: be.knn(q2048, xyz16, 32),
    "knn 16x2048x2048 k16": lambda: be.knn(q2048, q2048, 16),
    "knn 24x8192qx2048 k3": lambda: be.knn(xyz24, xyz24[:, :2048].contiguous(), 3),
    "knn_cosine 16x2048 c64": lambda: be.knn_cosine(f64, f64b, 16),
    "knn_cosine 16x512 c128": lambda: be.knn_cosine(f128, f128b, 16),
    "cross 16x2048 d64": lambda: be.cross_volume(q2048, q2048, f64, f64b, idx32, pk64),
    "cross 16x512 d128": lambda: be.cross_volume(q512, q512, f128, f128b, idx32b, pk128),
    "fusion 24x8192": lambda: be.fusion_mlp(xyz24, xyz24, idx64, w(64, 4), w(64), w(64, 64), w(64), w(128, 64), w(128)),
    "attention 80x8h x2048 hd8": lambda: be.attention(qa, kva, 8),
    "attention 80x8h x512 hd16": lambda: be.attention(qb, kvb, 8),
    "attention 8x8h x2048 hd8": lambda: be.attention(qa[:8].contiguous(), kva[:8].contiguous(), 8),
    "pointconv_agg 16x8192 d32": lambda: be.pointconv_agg(xyz16, xyz16, torch.randn(16, 8192, 32, device=dev), idx64[:16, :, :32].contiguous(), w(8, 3), w(8), w(8, 8), w(8), w(8, 8), w(8)),
}
only = sys.argv[1:] 
times = {k: [] for k in cases if not only or any(o in k for o in only)}
for k in times: cases[k]()
torch.cuda.synchronize()
for r in range(5):
    for k in times:
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); cases[k](); e.record(); torch.cuda.synchronize()
        times[k].append(s.elapsed_time(e) * 1e3)
for k, v in times.items():
    print(f"{k:32s} median {statistics.median(v):9.1f} us  min {min(v):9.1f} us")
