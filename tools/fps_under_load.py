"""Diagnostic: furthest point sampling on a side stream while the main stream is busy with large kernels -- the sampled indices must
not depend on what else the chip is doing."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mocopci_amd import ops, synth
be = ops.backend()
dev = "cuda"
x1, x2, _ = synth.make_batch(2, 8, 8192, device=dev)
xyz = torch.cat([x1, x2]).transpose(1, 2).contiguous()
levels = []
cur = xyz
for m in (2048, 512, 256, 64):
    sel = be.fps(cur, m)
    levels.append((cur, m, sel))
    cur = be.group_rows(cur, sel)
torch.cuda.synchronize()
a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream()
kind = sys.argv[1] if len(sys.argv) > 1 else "gemm"
p1 = torch.randn(24, 8192, 3, device=dev) * 20
idx = torch.randint(0, 8192, (24, 8192, 64), device=dev, dtype=torch.int32)
g = torch.Generator(device=dev).manual_seed(1)
ws = [torch.randn(64, 4, device=dev) * 0.5, torch.randn(64, device=dev) * 0.1, torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev) * 0.1,
      torch.randn(128, 64, device=dev) / 8, torch.randn(128, device=dev) * 0.1]
for trial in range(3):
    bad = {m: 0 for _, m, _ in levels}
    for rep in range(60):
        if kind == "gemm":
            for _ in range(3): a @ b
        elif kind == "fusion":
            for _ in range(2): be.fusion_mlp(p1, p1, idx, *ws)
        side.wait_stream(torch.cuda.current_stream()) if kind == "serial" else None
        with torch.cuda.stream(side):
            outs = [(m, be.fps(c, m), want) for c, m, want in levels]
        torch.cuda.synchronize()
        for m, got, want in outs:
            bad[m] += int(not torch.equal(got, want))
    print(kind, "trial", trial, "launches with a different sample, of 60:", bad, flush=True)
