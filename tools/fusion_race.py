"""Diagnostic: run the fusion kernel repeatedly on one input and describe any run-to-run difference (which points, how far off)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mocopci_amd import ops, grad
be = ops.backend()
torch.manual_seed(0)
DEV = "cuda"
B, N = 24, 8192
rnd = lambda *s, scale=1.0: torch.randn(*s, device=DEV) * scale
p1 = rnd(B, N, 3, scale=20.0)
p2 = p1 + rnd(B, N, 3, scale=0.05)
idx = torch.randint(0, N, (B, N, 64), device=DEV, dtype=torch.int32)
ws = [rnd(64, 4, scale=0.5), rnd(64, scale=0.1), rnd(64, 64, scale=0.125), rnd(64, scale=0.1), rnd(128, 64, scale=0.125), rnd(128, scale=0.1)]
ref = None
with torch.no_grad():
    want = torch.cat([grad.fusion_twin(be.group_rows, p1[b:b+1], p2[b:b+1], idx[b:b+1], *ws) for b in range(B)])
outs = [be.fusion_mlp(p1, p2, idx, *ws).clone() for _ in range(12)]
torch.cuda.synchronize()
for i, o in enumerate(outs):
    bad = ((o - want).abs() > 1e-3 * want.abs().clamp_min(1.0)).any(-1).reshape(-1)
    ids = bad.nonzero().flatten().tolist()
    print(f"run {i}: {len(ids)} points off vs unfused twin", ids[:12], flush=True)
    for p in ids[:4]:
        b, n = divmod(p, N)
        print("    point", p, "wave slot", p % 16384, "got", o[b, n].tolist(), "want", want[b, n].tolist())
