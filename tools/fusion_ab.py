import os, sys, torch, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
be = ops.backend()
torch.manual_seed(0)
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
p1 = torch.cat([x1, x1, x2]).transpose(1, 2).contiguous()
p2 = (p1 + 0.05 * torch.randn_like(p1)).contiguous()
idx = torch.cat([be.knn(p1, p1, 32), be.knn(p1, p2, 32)], dim=-1).contiguous()
g = torch.Generator(device="cuda").manual_seed(1)
ws = [torch.randn(64, 4, device="cuda", generator=g) * 0.5, torch.randn(64, device="cuda", generator=g) * 0.1,
      torch.randn(64, 64, device="cuda", generator=g) / 8, torch.randn(64, device="cuda", generator=g) * 0.1,
      torch.randn(128, 64, device="cuda", generator=g) / 8, torch.randn(128, device="cuda", generator=g) * 0.1]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e))
    return statistics.median(v)
out = be.fusion_mlp(p1, p2, idx, *ws)
tag = os.path.basename(os.environ.get("MCP_HIP_LIB", "shipped")).replace(".so", "") + "_f32" * (os.environ.get("MCP_FUSION_F32_MFMA", "0") == "1")
print(tag, "fusion 24x8192: %.3f ms" % t(lambda: be.fusion_mlp(p1, p2, idx, *ws), reps=15), " checksum %.9g" % float(out.double().sum()), flush=True)
