import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, _lib
be = ops.backend()
def t(fn, reps=7):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e) * 1e3)
    return statistics.median(v)
g = torch.Generator().manual_seed(1)
for b, n in ((24, 8192), (16, 2048)):
    x = ((torch.rand(b, n, 3, generator=g) * 2 - 1) * torch.tensor([40.0, 40.0, 3.0])).cuda().contiguous()
    def build():
        be._clouds = []
        return be._sorted_cloud(x)
    print(f"B={b} N={n}: full _sorted_cloud {t(build):.1f} us")
    box = torch.cat([x.amin(dim=1), x.amax(dim=1)], dim=-1).contiguous()
    print("   amin/amax/cat", t(lambda: torch.cat([x.amin(dim=1), x.amax(dim=1)], dim=-1).contiguous()))
    codes = torch.empty((b, n), dtype=torch.int32, device="cuda")
    print("   morton", t(lambda: ops._call("mcp_morton_codes", x, b, n, _lib.fptr(x), _lib.fptr(box), _lib.iptr(codes))))
    print("   torch.sort", t(lambda: torch.sort(codes, dim=1)))
    perm = torch.sort(codes, dim=1)[1]
    print("   .int()", t(lambda: perm.int()))
    p32 = perm.int()
    print("   group_rows", t(lambda: be.group_rows(x, p32)))
