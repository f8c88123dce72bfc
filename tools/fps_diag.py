"""Cycles per phase of one FPS iteration (thread 0 of workgroup 0; needs a library built with -DMCP_FPS_DIAG, passed as MCP_HIP_LIB).
Phases: 0 centre read, 1 box test, 2 update + wave reductions, 3 LDS atomic + barrier, 4 slot read + store."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import _lib, ops, synth
lib = _lib.load()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
xyz = torch.cat([x1, x2]).transpose(1, 2).contiguous()
be = ops.backend()
buf = (ctypes.c_ulonglong * 8)()
be.fps(xyz, 2048); torch.cuda.synchronize(); lib.mcp_fps_diag_read(buf)
m = 2048
be.fps(xyz, m); torch.cuda.synchronize(); lib.mcp_fps_diag_read(buf)
tot = sum(buf[i] for i in range(5))
names = ["centre read", "box test", "update + wave reductions", "LDS atomic + barrier", "slot read + store"]
for i, n in enumerate(names):
    print(f"{n:28s} {buf[i] / (m - 1):8.1f} cycles/iteration")
print(f"{'total':28s} {tot / (m - 1):8.1f} cycles/iteration (s_memtime ticks)")
