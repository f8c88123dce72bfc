import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, _lib
be = ops.backend(); lib = _lib.load()
lib.mcp_fps_diag_read.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
for n, m in ((8192, 2048), (2048, 512), (16384, 2048)):
    x = (torch.rand(16, n, 3, device="cuda") * 80 - 40).contiguous()
    be.fps(x, m); torch.cuda.synchronize(); lib.mcp_fps_diag_read(buf)   # warm + clear
    be.fps(x, m); torch.cuda.synchronize(); lib.mcp_fps_diag_read(buf)
    it = m - 1
    names = ["centre read", "scan", "wave reduce+select", "atomic+barrier", "slot read+decode"]
    tot = sum(buf[i] for i in range(5))
    print(f"N={n} M={m}: cycles/iter total {tot/it:.0f}: " + ", ".join(f"{nm} {buf[i]/it:.0f}" for i, nm in enumerate(names)))
