"""Pruned-KNN work counters (needs a library built with -DMCP_KNN_DIAG, loaded through MCP_HIP_LIB)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, _lib, synth
be = ops.backend(); lib = _lib.load()
lib.mcp_knn_diag_read.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 4)()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
a = torch.cat([x1, x2]).transpose(1, 2).contiguous()
far = (torch.randn_like(a) * 0.3).contiguous()
cases = {"self 16x8192 k32": (a, a, 32), "2048q in 8192 k32": (a[:, :2048].contiguous(), a, 32), "self 16x8192 k16": (a, a, 16),
         "blob refs k32": (a, far, 32), "2048x2048 k16": (a[:, :2048].contiguous(), a[:, 2048:4096].contiguous(), 16)}
be.PRUNE_MIN_REFS = 64; be.PRUNE_MIN_QUERIES = 64
for name, (q, r, k) in cases.items():
    be.knn(q, r, k); torch.cuda.synchronize(); lib.mcp_knn_diag_read(buf)
    be.knn(q, r, k); torch.cuda.synchronize(); lib.mcp_knn_diag_read(buf)
    w = buf[0]
    print(f"{name:22s} waves {w}: tiles/wave {buf[1]/w:.2f} flushes/wave {buf[2]/w:.2f} pushes/lane {buf[3]/w/64:.1f}")
