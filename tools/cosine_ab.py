"""Cosine-KNN kernels: round 4's split-reference kernel against round 3's shared-tile kernel (needs an -DMCP_AB build:
MCP_HIP_LIB=tools/ab/libknn_ab.so python tools/cosine_ab.py): identical indices / distances and time on the three shapes of the step."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, _lib
be = ops.backend()
lib = ctypes.CDLL(_lib.SO_PATH)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator(device="cuda").manual_seed(1)
for b, q, n, c in ((16, 2048, 2048, 64), (16, 512, 512, 128), (16, 256, 256, 256), (2, 300, 777, 64), (1, 33, 70, 128), (3, 1000, 40, 256)):
    qf, rf = torch.randn(b, q, c, device="cuda", generator=g), torch.randn(b, n, c, device="cuda", generator=g)
    rf[:, : min(n, 8)] = rf[:, min(n, 8) : 2 * min(n, 8)] if n >= 16 else rf[:, : min(n, 8)]   # duplicate rows: exact ties
    res = {}
    for old in (1, 0):
        lib.mcp_knn_cosine_use_old(old)
        i, d = be.knn_cosine(qf, rf, 16, return_dist=True)
        res[old] = (i, d, t(lambda: be.knn_cosine(qf, rf, 16)))
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    print(f"B={b} Q={q} N={n} C={c}: round 3 {res[1][2]:7.1f} us   split {res[0][2]:7.1f} us   identical: {same}")
lib.mcp_knn_cosine_use_old(0)
