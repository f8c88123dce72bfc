"""Cycles per phase of cross_kernel's per-point loop (wave 0 of workgroup 0; needs a library built with -DMCP_CROSS_DIAG passed as
MCP_HIP_LIB): x0 build (waits for the prefetched loads, pos MFMA, epilogue, split), issue of the next point's loads, the D x D MFMA
chain per output tile, neighbour max + store."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, _lib
be = ops.backend()
lib = ctypes.CDLL(_lib.SO_PATH)
lib.mcp_cross_diag_read.argtypes = [ctypes.c_void_p]
dev = "cuda"
w = lambda *s: torch.randn(*s, device=dev) * 0.1
names = ["loop overhead", "x0 build (incl. waiting for loads)", "issue next point's loads", "D x D MFMA chain (all tiles)", "neighbour max + store (all tiles)"]
for name, b, n, d in (("D=64", 40, 2048, 64), ("D=128", 48, 512, 128)):
    xyz1, xyz2 = torch.randn(b, n, 3, device=dev) * 10, torch.randn(b, n, 3, device=dev) * 10
    f1, f2 = torch.randn(b, n, d, device=dev), torch.randn(b, n, d, device=dev)
    base = torch.arange(n, device=dev).view(1, n, 1)
    idx = ((base + torch.randint(-64, 64, (b, n, 32), device=dev)) % n).int().contiguous()
    pk = be.cross_pack(w(d, 3), w(d), w(d, d), w(d))
    buf = (ctypes.c_ulonglong * 8)()
    be.cross_volume(xyz1, xyz2, f1, f2, idx, pk); torch.cuda.synchronize(); lib.mcp_cross_diag_read(buf)
    be.cross_volume(xyz1, xyz2, f1, f2, idx, pk); torch.cuda.synchronize(); lib.mcp_cross_diag_read(buf)
    tot = sum(buf[i] for i in range(5))
    grid_waves = (768 if d == 64 else 256) * (4 if d == 64 else 8)
    pts = b * n / grid_waves
    print(f"{name}: {pts:.0f} points per wave, {tot / pts:.0f} memtime ticks per point (100 MHz clock: x{10 * 2.4:.0f} shader cycles at 2.4 GHz)")
    for i in range(5):
        print(f"   {names[i]:40s} {buf[i] / pts:8.1f} ticks  {100.0 * buf[i] / tot:5.1f} %")
