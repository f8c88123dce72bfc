"""The split-K few-row Linear (linear_splitk_kernel) against the library chain (F.linear + LeakyReLU) at the few-row shapes of the
N=8192, B=8 pipeline: device time per call and max deviation."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops
be = ops.backend()
ops.HipBackend._LIN_MIN_ROWS = 0
ops.HipBackend._LIN_MAX_K = 1 << 20
ops.HipBackend._LIN_MAX_N = 1 << 20
dev = "cuda"
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
torch.manual_seed(0)
for rows, k, n, slope in ((4096, 1048, 256, 0.1), (8192, 536, 128, 0.1), (1024, 2072, 256, 0.1), (8192, 128, 128, 0.1), (8192, 128, 64, 0.1), (4096, 256, 256, 0.1),
                          (4096, 256, 768, 1.0), (4096, 256, 1536, 1.0), (6144, 256, 256, 1.0), (6144, 256, 1024, 1.0), (4096, 576, 256, 1.0), (2048, 512, 256, 1.0),
                          (8192, 64, 64, 0.1), (12288, 128, 128, 1.0)):
    x = torch.randn(rows, k, device=dev)
    w, b = torch.randn(n, k, device=dev) * k ** -0.5, torch.randn(n, device=dev) * 0.1
    lib = (lambda: F.leaky_relu(F.linear(x, w, b), slope)) if slope != 1.0 else (lambda: F.linear(x, w, b))
    if not be.linear_supported(x, n):
        print(f"{rows:6d} x {k:4d} -> {n:4d}: library {t(lib):6.1f} us, kernel: shape not supported", flush=True)
        continue
    pk = be.linear_pack(w, b, [k])
    ours = lambda: be.linear(x, w, b, slope, None, packed=pk)
    d = (ours() - lib()).abs().max().item()
    print(f"{rows:6d} x {k:4d} -> {n:4d}: library {t(lib):6.1f} us, split-K kernel {t(ours):6.1f} us, max |diff| {d:.2e}", flush=True)
