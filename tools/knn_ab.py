"""Round-4 search kernel (knn_walk_kernel) against round 3's (knn_pruned_kernel): equality and time on the shapes of the step, and
the per-phase cycle / event counters of both.  Needs a library that holds both kernels:
    MCP_HIP_LIB=tools/ab/libknn_ab.so   python tools/knn_ab.py            (tools/build_variant.sh knn_ab -DMCP_AB: timing)
    MCP_HIP_LIB=tools/ab/libknn_diag.so python tools/knn_ab.py --diag     (-DMCP_KNN_DIAG: counters; its times include the stamps)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth, _lib
be = ops.backend()
lib = ctypes.CDLL(_lib.SO_PATH)
diag = "--diag" in sys.argv
once = "--once" in sys.argv   # one launch per kernel and case (counter passes)
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
c = torch.cat([x1, x2, x1]).transpose(1, 2).contiguous()   # (24,8192,3)
g = torch.Generator(device="cuda").manual_seed(0)
near = (c + 0.3 * torch.randn(c.shape, device="cuda", generator=g)).contiguous()
blob = (torch.randn(c.shape, device="cuda", generator=g) * 0.5).contiguous()
c2k, c512 = c[:, :2048].contiguous().repeat(2, 1, 1), c[:, :512].contiguous()
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
NAMES = ["waves", "tiles scanned", "tiles examined", "tau updates", "soft compactions", "hard compactions", "appended", "survivors",
         "cyc setup", "cyc walk", "cyc staging", "cyc scan", "cyc tau update", "cyc compaction", "cyc final"]
def read_diag():
    buf = (ctypes.c_ulonglong * 28)()
    lib.mcp_knn_diag_read(buf)
    return list(buf)
cases = [("self 24x8192 K32", c, c, 32, 0), ("near 24x8192 K32", c, near, 32, 0), ("blob 24x8192 K32", c, blob, 32, 0),
         ("self 16x8192 K32", c[:16].contiguous(), c[:16].contiguous(), 32, 0),
         ("self 48x2048 K16 direct", c2k, c2k, 16, 1), ("24x8192 -> 2048 K3", c, c[:, :2048].contiguous(), 3, 0), ("self 24x8192 K16 direct", c, c, 16, 1)]
for name, qq, rr, k, mode in cases:
    with be.cloud_scope():
        res = {}
        for old in (1, 0):
            lib.mcp_knn_pruned_use_old(1 if old else 2)   # 1: round 3's kernel for every K, 2: the walk kernel for every K
            if diag: read_diag()
            i, d = be.knn(qq, rr, k, mode=mode, return_dist=True)
            torch.cuda.synchronize()
            dg = read_diag() if diag else None
            us = 0.0 if once else t(lambda: be.knn(qq, rr, k, mode=mode))
            res[old] = (i, d, us, dg)
        same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        print(f"{name:28s} old {res[1][2]:7.1f} us   new {res[0][2]:7.1f} us   identical: {same}")
        if not same:
            bad = (res[0][0] != res[1][0]).any(-1)
            print("   differing queries:", int(bad.sum()), "of", bad.numel(), " first:", bad.nonzero()[:3].tolist())
        if diag:
            o, nw = res[1][3], res[0][3]
            w = max(o[0], 1)
            print(f"   old: waves {o[0]}  tiles/wave {o[1]/w:.1f}  flushes/wave {o[2]/w:.1f}  pushes/query {o[3]/(w*16):.1f}")
            ON = ["setup", "walk", "staging", "scan", "flush", "final"]
            otot = sum(o[20:26])
            print("   old cycles/wave: " + "  ".join(f"{ON[j]} {o[20+j]/w:.0f} ({100*o[20+j]/max(otot,1):.0f}%)" for j in range(6)) + f"  total {otot/w:.0f}")
            w = max(nw[4], 1)
            print("   new: " + "  ".join(f"{NAMES[j]}/wave {nw[4+j]/w:.1f}" for j in range(1, 8)))
            tot = sum(nw[12:19])
            print("   new cycles/wave: " + "  ".join(f"{NAMES[8+j][4:]} {nw[12+j]/w:.0f} ({100*nw[12+j]/max(tot,1):.0f}%)" for j in range(7)) + f"  total {tot/w:.0f}")
lib.mcp_knn_pruned_use_old(0)
if hasattr(lib, "mcp_knn_occupancy"):
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib.mcp_knn_occupancy(ctypes.byref(a), ctypes.byref(b))
    print(f"occupancy query (workgroups of one wave per CU): walk {a.value}, round 3 {b.value} (rc {rc})")
