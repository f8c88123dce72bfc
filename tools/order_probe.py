"""Does the ORDER in which a gather-heavy kernel walks its points matter?  The pipeline's clouds come in scan-independent order (the
synthetic frames are randomly permuted; the FPS-sampled levels are in sampling order, i.e. consecutive points are far apart), so the 32
rows a wave gathers share nothing with its neighbours' rows.  Here the same call is timed with its QUERY points physically re-ordered
along the Hilbert curve of the sorted clouds (inputs permuted row-wise: same work, same results up to the row order).
    python tools/order_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
be = ops.backend()
dev = "cuda"


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def hilbert_perm(xyz):
    with be.cloud_scope():
        _, perm, _ = be._sorted_cloud(xyz) if xyz.shape[1] >= 1024 else be._build_cloud(xyz)
    return perm.long()


def rows(tns, perm):
    return torch.gather(tns, 1, perm.view(*perm.shape, *([1] * (tns.dim() - 2))).expand(-1, -1, *tns.shape[2:])).contiguous()


x1, x2, _ = synth.make_batch(2, 8, 8192, device=dev)
xyz = torch.cat([x1, x2]).transpose(1, 2).contiguous()                    # (16,8192,3)
w = lambda *s: torch.randn(*s, device=dev) * 0.1
for name, n, d in (("level 1", 2048, 64), ("level 2", 512, 128)):
    B = 48 if n == 512 else 36
    base = xyz[:, :1].expand(-1, 1, -1)
    pc, _ = be.fps(xyz, n, with_points=True) if False else (None, None)
    sel = be.fps(xyz, n)
    pts = be.group_rows(xyz, sel)                                         # (16,n,3) in sampling order
    pts = pts.repeat(B // 16 + 1, 1, 1)[:B].contiguous()
    other = (pts + 0.3 * torch.randn_like(pts)).contiguous()
    f1, f2 = torch.randn(B, n, d, device=dev), torch.randn(B, n, d, device=dev)
    idx = be.knn(pts, other, 32)
    pk = be.cross_pack(w(d, 3), w(d), w(d, d), w(d))
    perm = hilbert_perm(pts)
    p_pts, p_f1, p_idx = rows(pts, perm), rows(f1, perm), rows(idx, perm)
    # also the REFERENCE side re-ordered (rows of other / f2 moved, indices renamed): neighbours of neighbouring queries are then neighbouring rows
    perm2 = hilbert_perm(other)
    inv2 = torch.empty_like(perm2); inv2.scatter_(1, perm2, torch.arange(n, device=dev).expand(B, n))
    o2, g2 = rows(other, perm2), rows(f2, perm2)
    idx2 = torch.gather(inv2, 1, p_idx.reshape(B, -1).long()).reshape(B, n, 32).int().contiguous()
    a = t(lambda: be.cross_volume(pts, other, f1, f2, idx, pk))
    b = t(lambda: be.cross_volume(p_pts, other, p_f1, f2, p_idx, pk))
    c = t(lambda: be.cross_volume(p_pts, o2, p_f1, g2, idx2, pk))
    print(f"cross {name} ({B} x {n}, D = {d}): sampling order {a:.1f} us, queries along the curve {b:.1f} us, queries and references along the curve {c:.1f} us")
# PointConv level 0: centres = every point of the cloud, neighbours from the self search
f0 = torch.randn(16, 8192, 32, device=dev)
idx0 = be.knn(xyz, xyz, 32)
wn = [w(8, 3), w(8), w(8, 8), w(8), w(8, 8), w(8)]
wl, bl = w(32, 35 * 8), w(32)
pk = be.pointconv_linear_pack(wl, bl)
perm = hilbert_perm(xyz)
inv = torch.empty_like(perm); inv.scatter_(1, perm, torch.arange(8192, device=dev).expand(16, 8192))
pxyz, pf0 = rows(xyz, perm), rows(f0, perm)
pidx = torch.gather(inv, 1, rows(idx0, perm).reshape(16, -1).long()).reshape(16, 8192, 32).int().contiguous()
a = t(lambda: be.pointconv_linear(xyz, xyz, f0, idx0, *wn, wl, bl, 0.1, packed=pk))
b = t(lambda: be.pointconv_linear(xyz, rows(xyz, perm), f0, rows(idx0, perm), *wn, wl, bl, 0.1, packed=pk) if False else None) if False else 0.0
c = t(lambda: be.pointconv_linear(pxyz, pxyz, pf0, pidx, *wn, wl, bl, 0.1, packed=pk))
print(f"pointconv level 0 (16 x 8192, D = 32): scan-independent order {a:.1f} us, cloud stored along the curve {c:.1f} us")
idx64 = torch.cat([idx0, idx0], dim=-1).contiguous()
pidx64 = torch.cat([pidx, pidx], dim=-1).contiguous()
ws = [w(64, 4), w(64), w(64, 64), w(64), w(128, 64), w(128)]
a = t(lambda: be.fusion_mlp(xyz, xyz, idx64, *ws))
c = t(lambda: be.fusion_mlp(pxyz, pxyz, pidx64, *ws))
print(f"fusion (16 x 8192): scan-independent order {a:.1f} us, cloud stored along the curve {c:.1f} us")
