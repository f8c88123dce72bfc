import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from mocopci_amd import ops
from tests.test_grad_gpu import cloud, rnd
DEV = "cuda:0"
be = ops.backend()
for (B, N) in ((2, 300), (3, 1501)):
    p1 = cloud(30, B, N).to(DEV)
    p2 = p1 + rnd(31, B, N, 3, scale=0.2).to(DEV)
    idx = (be.knn(p1, p1, 32), be.knn(p1, p2, 32))
    conv = [t.to(DEV) for t in (rnd(32, 64, 4, scale=0.5), rnd(33, 64, scale=0.3), rnd(34, 64, 64, scale=0.125), rnd(35, 64, scale=0.3), rnd(36, 128, 64, scale=0.125), rnd(37, 128, scale=0.3))]
    aff = [t.to(DEV) for t in (1 + rnd(38, 64, scale=0.2), rnd(39, 64, scale=0.2), 1 + rnd(40, 64, scale=0.2), rnd(41, 64, scale=0.2), 1 + rnd(42, 128, scale=0.2), rnd(43, 128, scale=0.2))]
    out, bn, var = be.fusion_bn_forward(p1, p2, idx, conv, aff, 1e-3)
    whole = torch.cat(idx, dim=-1).long()
    nb = p2.double()[torch.arange(B, device=DEV).view(B, 1, 1), whole]
    r = nb - p1.double().unsqueeze(2)
    x = torch.cat([r, r.norm(dim=-1, keepdim=True)], dim=-1)
    off = 0
    for i, c in enumerate((64, 64, 128)):
        z = x @ conv[2 * i].double().T + conv[2 * i + 1].double()
        flat = z.reshape(-1, c)
        mean, v = flat.mean(0), flat.var(0, unbiased=False)
        print(f"B={B} N={N} layer {i+1}: mean err {float((bn[off:off+c].double() - mean).abs().max()):.2e} (scale {float(mean.abs().max()):.2e})  "
              f"var rel err {float(((var[[0,64,128][i]:[0,64,128][i]+c].double() - v).abs() / v).max()):.2e}  rstd rel err {float(((bn[off+c:off+2*c].double() - (v + 1e-3).rsqrt()).abs() * (v + 1e-3).sqrt()).max()):.2e}")
        x = torch.relu((z - mean) * (aff[2 * i].double() * torch.rsqrt(v + 1e-3)) + aff[2 * i + 1].double())
        off += 4 * c
    wgt = torch.softmax(x.max(dim=-1)[0], dim=-1)
    want = torch.sum(wgt.unsqueeze(-1) * nb, dim=2)
    print(f"   out max err {float((out.double() - want).abs().max()):.2e} (scale {float(want.abs().max()):.2e})")
