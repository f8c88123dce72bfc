"""Gradients of the fused point-set layers.

The reference has no fused backward either: its layers are compositions of torch ops whose gradients autograd derives, with
three hand-written scatter-add kernels for gather / group / interpolate (K3/K6/K9: sampling_gpu.cu:46-83, group_points_gpu.cu:8-44,
interpolate_gpu.cu:120-161, all atomicAdd).  Here

  * forward in training is the SAME fused HIP kernel as in inference (identical numerics);
  * backward re-evaluates the layer in its unfused form -- the "twin" functions below, restatements of the reference layers on
    channel-last tensors (citations on each) built from the differentiable row gather -- and lets autograd differentiate that:
    the standard recompute-in-backward trade (nothing of size B x N x K x C is kept between forward and backward);
  * the only non-dense piece of any of these gradients, the scatter-add of the row gather, is a DETERMINISTIC segmented
    reduction (mcp_group_rows_grad_sorted: stable sort by destination row + in-order sums) instead of atomics, so a training
    step is bit-reproducible.
Gradients flow to every floating-point input: features, weights, and coordinates (the reference's grouping of xyz is
differentiable too, and warped coordinates depend on predicted flows).
"""
import torch
import torch.nn.functional as F


class RecomputeFn(torch.autograd.Function):
    """y = fused(*args) with backward through twin(*args).  Tensor arguments that need a gradient are re-created as leaves."""

    @staticmethod
    def forward(ctx, fused, twin, *args):
        ctx.twin = twin
        ctx.is_tensor = [isinstance(a, torch.Tensor) for a in args]
        ctx.others = [a for a in args if not isinstance(a, torch.Tensor)]
        ctx.save_for_backward(*[a for a in args if isinstance(a, torch.Tensor)])
        with torch.no_grad():
            return fused(*args)

    @staticmethod
    def backward(ctx, grad_out):
        tensors, others = iter(ctx.saved_tensors), iter(ctx.others)
        needs = ctx.needs_input_grad[2:]
        args, leaves, slots = [], [], []
        for k, (is_t, need) in enumerate(zip(ctx.is_tensor, needs)):
            if not is_t:
                args.append(next(others))
                continue
            t = next(tensors)
            if need and t.is_floating_point():
                t = t.detach().requires_grad_(True)
                leaves.append(t)
                slots.append(k)
            args.append(t)
        with torch.enable_grad():
            y = ctx.twin(*args)
        grads = torch.autograd.grad(y, leaves, grad_out.contiguous(), allow_unused=True)
        out = [None] * len(ctx.is_tensor)
        for k, g in zip(slots, grads):
            out[k] = g
        return (None, None, *out)


def wants_grad(*args):
    return torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad for a in args)


def run(fused, twin, *args):
    """fused(*args), differentiable through twin when any argument asks for a gradient."""
    if wants_grad(*args):
        return RecomputeFn.apply(fused, twin, *args)
    return fused(*args)


def whole(idx):
    """A neighbour list given as its two halves (the two searches' outputs) or as one tensor -> one tensor."""
    return torch.cat(list(idx), dim=-1) if isinstance(idx, (tuple, list)) else idx


# ---- twins: the layers in unfused, differentiable form.  G = the differentiable row gather (ops.HipBackend.group_rows) ----
def pointconv_agg_twin(G, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2):
    """group / group_query + WeightNet + aggregation, mocopci.py:1218-1266, :1289-1300, :1330-1335."""
    B, S, _ = new_xyz.shape
    g_xyz = G(s_xyz, idx) - new_xyz.unsqueeze(2)
    new_points = torch.cat([g_xyz, G(s_points, idx)], dim=-1)
    w = g_xyz
    for ww, bb in ((w0, b0), (w1, b1), (w2, b2)):
        w = torch.relu(F.linear(w, ww, bb))
    return torch.matmul(new_points.transpose(2, 3), w).reshape(B, S, -1)


def cross_twin(G, xyz1, xyz2, points1, points2, idx, wpos, bpos, wmlp, bmlp):
    """cross() after its neighbour searches, pointconv_util.py:765-781 (one mlp layer).  idx: (B,N1,32) or its two halves."""
    idx = whole(idx)
    direction = G(xyz2, idx) - xyz1.unsqueeze(2)
    x = F.leaky_relu((G(points2, idx) + points1.unsqueeze(2)) + F.linear(direction, wpos, bpos), 0.1)
    return F.leaky_relu(F.linear(x, wmlp, bmlp), 0.1).max(dim=2)[0]


def fusion_twin(G, p1, p2, idx, w1, b1, w2, b2, w3, b3):
    """knn_group + fusion after the searches, mocopci.py:803-819 (BatchNorm already folded into (w, b)).  idx: (B,N,64) or halves."""
    idx = whole(idx)
    nb = G(p2, idx)
    resi = nb - p1.unsqueeze(2)
    x = torch.cat([resi, torch.norm(resi, dim=-1, keepdim=True)], dim=-1)
    for w, b in ((w1, b1), (w2, b2), (w3, b3)):
        x = torch.relu(F.linear(x, w, b))
    wgt = torch.softmax(x.max(dim=-1)[0], dim=-1)
    return torch.sum(wgt.unsqueeze(-1) * nb, dim=2)


def ptblock_twin(G, xyz, q, k, v, idx, wd1, bd1, wd2, bd2, wg1, bg1, wg2, bg2):
    """TransformerBlock.forward after knn and the projections, pointT_layer2.py:64-75."""
    kk, vv = G(k, idx), G(v, idx)
    pos = F.linear(torch.relu(F.linear(xyz.unsqueeze(2) - G(xyz, idx), wd1, bd1)), wd2, bd2)
    attn = F.linear(torch.relu(F.linear((q.unsqueeze(2) - kk) + pos, wg1, bg1)), wg2, bg2)
    attn = torch.softmax(attn / (kk.shape[-1] ** 0.5), dim=-2)
    return torch.sum(attn * (vv + pos), dim=2)


def attention_twin(q, kv, heads, scale):
    """softmax(q k^T scale) v per head, mocopci.py:72-86 / :650-667; q (BF,Nq,C), kv (BF,Nk,2C) = [k | v]."""
    BF, Nq, C = q.shape
    Nk, hd = kv.shape[1], C // heads
    qh = q.reshape(BF, Nq, heads, hd).permute(0, 2, 1, 3)
    kvh = kv.reshape(BF, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(qh, kvh[0], kvh[1], scale=scale)
    return o.permute(0, 2, 1, 3).reshape(BF, Nq, C)


def linear_twin(xs, w, b, slope, res):
    """Linear over the concatenation of the pieces + one-slope activation + residual (Conv1d wrapper, mocopci.py:1111-1127)."""
    x = torch.cat(list(xs), dim=-1) if isinstance(xs, (tuple, list)) else xs
    y = F.linear(x, w, b)
    if slope != 1.0:
        y = torch.where(y > 0, y, y * slope)
    return y if res is None else y + res


def mlp2_twin(x, res, w1, b1, w2, b2, slope):
    """Two-layer per-point MLP with a one-slope PReLU (Mlp_T, mocopci.py:1558-1565, with its affine neighbours folded in)."""
    hid = F.linear(x, w1, b1)
    out = F.linear(torch.where(hid > 0, hid, hid * slope), w2, b2)
    return out if res is None else out + res


def interp3_weights_twin(G, dense, sparse, idx3):
    """Inverse-distance weights of UpsampleFlow / PointWarping, mocopci.py:1475-1478, :1495-1498."""
    dist = torch.norm(G(sparse, idx3) - dense.unsqueeze(2), dim=3).clamp(min=1e-10)
    inv = 1.0 / dist
    return inv / inv.sum(dim=2, keepdim=True)


def interp3_apply_twin(G, feat, idx3, w3):
    """Weighted sum of the three neighbours' rows, mocopci.py:1480-1481, :1500-1501."""
    return torch.sum(w3.unsqueeze(-1) * G(feat, idx3), dim=2)


def chamfer_twin(G, x, y, ixy, iyx, per_sample=False):
    """chamfer_loss, models/utils.py:36-45 (pytorch3d defaults): nearest neighbours fixed by the search, squared L2 to them."""
    dxy = ((x - G(y, ixy.unsqueeze(-1)).squeeze(2)) ** 2).sum(-1)
    dyx = ((y - G(x, iyx.unsqueeze(-1)).squeeze(2)) ** 2).sum(-1)
    v = dxy.mean(1) + dyx.mean(1)
    return v if per_sample else v.mean()
