"""The reference's training objective (train.py:135-160) on the outputs of MoCoPCI.forward(train=True)."""
from . import ops

ALPHA = (1.0, 0.8, 0.4, 0.2)  # train.py:138


def chamfer_loss(pred, gt, rows=None):
    """models/utils.py:36-45 on the layouts train.py uses: pred (B,n,3) (the reference permutes its (B,n,3) frame to (B,3,n) and
    chamfer_loss permutes it back), gt (B,3,n).  rows: a memo {id(gt): (gt, its (B,n,3) copy)} so that a ground-truth cloud compared
    with several predictions is laid out once (and, inside a cloud_scope, sorted for the neighbour searches once).  The entry holds
    gt itself: a temporary view (zip over a stacked tensor) would otherwise die and hand its id to the next frame's view."""
    if rows is None:
        g = gt.transpose(1, 2).contiguous()
    else:
        hit = rows.get(id(gt))
        if hit is None or hit[0] is not gt:
            hit = rows[id(gt)] = (gt, gt.transpose(1, 2).contiguous())
        g = hit[1]
    return ops.backend().chamfer(pred.contiguous(), g)


def multiscale_loss(frames_lst_f, frames_lst_b, gt_frame, out_lst, gt):
    """losssum of train.py:135-160: final frames vs gt, the two full-resolution warps of both directions, and the level 1..3
    frames against the FPS-downsampled ground truth with weights alpha[1:].  The 33 Chamfer terms share 12 ground-truth clouds;
    terms that compare several predictions with the SAME cloud are evaluated as one call on a stacked batch (per-sample values,
    then the reference's sums): 15 calls, 30 searches, instead of 33 / 66 -- the same terms, added in the reference's order."""
    with ops.backend().cloud_scope():
        return _multiscale_loss(frames_lst_f, frames_lst_b, gt_frame, out_lst, gt)


def _stacked(preds, gt_rows):
    """Chamfer of every prediction in `preds` (each (B,n,3)) against the same ground-truth cloud (B,m,3): a (len(preds),) tensor, the
    mean over B per prediction -- one call on the stacked batch."""
    import torch
    B = preds[0].shape[0]
    if len(preds) == 1:
        return ops.backend().chamfer(preds[0].contiguous(), gt_rows).reshape(1)
    v = ops.backend().chamfer(torch.cat([p.contiguous() for p in preds], dim=0), gt_rows.repeat(len(preds), 1, 1), per_sample=True)
    return v.reshape(len(preds), B).mean(dim=1)   # one mean (backward: one expand), not a slice per term


_weights = {}   # (signature, device) -> (total weights (T,), part weights (5,T)) on the device, built once


def _multiscale_loss(frames_lst_f, frames_lst_b, gt_frame, out_lst, gt):
    """The terms as ONE vector t (in the order below) and the objective as one dot product with a constant weight vector -- the
    reference's nested sums (train.py:135-160) are 70 scalar adds and multiplies, each an autograd node with a launch of its own:
        final      sum of the terms of out_lst                                  weight 1     in the total
        straight   0.5 (f0 + f1) per triple, forward and backward               weight 1/2
        multi      ALPHA[l+1] x term per triple and level, forward and backward weight 1/4"""
    import torch
    rows = lambda g: g.transpose(1, 2).contiguous()   # (B,3,n) as train.py holds the ground truth -> (B,n,3)
    vals, spec = [], []                               # spec: (part index, weight inside the part) per term
    for frames, g in zip(out_lst, gt):
        vals.append(_stacked([frames], rows(g)))
        spec.append((0, 1.0))
    for frames_f, frames_b, gts in zip(frames_lst_f, frames_lst_b, gt_frame):
        vals.append(_stacked([frames_f[0], frames_f[1], frames_b[0], frames_b[1]], rows(gts[0])))
        spec += [(1, 0.5), (1, 0.5), (2, 0.5), (2, 0.5)]
        for l in range(len(ALPHA) - 1):
            vals.append(_stacked([frames_f[l + 2], frames_b[l + 2]], rows(gts[l + 1])))
            spec += [(3, ALPHA[l + 1]), (4, ALPHA[l + 1])]
    t = torch.cat(vals)
    key = (tuple(spec), str(t.device))
    if key not in _weights:
        in_total = (1.0, 0.5, 0.5, 0.25, 0.25)        # final, straight_f, straight_b, multi_f, multi_b
        parts_w = torch.zeros(5, len(spec))
        for i, (k, w) in enumerate(spec):
            parts_w[k, i] = w
        total_w = (torch.tensor(in_total).unsqueeze(1) * parts_w).sum(0)
        _weights[key] = (total_w.to(t.device), parts_w.to(t.device))
    total_w, parts_w = _weights[key]
    total = torch.dot(t, total_w)
    parts = parts_w @ t.detach()
    return total, {"final": parts[0], "straight_f": parts[1], "straight_b": parts[2], "multi_f": parts[3], "multi_b": parts[4]}


def train_step(net, optimizer, xyz1, xyz2, gt, clip=2.0):
    """One iteration of train.py:123-167: forward, loss, backward, gradient-norm clipping at 2.0, optimizer step."""
    import torch
    frames_f, frames_b, gt_frame, out_lst = net(xyz1, xyz2, gt, None, True)
    loss, parts = multiscale_loss(frames_f, frames_b, gt_frame, out_lst, gt)
    optimizer.zero_grad()
    with ops.segments_memo():
        loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), clip)
    optimizer.step()
    return float(loss.detach()), {k: float(v.detach()) for k, v in parts.items()}
