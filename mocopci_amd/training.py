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
    """Chamfer of every prediction in `preds` (each (B,n,3)) against the same ground-truth cloud (B,m,3): [mean over B] per prediction."""
    import torch
    B = preds[0].shape[0]
    if len(preds) == 1:
        return [ops.backend().chamfer(preds[0].contiguous(), gt_rows)]
    v = ops.backend().chamfer(torch.cat([p.contiguous() for p in preds], dim=0), gt_rows.repeat(len(preds), 1, 1), per_sample=True)
    return list(v.reshape(len(preds), B).mean(dim=1).unbind(0))   # one mean and one unbind (backward: one stack), not a slice per term


def _multiscale_loss(frames_lst_f, frames_lst_b, gt_frame, out_lst, gt):
    rows = lambda g: g.transpose(1, 2).contiguous()   # (B,3,n) as train.py holds the ground truth -> (B,n,3)
    loss_f = sum(_stacked([frames], rows(g))[0] for frames, g in zip(out_lst, gt))
    loss_s_f = loss_s_b = loss_m_f = loss_m_b = 0.0
    for frames_f, frames_b, gts in zip(frames_lst_f, frames_lst_b, gt_frame):
        f0, f1, b0, b1 = _stacked([frames_f[0], frames_f[1], frames_b[0], frames_b[1]], rows(gts[0]))
        loss_s_f = loss_s_f + 0.5 * f0 + 0.5 * f1
        loss_s_b = loss_s_b + 0.5 * b0 + 0.5 * b1
        for l in range(len(ALPHA) - 1):
            mf, mb = _stacked([frames_f[l + 2], frames_b[l + 2]], rows(gts[l + 1]))
            loss_m_f = loss_m_f + ALPHA[l + 1] * mf
            loss_m_b = loss_m_b + ALPHA[l + 1] * mb
    total = loss_f + (loss_s_f + loss_s_b) / 2 + 0.25 * loss_m_b + 0.25 * loss_m_f
    return total, {"final": loss_f, "straight_f": loss_s_f, "straight_b": loss_s_b, "multi_f": loss_m_f, "multi_b": loss_m_b}


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
