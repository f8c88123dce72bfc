"""oracle/backend.py -- TEST INFRASTRUCTURE.  CPU operator backend with the interface of
mocopci_amd.ops.HipBackend, built on the C oracle (oracle/pointset.py).  Installed with
mocopci_amd.ops.set_backend(OracleBackend()) by tests and by bench.py's cpu_baseline leg
to run the model harness graph on host cores; never used on the product path."""
import torch

from . import pointset as orc


class OracleBackend:
    name = "oracle-cpu"

    def prebuild_cloud(self, xyz):
        pass

    def cloud_scope(self, seed=None):
        import contextlib
        return contextlib.nullcontext()

    def attention_rot(self, q, k, v, heads, kv_shift, scale=None):
        BF, Nq, C = q.shape
        hd = C // heads
        k, v = torch.roll(k, -kv_shift, dims=0), torch.roll(v, -kv_shift, dims=0)  # batch b reads (b + kv_shift) mod BF
        sp = lambda t: t.reshape(BF, t.shape[1], heads, hd).permute(0, 2, 1, 3)
        o = torch.nn.functional.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=scale)
        return o.permute(0, 2, 1, 3).reshape(BF, Nq, C)

    def add_layernorm(self, x, y=None, bias=None, eps=1e-6):
        z = x if y is None else x + y
        z = z if bias is None else z + bias
        return torch.nn.functional.layer_norm(z, (z.shape[-1],), None, None, eps)

    def mfa_prepare(self, fea, src_self, src_partner, te_self, te_partner, scale, shift):
        x = fea[src_self.long()] + te_self[:, None, :]
        return x, torch.addcmul(shift, x, scale), torch.addcmul(shift, fea[src_partner.long()] + te_partner[:, None, :], scale)

    def fps(self, xyz, npoint, with_points=False):
        sel = orc.furthest_point_sample(xyz.detach(), npoint)
        return (sel, self.group_rows(xyz, sel)) if with_points else sel

    def knn(self, query, ref, k, mode=0, return_dist=False):
        return orc.knn(query.detach(), ref.detach(), k, mode=mode, return_dist=return_dist)

    def knn_cosine(self, qfeat, rfeat, k, return_dist=False):
        return orc.knn_cosine(qfeat.detach(), rfeat.detach(), k, return_dist=return_dist)

    def group_rows(self, points, idx):
        """Row gather; differentiable (plain torch indexing) when a gradient is wanted, the C restatement otherwise."""
        if points.requires_grad and torch.is_grad_enabled():
            B = points.shape[0]
            bidx = torch.arange(B).view(B, *([1] * (idx.dim() - 1)))
            return points[bidx, idx.long()]
        return orc.group_rows(points, idx.int())

    def group_rows_add_leaky(self, points, idx, centre, slope=0.1):
        return torch.nn.functional.leaky_relu(self.group_rows(points, idx) + centre.unsqueeze(2), slope)

    def interp3_search(self, dense, sparse):
        idx3 = orc.knn(dense.detach(), sparse.detach(), 3, mode=0)
        if torch.is_grad_enabled() and (dense.requires_grad or sparse.requires_grad):  # mocopci.py:1495-1498, differentiable
            dist = torch.norm(self.group_rows(sparse, idx3) - dense.unsqueeze(2), dim=3).clamp(min=1e-10)
            return idx3, (1.0 / dist) / torch.sum(1.0 / dist, dim=2, keepdim=True)
        B, N, _ = dense.shape
        w3 = torch.empty(B, N, 3, dtype=torch.float32)
        orc.lib().orc_interp3_weights(orc._f(dense.contiguous()), orc._f(sparse.contiguous()), orc._i(idx3), orc._f(w3), B, N,
                                      sparse.shape[1])
        return idx3, w3

    def interp3_apply(self, feat, idx3, w3):
        if torch.is_grad_enabled() and (feat.requires_grad or w3.requires_grad):
            return torch.sum(w3.unsqueeze(-1) * self.group_rows(feat, idx3), dim=2)
        B, N, _ = idx3.shape
        S, C = feat.shape[1], feat.shape[2]
        out = torch.empty(B, N, C, dtype=torch.float32)
        orc.lib().orc_interp3_apply(orc._f(feat.contiguous()), orc._i(idx3.contiguous()), orc._f(w3.contiguous()), orc._f(out), B,
                                    N, S, C)
        return out

    def interp3(self, dense, sparse, feat):
        if torch.is_grad_enabled() and (dense.requires_grad or sparse.requires_grad or feat.requires_grad):
            return self.interp3_apply(feat, *self.interp3_search(dense, sparse))
        return orc.interp3(dense, sparse, feat)

    def fusion_mlp(self, p1, p2, idx, w1, b1, w2, b2, w3, b3):
        """Unfused restatement of mocopci.py:803-819 with BN already folded into (w,b)."""
        idx = torch.cat(list(idx), dim=-1) if isinstance(idx, (tuple, list)) else idx
        nb = self.group_rows(p2, idx)
        resi = nb - p1.unsqueeze(2)
        x = torch.cat([resi, torch.norm(resi, dim=-1, keepdim=True)], dim=-1)
        for w, b in ((w1, b1), (w2, b2), (w3, b3)):
            x = torch.relu(torch.nn.functional.linear(x, w, b))
        wgt = torch.softmax(x.max(dim=-1)[0], dim=-1)
        return torch.sum(wgt.unsqueeze(-1) * nb, dim=2)

    def cross_pack(self, wpos, bpos, wmlp, bmlp):
        return (wpos, bpos, wmlp, bmlp)

    def cross_volume(self, xyz1, xyz2, points1, points2, idx, packed):
        """Unfused restatement of pointconv_util.py:765-781 (one mlp layer)."""
        idx = torch.cat(list(idx), dim=-1) if isinstance(idx, (tuple, list)) else idx
        wpos, bpos, wmlp, bmlp = packed
        F = torch.nn.functional
        direction = self.group_rows(xyz2, idx) - xyz1.unsqueeze(2)
        g2 = self.group_rows(points2, idx)
        x = F.leaky_relu((g2 + points1.unsqueeze(2)) + F.linear(direction, wpos, bpos), 0.1)
        x = F.leaky_relu(F.linear(x, wmlp, bmlp), 0.1)
        return x.max(dim=2)[0]

    def ptblock_pack(self, *weights):
        return weights

    def cross_layer(self, xyz1, xyz2, points1, points2, idx, wpos, bpos, wmlp, bmlp, packed=None, bmap=None, shared=0):
        if bmap is not None:  # replicated / selected batch: element b of the flagged tensors comes from their element bmap[b]
            m = bmap.long()
            points1 = points1[m] if shared & 1 else points1
            points2 = points2[m] if shared & 2 else points2
            idx = (idx[0][m], idx[1]) if shared & 4 else idx
        return self.cross_volume(xyz1, xyz2, points1, points2, idx, (wpos, bpos, wmlp, bmlp))

    def ptblock_layer(self, xyz, q, k, v, idx, weights, packed=None):
        return self.ptblock_attention(xyz, q, k, v, idx, tuple(weights))

    def ptblock_attention(self, xyz, q, k, v, idx, packed):
        """Unfused restatement of pointT_layer2.py:64-75 (after knn and the q/k/v projections)."""
        F = torch.nn.functional
        wd1, bd1, wd2, bd2, wg1, bg1, wg2, bg2 = packed
        knn_xyz = self.group_rows(xyz, idx)
        kk, vv = self.group_rows(k, idx), self.group_rows(v, idx)
        pos = F.linear(torch.relu(F.linear(xyz.unsqueeze(2) - knn_xyz, wd1, bd1)), wd2, bd2)
        attn = F.linear(torch.relu(F.linear((q.unsqueeze(2) - kk) + pos, wg1, bg1)), wg2, bg2)
        attn = torch.softmax(attn / (kk.shape[-1] ** 0.5), dim=-2)
        return torch.sum(attn * (vv + pos), dim=2)

    def pointconv_agg(self, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2):
        """Unfused restatement of mocopci.py:1218-1266 + :1289-1300 + :1330-1335."""
        F = torch.nn.functional
        B, S, _ = new_xyz.shape
        g_xyz = self.group_rows(s_xyz, idx) - new_xyz.unsqueeze(2)
        new_points = torch.cat([g_xyz, self.group_rows(s_points, idx)], dim=-1)
        w = g_xyz
        for ww, bb in ((w0, b0), (w1, b1), (w2, b2)):
            w = torch.relu(F.linear(w, ww, bb))
        return torch.matmul(new_points.transpose(2, 3), w).reshape(B, S, -1)

    def pointconv_linear_supported(self, d, c_out, k=32, rows=None):
        return True

    def pointconv_linear_pack(self, w, b):
        return None

    def pointconv_linear(self, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2, w, b, slope, packed=None):
        """mocopci.py:1330-1342: the aggregate above, then Linear and LeakyReLU."""
        return self.linear(self.pointconv_agg(s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2), w, b, slope)

    def attention(self, q, kv, heads, scale=None):
        BF, Nq, C = q.shape
        Nk = kv.shape[1]
        hd = C // heads
        qh = q.reshape(BF, Nq, heads, hd).permute(0, 2, 1, 3)
        kvh = kv.reshape(BF, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
        o = torch.nn.functional.scaled_dot_product_attention(qh, kvh[0], kvh[1], scale=scale)
        return o.permute(0, 2, 1, 3).reshape(BF, Nq, C)

    def mlp2_pack(self, w1, b1, w2, b2):
        return None

    def linear_supported(self, xs, n, few_rows=True, policy_rows=None):
        return True

    def linear_pack(self, w, b, ks):
        return None

    def linear(self, xs, w, b=None, slope=1.0, res=None, packed=None, policy_rows=None):
        """Conv1d wrapper of the reference (mocopci.py:1111-1127) generalised: Linear over the concatenated pieces, one-slope
        activation, residual."""
        x = torch.cat(list(xs), dim=-1) if isinstance(xs, (tuple, list)) else xs
        y = torch.nn.functional.linear(x, w, b)
        if slope != 1.0:
            y = torch.where(y > 0, y, y * slope)
        return y if res is None else y + res

    def linear_narrow_supported(self, rows, k, n):
        return n <= 4

    def linear_narrow(self, x, w, b, in_slope):
        return torch.nn.functional.linear(torch.where(x > 0, x, x * in_slope), w, b)

    def mlp2(self, x, w1, b1, w2, b2, slope, res=None, packed=None):
        """Linear, one-slope PReLU, Linear (+ residual): Mlp_T with its affine neighbours folded in, mocopci.py:1558-1565."""
        hid = torch.nn.functional.linear(x, w1, b1)
        out = torch.nn.functional.linear(torch.where(hid > 0, hid, hid * slope), w2, b2)
        return out if res is None else out + res

    def prelu_dropout(self, z, slope, drop_p):
        """Mlp_T's act + drop in a training forward (mocopci.py:1561-1562)."""
        return torch.nn.functional.dropout(torch.nn.functional.prelu(z, slope.reshape(-1)), drop_p, training=True)

    def mlp2_supported(self, cin, hidden, cout):
        return True

    def chamfer(self, x, y, per_sample=False):
        """per_sample: the (B,) values whose mean the loss is (several loss terms evaluated as one batch, mocopci_amd/training.py)."""
        if per_sample or (torch.is_grad_enabled() and (x.requires_grad or y.requires_grad)):  # models/utils.py:36-45 with pytorch3d's defaults
            d = ((x.unsqueeze(2) - y.unsqueeze(1)) ** 2).sum(-1)
            v = d.min(2)[0].mean(1) + d.min(1)[0].mean(1)
            return v if per_sample else v.mean()
        return torch.tensor(orc.chamfer(x.detach(), y.detach()), dtype=torch.float32)
