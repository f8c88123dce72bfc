"""Inference graph of models.m_models.mocopci.MoCoPCI (reference mocopci.py:1062-1097) on the
MI355X point-set operators.

This is the caller of the hot path, re-expressed -- not copied -- on
channel-last tensors: xyz (B,N,3), features (B,N,C).  The point-set half (FPS, KNN, grouping,
3-NN interpolation, cost volumes, PointConv, fusion) goes through mocopci_amd.ops; the dense
half (EI cross-formers, frame attention, MLPs) stays on PyTorch-ROCm dense ops.

State-dict compatibility: the parameter tree is generated from state_dict_spec.json (names,
shapes, dtypes of the reference's 487 state-dict entries, including modules the reference
constructs but never calls: fusion_gru, recurrent0, rf_block0, deconv1_0), so a reference
checkpoint's ['net'] loads with load_state_dict(strict=True).

Differences from the reference graph that do not change out_lst:
  * the two encoder passes run as one 2B batch; forward/backward decoder directions run as one
    2B batch; the three level-0 refinements run as one 3B batch (every layer is per-sample in
    eval mode, so batching is exact);
  * dead work is skipped: the up_feat*_lst upsamples, the third cross() of cross3, the frames
    bookkeeping lists used only by the training loss;
  * identical 3-NN searches on the same (dense, sparse) pair are done once and reused.
Eval-mode semantics only (BatchNorm running stats, dropout/drop-path off), as the goldens are.
"""
import json
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import grad, ops, schedule
from .schedule import Schedule

_SPEC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_dict_spec.json")
LEAKY = 0.1  # mocopci.py:1107, pointconv_util.py:10


def leaky(x):
    return F.leaky_relu(x, LEAKY)


class _Epoch:
    """Assignment counter of ONE model (shared by the nodes of its parameter tree)."""
    __slots__ = ("n",)

    def __init__(self):
        self.n = 0


class _Node(nn.Module):
    """A node of the generated parameter tree.  Assigning a tensor to one of its attributes (net.x.weight = nn.Parameter(..),
    load_state_dict(assign=True)) bumps the epoch of the model it belongs to, which that model's inference cache is keyed on beside
    the version counters.  (Per model: constructing or editing another MoCoPCI leaves this one's cache alone.)"""

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor):
            ref = self.__dict__.get("_epoch_ref")
            if ref is not None:
                ref.n += 1
        super().__setattr__(name, value)


def _dtype(name):
    return getattr(torch, name)


class MoCoPCI(nn.Module):
    T_F = [0.0, 0.41666666666666663, 0.5, 0.5833333333333333, 1.0]  # mocopci.py:824
    T_B = [1.0, 0.5833333333333333, 0.5, 0.41666666666666663, 0.0]  # mocopci.py:825
    # net.train() + forward(train=True): the rates the reference constructs its blocks with (mocopci.py:165-168, :781-783);
    # plain attributes so that a run can change them (0 switches a kind of dropout off)
    drop_rate = 0.05        # Mlp_T / EasyMlp hidden + output dropout, attention proj_drop
    attn_drop_rate = 0.05   # dropout on the softmax matrices of Multi_Frame_Att / Cross_Frame_Att
    drop_path_rate = 0.04   # stochastic depth of Multi_Frame_Att's two residual branches (Cross_Frame_Att: 0)
    BN_MOMENTUM = 0.1       # nn.BatchNorm default
    PAIR_CFA = True         # inference: cross_block3 evaluated once for both decoder directions (it is symmetric in its two frames)
    LANE_MAP = None         # side lanes folded onto fewer HIP streams, e.g. (0, 1, 1, 1, 0, 0) (experiments with several caller streams: tools/two_stream.py)
    NODE_LANES = schedule.NODE_LANES   # which side lane each independent chain of a forward runs on (data: tools/lane_order.py sweeps it)
    SIDE_PROJECTIONS = True # the decoder's feature-only chains on side lanes (A/B switch: False runs schedule.SIDE_PROJECTION_NODES inline)
    FUSE_POINTCONV = True  # PointConv's Linear inside the grouped kernel where it is built for the shape (A/B switch)
    FOLD_EI = True          # inference: EI cross-formers in their folded 9-launch form (A/B: tools/step_time.py net.FOLD_EI=0)

    def __init__(self):
        super().__init__()
        self.__dict__["_epoch_ref"] = _Epoch()
        with open(_SPEC_PATH) as fh:
            self._spec = json.load(fh)
        for name, meta in self._spec.items():
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    child = _Node()
                    child.__dict__["_epoch_ref"] = self.__dict__["_epoch_ref"]
                    node.add_module(p, child)
                node = getattr(node, p)
            t = torch.zeros(meta["shape"], dtype=_dtype(meta["dtype"]))
            if meta["buffer"]:
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t))
        self._cache = None
        self._live = None  # training forward: {name: live parameter / buffer}; derived tensors are then rebuilt, not cached
        self._mode = None  # training forward in net.train() mode: (drop, attn_drop, drop_path) -- BatchNorm then uses batch statistics
        self.eval()

    def __setattr__(self, name, value):
        if isinstance(value, torch.Tensor):  # a tensor assigned on the root module itself
            ref = self.__dict__.get("_epoch_ref")
            if ref is not None:
                ref.n += 1
        super().__setattr__(name, value)

    def _mark(self, name):
        """Timeline marker (tools/step_sections.py sets self._marks = []): an event on the current stream; nothing otherwise."""
        marks = self.__dict__.get("_marks")
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))

    # ---- parameter access ---------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._cache = None
        self.__dict__.pop("_state_tensors", None)  # .to() / .cuda() replace the parameter tensors
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        try:
            return super().load_state_dict(*a, **k)
        finally:
            self.invalidate()  # assign=True replaces the parameter tensors themselves

    def invalidate(self):
        """Drop everything cached from the parameters (folded BatchNorms, packed kernel operands).  Needed by hand only after
        writes the version counters do not see: `p.data.copy_()` / `p.data.mul_()` (EMA or weight-loading code that goes through
        .data).  optimizer.step(), in-place ops on the parameters, load_state_dict (also assign=True), re-assigned parameters and
        .to() / .cuda() are detected."""
        self._cache = None
        self.__dict__.pop("_state_tensors", None)

    def _state_version(self):
        """Changes whenever a parameter or buffer is written in place (optimizer.step(), a training forward's running statistics,
        copy_ / load) or REPLACED by another tensor (load_state_dict(assign=True), net.x.weight = nn.Parameter(...)) -- everything
        cached from them (folded BatchNorms, packed kernel operands) is then stale (assignments bump this model's epoch)."""
        ts = self.__dict__.get("_state_tensors")
        epoch = self.__dict__["_epoch_ref"].n
        if ts is None or ts[0] != epoch:  # the module tree is fixed after construction: walked again only after an assignment
            ts = self.__dict__["_state_tensors"] = (epoch, [*self.parameters(), *self.buffers()])
        v = 0
        for t in ts[1]:
            v += t._version
        return (v, ts[0])

    def _check_cache(self):
        """Once per forward: drop the inference cache if any parameter / buffer changed since it was built."""
        if self._cache is not None and self._cache.get(("state_version",)) != self._state_version():
            self.invalidate()

    def _params(self):
        if self._live is not None:
            return self._live
        if self._cache is None:
            c = {}
            for n, p in self.named_parameters():
                c[n] = p.detach()
            for n, b in self.named_buffers():
                c[n] = b
            c[("state_version",)] = self._state_version()
            self._cache = c
            self._time_cache = {}
        return self._cache

    def derived(self, key, fn):
        """A tensor computed from parameters (folded BatchNorm, packed kernel operands, ...): cached for inference, rebuilt from
        the live parameters -- so gradients reach them -- in a training forward."""
        if self._live is not None:
            return fn()
        P = self._params()
        if key not in P:
            P[key] = fn()
        return P[key]

    def W(self, name):
        """weight of a 1x1 conv / linear as (out,in)."""
        w = self._params()[name + ".weight"]
        return w.reshape(w.shape[0], -1)

    def Bv(self, name):
        return self._params().get(name + ".bias")

    def lin(self, x, name, slope=1.0, res=None, like_rows=None):
        """Linear / 1x1 conv `name` over the last axis with a one-slope activation (1.0 = none) and a residual: ONE kernel
        (ops.linear) where the backend takes the shape -- the tall per-point layers -- and the BLAS chain otherwise.
        like_rows: x holds SOME of the rows of a product that another form of the same forward computes in full (the sampled rows
        of a PointConvD whose every candidate row the speculative form evaluates): take the kernel that product takes, so that both
        forms return the same bits whatever the batch size (the few-row kernels sum K in another order)."""
        w, b = self.W(name), self.Bv(name)
        be = ops.backend()
        pieces = isinstance(x, (tuple, list))  # the pieces of a concatenation along the channel axis: read in place by the kernel
        like = {} if like_rows is None or self._live is not None else {"policy_rows": like_rows}
        if be.linear_supported(list(x) if pieces else x, w.shape[0], few_rows=self._live is None, **like):
            ks = [t.shape[-1] for t in x] if pieces else [x.shape[-1]]
            packed = None if self._live is not None else self.derived(("lin_pack", be.name, name, tuple(ks)), lambda: be.linear_pack(w, b, ks))
            return be.linear(list(x) if pieces else x, w, b, slope, res, packed=packed, **like)
        if pieces:
            x = torch.cat(list(x), dim=-1)
        if self._live is not None and x.is_cuda and torch.is_grad_enabled() and x.numel() // x.shape[-1] >= 16384 and 0.0 <= slope <= 1.0:
            y = ops.plain_linear(x, w, b, slope)   # training, tall input: the library's forward with the streaming backward kernels
            return y if res is None else y + res
        y = F.linear(x, w, b)
        if slope != 1.0:
            y = F.leaky_relu(y, slope)
        return y if res is None else y + res

    def conv1d_block(self, x, name, like_rows=None):
        """Conv1d wrapper of the reference (mocopci.py:1111-1127): 1x1 conv + LeakyReLU(0.1)."""
        return self.lin(x, name + ".composed_module.0", slope=LEAKY, like_rows=like_rows)

    def bn_eval(self, x, name, eps):
        """BatchNorm in eval mode on a channel-last tensor: one fused multiply-add with cached (scale, shift)."""
        P = self._params()

        def fold():
            scale = P[name + ".weight"] * torch.rsqrt(P[name + ".running_var"] + eps)
            return scale.contiguous(), (P[name + ".bias"] - P[name + ".running_mean"] * scale).contiguous()
        scale, shift = self.derived(("bn", name, eps), fold)
        return torch.addcmul(shift, x, scale)

    def bn_batch(self, x, name, eps):
        """BatchNorm in TRAINING mode applied x.shape[0] times in sequence: call g normalises x[g] (any shape, channels last) with
        the statistics of x[g] alone and then updates the running statistics, exactly what the reference's per-sample loops
        (`for i, x in enumerate(xs): self.norm1(x)`, mocopci.py:503-506, :554-561) and its three fusion calls do with
        nn.BatchNorm1d / 2d: biased variance for the normalisation, unbiased for the running estimate, momentum 0.1."""
        P = self._params()
        G, C = x.shape[0], x.shape[-1]
        flat = x.reshape(G, -1, C)
        n = flat.shape[1]
        mean = flat.mean(dim=1)
        var = flat.var(dim=1, unbiased=False)
        with torch.no_grad():
            # G momentum updates in sequence, r <- (1 - m) r + m s_g, in closed form: r <- (1 - m)^G r + sum_g m (1 - m)^(G-1-g) s_g
            # (one matrix-vector product per statistic instead of 2 G tiny in-place kernels)
            rm, rv, m = P[name + ".running_mean"], P[name + ".running_var"], self.BN_MOMENTUM
            coef, keep = self.ema_coefficients(G, m, x.device)
            rm.mul_(keep).add_(coef @ mean.detach())
            rv.mul_(keep).add_(coef @ var.detach(), alpha=n / (n - 1))
            P[name + ".num_batches_tracked"].add_(G)
        shape = (G,) + (1,) * (x.dim() - 2) + (C,)
        scale = P[name + ".weight"] * torch.rsqrt(var + eps)
        return (x - mean.reshape(shape)) * scale.reshape(shape) + P[name + ".bias"]

    @staticmethod
    def halves(t):
        """(t[:h], t[h:]) of a tensor stacked along axis 0.  In a training graph as ONE unbind of the (2, h, ...) view: its backward is
        one stack, two slices' are two zero-filled tensors, two copies and an add."""
        h = t.shape[0] // 2
        if t.requires_grad and torch.is_grad_enabled():
            return t.reshape(2, h, *t.shape[1:]).unbind(0)
        return t[:h], t[h:]

    def ema_coefficients(self, G, m, device):
        """(coef (G,), keep) of G momentum updates in sequence: r <- keep r + coef @ s; the device tensor is built once (a host-to-device
        copy inside a forward would stall the stream)."""
        key = ("ema", G, m, str(device))
        cache = self.__dict__.setdefault("_time_cache", {})
        if key not in cache:
            cache[key] = torch.tensor([m * (1.0 - m) ** (G - 1 - g) for g in range(G)], dtype=torch.float32, device=device)
        return cache[key], (1.0 - m) ** G

    def norm(self, x, name, eps):
        """The block's BatchNorm on (samples, ..., C): batch statistics per sample in a net.train() training forward, the running
        statistics otherwise."""
        return self.bn_batch(x, name, eps) if self._mode is not None else self.bn_eval(x, name, eps)

    def dropout(self, x, p):
        return F.dropout(x, p, training=True) if self._mode is not None and p > 0.0 else x

    def drop_path(self, x):
        """timm DropPath on the leading two axes (sample, frame): the reference applies it to one sample's (frames, C, N) stack,
        so every frame of every sample draws its own keep / drop (mocopci.py:559, :562)."""
        p = self._mode[2] if self._mode is not None else 0.0
        if p <= 0.0:
            return x
        keep = torch.empty(x.shape[:2] + (1,) * (x.dim() - 2), device=x.device, dtype=x.dtype).bernoulli_(1.0 - p)
        return x * (keep / (1.0 - p))

    TRAIN_PYRAMID_LANE = True   # training forwards: the encoder's FPS chain on side lane 0, beside the level-0 layers
    _train_lane0 = False
    TRAIN_LANES = (0, 4, 5, 6)
    SDPA_DROPOUT = True         # net.train() on the GPU: attention dropout inside the library's fused attention kernel
    CHECKPOINT_BYTES = 1 << 30  # net.train() forwards: unfused blocks whose intermediates exceed this are recomputed in the backward, in chunks of about this size

    def attend(self, q, kv, heads, scale=None):
        """softmax(q k^T scale) v per head; with attention dropout (net.train()) the probabilities are materialised -- 10.7 GB per
        tensor at level 1 of the B = 8, N = 8192 step, so large calls run in chunks of batch elements under
        torch.utils.checkpoint: nothing of size heads x Nq x Nk outlives a chunk, the backward rebuilds one chunk at a time (the
        dropout masks come back through the generator state the checkpoint restores)."""
        p = self._mode[1] if self._mode is not None else 0.0
        if p <= 0.0:
            return ops.backend().attention(q, kv, heads, scale=scale)
        BF, Nq, C = q.shape
        Nk, hd = kv.shape[1], C // heads
        sc = hd ** -0.5 if scale is None else scale
        if q.is_cuda and self.SDPA_DROPOUT and hd in (8, 16):
            # head widths 8 / 16: this repo's attention kernels with the mask generated inside (forward and backward regenerate it
            # from one seed drawn from torch's generator); nothing of size heads x Nq x Nk exists
            return ops.backend().attention(q, kv, heads, scale=sc, dropout_p=p)
        if q.is_cuda and self.SDPA_DROPOUT:
            # the library's fused attention draws the dropout mask inside the kernel (forward and backward from one counter-based
            # stream seeded by torch's generator): same distribution as softmax -> F.dropout -> matmul, nothing of size
            # heads x Nq x Nk exists.  The CPU (oracle backend) path below keeps the explicit form.
            qh = q.reshape(BF, Nq, heads, hd).permute(0, 2, 1, 3)
            kvh = kv.reshape(BF, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
            o = F.scaled_dot_product_attention(qh, kvh[0], kvh[1], dropout_p=p, scale=sc)
            return o.permute(0, 2, 1, 3).reshape(BF, Nq, C)

        def dense(q_, kv_):
            n = q_.shape[0]
            qh = q_.reshape(n, Nq, heads, hd).permute(0, 2, 1, 3)
            kvh = kv_.reshape(n, Nk, 2, heads, hd).permute(2, 0, 3, 1, 4)
            attn = torch.softmax((qh @ kvh[0].transpose(-2, -1)) * sc, dim=-1)
            return (F.dropout(attn, p, training=True) @ kvh[1]).permute(0, 2, 1, 3).reshape(n, Nq, C)
        per_element = heads * Nq * Nk * 4
        if BF * per_element <= self.CHECKPOINT_BYTES or not torch.is_grad_enabled():
            return dense(q, kv)
        from torch.utils.checkpoint import checkpoint
        step = max(1, self.CHECKPOINT_BYTES // per_element)
        return torch.cat([checkpoint(dense, q[i:i + step], kv[i:i + step], use_reentrant=False) for i in range(0, BF, step)], dim=0)

    # ---- point-set layers ---------------------------------------------------------------
    def pointconv(self, prefix, s_xyz, new_xyz, s_points, nsample=32, idx=None, like_rows=None):
        """PointConv / PointConvD body after sampling (mocopci.py:1315-1346, :1362-1396; group /
        group_query :1218-1266; WeightNet :1289-1300).  idx: the (B,S,nsample) neighbour lists when the caller
        already has them (see sampled_neighbours).  like_rows: the centres are a subset of that many candidate centres which
        another form of the same forward evaluates all of (see lin): same kernels, same bits."""
        be = ops.backend()
        B, S, _ = new_xyz.shape
        if idx is None:
            idx = be.knn(new_xyz, s_xyz, nsample)
        wn = [t for i in range(3) for t in (self.W(f"{prefix}.weightnet.mlp_convs.{i}"), self.Bv(f"{prefix}.weightnet.mlp_convs.{i}"))]
        w, b = self.W(prefix + ".linear"), self.Bv(prefix + ".linear")
        if self.FUSE_POINTCONV and self._live is None and be.pointconv_linear_supported(s_points.shape[-1], w.shape[0], idx.shape[-1],
                                                                                        rows=B * S if like_rows is None else like_rows):
            # levels 0 / 1 and the refinement stage: the (B,S,(3+D)*8) aggregate never leaves the compute unit
            packed = self.derived(("lin_pack", be.name, prefix + ".linear", (w.shape[1],)), lambda: be.pointconv_linear_pack(w, b))
            return be.pointconv_linear(s_xyz, new_xyz, s_points.contiguous(), idx, *wn, w, b, LEAKY, packed=packed)
        agg = be.pointconv_agg(s_xyz, new_xyz, s_points.contiguous(), idx, *wn)      # (B,S,(3+D)*8)
        return self.lin(agg, prefix + ".linear", slope=LEAKY, like_rows=like_rows)

    def fps_gather(self, xyz, npoint, return_idx=False):
        """furthest_point_sample + index_points_gather (mocopci.py:1378-1379)."""
        be = ops.backend()
        if xyz.requires_grad and torch.is_grad_enabled():  # training: the gather carries the gradient to the (warped) coordinates
            sel = be.fps(xyz, npoint)
            pts = be.group_rows(xyz, sel)
        else:
            sel, pts = be.fps(xyz, npoint, with_points=True)          # one launch: the sample and its coordinates
        return (pts, sel) if return_idx else pts

    @staticmethod
    def sampled_neighbours(idx_self, sel):
        """Neighbour lists of FPS-sampled points among the cloud they were sampled from.  A sampled point IS a point of that
        cloud (bit-identical coordinates), so its K nearest in the cloud are the row of the cloud's self-search at its index:
        knn(cloud[sel], cloud, K) == knn(cloud, cloud, K)[sel], same distance form, same (distance, index) order.  The rows
        are moved as raw 32-bit words."""
        return ops.backend().group_rows(idx_self.view(torch.float32), sel).view(torch.int32)

    def side_stream(self, device, which=0):
        """Extra HIP streams beside the caller's.  0: the encoder's serial FPS chains (one workgroup per batch element, latency-
        bound), which overlap with the KNN / PointConv work of the main stream.  1-3: branches that depend only on encoder
        features (schedule.NODE_LANES).  4: the level-0 self search.  5: the refinement stage's FPS (a lane of its own: the NEXT batch's
        pyramid may already be queued on lane 0).  One set per caller stream, so forwards issued on different streams stay
        independent.  CPU backends run inline."""
        # a training forward runs on one stream (autograd replays it in order) -- except the lanes of TRAIN_LANES, whose nodes only
        # produce INDICES (no gradient, no backward node): the sampling pyramids of the inputs and the ground truth, the level-0 self
        # search, the refinement stage's sampling (TRAIN_PYRAMID_LANE; off when the inputs themselves ask for a gradient)
        if device.type != "cuda" or (self._live is not None and not (which in self.TRAIN_LANES and self._train_lane0)):
            return None
        if self.LANE_MAP is not None:
            which = self.LANE_MAP[which]
        key = (device.index, torch.cuda.current_stream(device).stream_id, which)
        sides = self.__dict__.setdefault("_sides", {})
        if key not in sides:
            sides[key] = torch.cuda.Stream(device=device)
        return sides[key]

    def issue_inputs_only(self, sched, xyz_fn, after=None):
        """Everything of a forward that depends on nothing but the inputs, as schedule nodes: the channel-last layout ("xyz"), the four
        FPS levels (mocopci.py:445-463; ("pc", level)), the sampled clouds in the decoder's "other frame" arrangement ("swap_pc") and
        the level-0 self search.  after: see Schedule.run (prefetch() issues these behind the inputs' event, a plain forward behind
        the caller's stream)."""
        sched.run("xyz", xyz_fn, after=after)
        chained = None if after is None else ()   # behind the layout: stream order on the same lane / inline

        def level(lvl, npoint):
            src = sched.peek("xyz") if lvl == 1 else sched.peek(("pc", lvl - 1))[0]
            return self.fps_gather(src, npoint, return_idx=True)        # (points, indices)
        for lvl, npoint in enumerate((2048, 512, 256, 64), start=1):
            sched.run(("pc", lvl), lambda lvl=lvl, npoint=npoint: level(lvl, npoint), after=chained)

        def swapped():  # three small copies that ride along here, off the main stream and off the lanes the feature branches queue on
            half = sched.peek("xyz").shape[0] // 2
            return {lvl: torch.cat([sched.peek(("pc", lvl))[0][half:], sched.peek(("pc", lvl))[0][:half]], dim=0) for lvl in (1, 2, 3)}
        sched.run("swap_pc", swapped, after=chained)
        # the level-0 self search (0.6 ms with its sorted cloud) on a lane of its own: it neither delays the sampling chain nor waits for it
        xyz = sched.peek("xyz")
        sched.run("self_search", lambda: ops.backend().knn(xyz, xyz, 32), reads=(xyz,), after=None if after is None else (sched.event("xyz"),))

    def run_encoder(self, xyz, sched=None, speculate=None):
        """PointConvEncoder.forward (mocopci.py:438-468), color == xyz.  The sampling pyramid and the level-0 self search depend only
        on the coordinates: they are schedule nodes, issued here unless the caller already did (prefetch(): they then ran under the
        PREVIOUS call's tail, and level 1 is not speculated).  A level's cloud is fetched where it is first read."""
        p = "encoder."
        standalone = sched is None   # the encoder alone (tests compare its pyramid / features): none of the decoder's early nodes
        if standalone:
            sched = Schedule(self, xyz.device)
        if speculate is None:
            speculate = self._live is None and not sched.has("xyz") and sched.lane(("pc", 1)) is not None   # inference only
        if not sched.has("xyz"):
            self.issue_inputs_only(sched, lambda: xyz)
        pc = lambda lvl: sched.get(("pc", lvl))[0]

        self._mark("enc start")
        f0 = self.conv1d_block(xyz, p + "level0_lift")
        idx0 = sched.get("self_search")
        f0 = self.pointconv(p + "level0", xyz, xyz, f0, idx=idx0)
        f0_1 = self.conv1d_block(f0, p + "level0_1")
        if speculate:
            # The main stream would now wait ~0.6 ms for the level-1 sampling.  Level 1 is a PointConvD whose centres are
            # SAMPLED points of xyz and whose neighbours are rows of the level-0 self search (sampled_neighbours), so its
            # output for every candidate centre can be computed before the sample is known -- 4x the work, on an otherwise
            # idle chip -- and the sampled rows gathered afterwards (same per-row arithmetic, same kernels: lin(like_rows=)).
            f1_all = self.conv1d_block(self.pointconv(p + "level1", xyz, xyz, f0_1, idx=idx0), p + "level1_0")
            pc1, sel1 = sched.get(("pc", 1))
            f1 = ops.backend().group_rows(f1_all, sel1)
        else:
            pc1, sel1 = sched.get(("pc", 1))
            # level 1 searches the 32 nearest of pc1 = xyz[sel1] in xyz: rows of the level-0 self search
            # (kernels as for all of xyz's rows: the speculative form above computes those, and both must give the same bits)
            like = xyz.shape[0] * xyz.shape[1]
            f1 = self.pointconv(p + "level1", xyz, pc1, f0_1, idx=self.sampled_neighbours(idx0, sel1), like_rows=like)
            f1 = self.conv1d_block(f1, p + "level1_0", like_rows=like)
        B = xyz.shape[0] // 2
        d = "multi_frame_inference."
        # the two frames swapped; in a training graph as a roll (its backward is a roll: two slices' are two zero-filled tensors, two copies
        # and an add)
        swap = lambda t: torch.roll(t, B, 0) if t.requires_grad else torch.cat([t[B:], t[:B]], dim=0)

        def branches(lvl, f):  # decoder work that needs nothing but this level's encoder features (both frames stacked)
            if standalone:
                return
            sched.run(("swap_f", lvl), lambda: swap(f))   # the "other frame" arrangement the decoder reads
            sched.run(("fus", lvl), lambda: (lambda g: torch.cat([g, g], dim=0))(self.ei_crossformer(d + f"ei{lvl}", *self.halves(f), stacked=f)))
            def cos():  # both directions of the feature-space search: the backward one is the forward one with its halves swapped
                i12 = ops.backend().knn_cosine(f, sched.get(("swap_f", lvl)), 16)   # the swapped copy made by the node above (same lane in the shipped table: no wait)
                return i12, swap(i12)
            sched.run(("cos", lvl), cos)

        branches(1, f1)
        if not standalone:
            sched.run("i3_01", lambda: ops.backend().interp3_search(xyz, pc1))
        f1_2 = self.conv1d_block(f1, p + "level1_1")
        self._mark("enc level1 done")
        pc2 = pc(2)
        f2 = self.pointconv(p + "level2", pc1, pc2, f1_2)
        f2 = self.conv1d_block(f2, p + "level2_0")
        branches(2, f2)
        f2_3 = self.conv1d_block(f2, p + "level2_1")
        self._mark("enc level2 done (before need 3)")
        pc3 = pc(3)
        f3 = self.pointconv(p + "level3", pc2, pc3, f2_3)
        f3 = self.conv1d_block(f3, p + "level3_0")
        f3_4 = self.conv1d_block(f3, p + "level3_1")
        self._mark("enc level3 done (before need 4)")
        pc4 = pc(4)
        f4 = self.pointconv(p + "level4", pc3, pc4, f3_4)
        self._mark("enc end")
        return [xyz, pc1, pc2, pc3, pc4], [f0, f1, f2, f3, f4]

    def cross(self, xyz1, xyz2, points1, points2, knn1, knn2, pos, mlp, sorted_p3d, idx_c=None, bmap=None, shared=0):
        """cost-volume cross() (pointconv_util.py:750-781; :894-922 with pytorch3d knn_points;
        :1126-1161).  16 feature-cosine neighbours then 16 xyz neighbours of set 2 per point of
        set 1, LeakyReLU(g2 + p1 + pos(dxyz)), 1x1 convs, max over the 32 neighbours."""
        be = ops.backend()
        if idx_c is None:  # depends only on the two feature sets: callers that reuse them pass it in
            idx_c = be.knn_cosine(knn1, knn2, 16)
        if sorted_p3d:
            # pointconv_util.py:910-911: knn_points(xyz2, xyz1) -- QUERY = xyz2, REF = xyz1 -- and the
            # resulting indices (into xyz1) are then used to index set 2.  Reproduced as is.
            idx_p = be.knn(xyz2, xyz1, 16, mode=ops.MCP_DIST_DIRECT)
        else:
            idx_p = be.knn(xyz1, xyz2, 16)
        idx = (idx_c.contiguous(), idx_p)                                 # the two 16-neighbour lists, read in place by the kernel
        # every cross() MoCoPCI builds has one D -> D mlp layer with D in {64, 128, 256} (pointconv_util.py:735-748)
        assert len(mlp) == 1 and points2.shape[-1] == points1.shape[-1]
        # bmap / shared: points1, points2 and idx_c hold the UNREPLICATED batch of multiframe_attention's three iterations
        conv = mlp[0] + ".composed_module.0"
        w = (self.W(pos), self.Bv(pos), self.W(conv), self.Bv(conv))
        # the layer's weights in the kernel's operand layout, built once (inference); a training forward packs the live weights
        packed = None if self._live is not None else self.derived(("cross_pack", be.name, pos, conv), lambda: be.cross_pack(*w))
        return be.cross_layer(xyz1, xyz2, points1, points2, idx, *w, packed=packed, bmap=bmap, shared=shared)

    def interp(self, dense, sparse, feat, cache=None, key=None):
        """UpsampleFlow (mocopci.py:1485-1502) with search reuse on a keyed (dense, sparse) pair."""
        be = ops.backend()
        if cache is None:
            return be.interp3(dense, sparse, feat)
        if key not in cache:
            cache[key] = be.interp3_search(dense, sparse)
        idx3, w3 = cache[key]
        return be.interp3_apply(feat, idx3, w3)

    def interp_prepare(self, dense, sparse, cache, key):
        """Make the keyed 3-NN search of interp() now, on the current stream (a side lane is about to read it as well)."""
        if key not in cache:
            cache[key] = ops.backend().interp3_search(dense, sparse)

    def interp_flows(self, dense, sparse, flows, cache, key):
        """The three per-frame flow upsamples (mocopci.py:870-878, :936-944) as ONE interpolation: flows (B,3,S,3) are
        laid side by side as 9 channels (same 3-NN and weights for all of them).  Returns them stacked frame-major,
        (3*B,N,3) -- the arrangement multiframe_attention batches its three iterations in."""
        B, R, S, _ = flows.shape
        up = self.interp(dense, sparse, flows.permute(0, 2, 1, 3).reshape(B, S, R * 3), cache, key)      # (B,N,9)
        return up.reshape(B, -1, R, 3).permute(2, 0, 1, 3).reshape(R * B, -1, 3)

    def warp(self, xyz1, xyz2, flow1):
        """PointWarping.forward (mocopci.py:1458-1482)."""
        return xyz2 - ops.backend().interp3(xyz2, xyz1 + flow1, flow1)

    # ---- dense blocks (callers; PyTorch dense ops) ---------------------------------------
    def cross_attention(self, prefix, x, c, heads=8):
        """CrossAttention.forward (mocopci.py:72-86)."""
        o = ops.backend().attention(self.lin(x, prefix + ".q"), self.lin(c, prefix + ".kv"), heads)
        return self.lin(o, prefix + ".proj")

    def layer_norm(self, x, name):
        P = self._params()
        return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], 1e-6)

    def ei_crossformer(self, prefix, x1, x2, stacked=None):
        """EI_Crossformer.forward (mocopci.py:147-151): Injector(x1,x2) | Extractor(x2,x1) -> pj.  stacked: the (2B,N,C) tensor
        whose halves x1 and x2 are, when the caller has it (inference then takes the folded form below)."""
        if self._live is None and stacked is not None and self.FOLD_EI:
            return self.ei_crossformer_folded(prefix, stacked)
        P = self._params()
        i, e = prefix + ".injector", prefix + ".extractor"
        res1 = P[i + ".gamma"] * self.cross_attention(i + ".attn", self.layer_norm(x1, i + ".query_norm"),
                                                      self.layer_norm(x2, i + ".feat_norm"))
        q = x2 + self.cross_attention(e + ".attn", self.layer_norm(x2, e + ".query_norm"), self.layer_norm(x1, e + ".feat_norm"))
        h = self.layer_norm(q, e + ".ffn_norm")
        res2 = self.lin(F.gelu(self.lin(h, e + ".ffn.fc1")), e + ".ffn.fc2")
        return F.linear(torch.cat([res1, res2], dim=-1), self.W(prefix + ".pj"))

    def ei_crossformer_folded(self, prefix, f, heads=8):
        """EI_Crossformer (mocopci.py:58-151) in inference, 26 launches folded to 9.  f (2B,N,C) = [x1; x2].
        Everything affine is folded once into neighbouring weights (cached): the four LayerNorms in front of the q / kv projections
        keep only their normalisation -- ONE call on the stacked batch -- and their scale / shift move into the projections, which
        become one batched C -> 3C map per half ([q_i | kv_e] from x1, [q_e | kv_i] from x2); both attentions are ONE launch on the
        stacked batch (queries of half h read keys / values of the other half: kv batch rotation); the Injector's proj, gamma and
        its half of pj collapse into one matrix A, the Extractor's ffn.fc2 and its half of pj into Bm; ffn_norm keeps its
        normalisation, fused with the residual sum in front of it, and its affine part moves into ffn.fc1:
            O = attention(...)                      o_i = O[:B], o_e = O[B:]
            h = normalise(x2 + o_e Wp_e^T + bp_e)   g = gelu(h W1'^T + b1')
            out = o_i A^T + g Bm^T + c"""
        be = ops.backend()
        B2, N, C = f.shape
        B = B2 // 2
        P = self._params()

        def fold():
            i, e = prefix + ".injector", prefix + ".extractor"
            ln = lambda n: (P[n + ".weight"], P[n + ".bias"])
            zero = lambda w: torch.zeros(w.shape[0], device=w.device, dtype=w.dtype)
            wb = lambda n: (self.W(n), self.Bv(n) if self.Bv(n) is not None else zero(self.W(n)))
            (wq_i, bq_i), (wkv_i, bkv_i), (wp_i, bp_i) = wb(i + ".attn.q"), wb(i + ".attn.kv"), wb(i + ".attn.proj")
            (wq_e, bq_e), (wkv_e, bkv_e), (wp_e, bp_e) = wb(e + ".attn.q"), wb(e + ".attn.kv"), wb(e + ".attn.proj")
            (g_iq, b_iq), (g_if, b_if), (g_eq, b_eq), (g_ef, b_ef), (g_fn, b_fn) = (ln(i + ".query_norm"), ln(i + ".feat_norm"), ln(e + ".query_norm"),
                                                                                     ln(e + ".feat_norm"), ln(e + ".ffn_norm"))
            lin_ln = lambda w, b, g, sh: (w * g[None, :], w @ sh + b)        # Linear(LayerNorm affine(x)) as one affine map of the normalised x
            (wa_q, ba_q), (wa_kv, ba_kv) = lin_ln(wq_i, bq_i, g_iq, b_iq), lin_ln(wkv_e, bkv_e, g_ef, b_ef)   # from x1: Injector query, Extractor feat
            (wb_q, bb_q), (wb_kv, bb_kv) = lin_ln(wq_e, bq_e, g_eq, b_eq), lin_ln(wkv_i, bkv_i, g_if, b_if)   # from x2: Extractor query, Injector feat
            w_in = torch.stack([torch.cat([wa_q, wa_kv], 0).t(), torch.cat([wb_q, wb_kv], 0).t()]).contiguous()        # (2, C, 3C)
            b_in = torch.stack([torch.cat([ba_q, ba_kv]), torch.cat([bb_q, bb_kv])]).unsqueeze(1).contiguous()         # (2, 1, 3C)
            (w1, b1), (w2, b2) = wb(e + ".ffn.fc1"), wb(e + ".ffn.fc2")
            w1f, b1f = lin_ln(w1, b1, g_fn, b_fn)
            wpj = self.W(prefix + ".pj")
            wpj1, wpj2 = wpj[:, :C], wpj[:, C:]
            gamma = P[i + ".gamma"]
            a = (wpj1 * gamma[None, :]) @ wp_i                                                                         # (C, C): o_i -> out
            bm = wpj2 @ w2                                                                                             # (C, H): g -> out
            const = wpj1 @ (gamma * bp_i) + wpj2 @ b2
            if self.Bv(prefix + ".pj") is not None:
                const = const + self.Bv(prefix + ".pj")
            return (w_in, b_in, wp_e.t().contiguous(), bp_e.contiguous(), w1f.t().contiguous(), b1f.contiguous(), a.t().contiguous(),
                    bm.t().contiguous(), const.contiguous())
        w_in, b_in, wpe_t, bp_e, w1_t, b1, a_t, bm_t, const = self.derived(("ei_fold", prefix), fold)
        xh = be.add_layernorm(f, eps=1e-6)                                                    # (2B,N,C)
        # [q | k | v] per half: two GEMMs with the bias in their epilogue, written into the halves of one buffer (a batched product
        # with a broadcast bias first materialises the bias at full size: a 35 us copy at level 1)
        xh2 = xh.reshape(2, B * N, C)
        if B * N >= 8192:
            y = torch.empty((2, B * N, 3 * C), dtype=f.dtype, device=f.device)
            torch.addmm(b_in[0, 0], xh2[0], w_in[0], out=y[0])
            torch.addmm(b_in[1, 0], xh2[1], w_in[1], out=y[1])
        else:  # few rows: the batched product fills the chip better than two small GEMMs in a row, and the bias copy is small
            y = torch.baddbmm(b_in, xh2, w_in)
        y = y.reshape(B2, N, 3 * C)
        o = be.attention_rot(y[..., :C], y[..., C:2 * C], y[..., 2 * C:], heads, B)           # (2B,N,C): [:B] Injector, [B:] Extractor
        o_i, o_e = o[:B].reshape(B * N, C), o[B:].reshape(B * N, C)
        h = be.add_layernorm(f[B:].reshape(B * N, C), o_e @ wpe_t, bp_e, eps=1e-6)
        g = F.gelu(torch.addmm(b1, h, w1_t))
        out = torch.addmm(const, o_i, a_t)
        out.addmm_(g, bm_t)
        return out.reshape(B, N, C)

    def folded_tail(self, fc2, mapping):
        """mapping(fc2(h)) as ONE affine map: W = Wmap Wfc2, b = Wmap bfc2 + bmap (cached).  Used where only the 3-channel
        flow is read downstream and the block's feature output is not (inference)."""
        def fold():
            wm, bm = self.W(mapping), self.Bv(mapping)
            return (wm @ self.W(fc2)).contiguous(), (wm @ self.Bv(fc2) + bm).contiguous()
        return self.derived(("tail", fc2, mapping), fold)

    def cross_frame_att(self, prefix, x, feats=True):
        """Cross_Frame_Att.forward (mocopci.py:499-522) batched over samples.  x (B,2,N,C) holds the two
        frames' features; the block's attention runs 4 heads of width C and sums over the two frames,
        so its 4 head slots come out as 4 'frames' (mocopci.py:619-621); the first is dropped."""
        B, Fr, N, C = x.shape
        P = self._params()
        drop = self._mode[0] if self._mode is not None else 0.0
        xn = self.norm(x, prefix + ".norm1", 1e-5)                                # per sample over its two frames (mocopci.py:505)
        xr = torch.flip(xn, dims=[1])
        a = prefix + ".attn_feats"
        # head slot 0 is the dropped one and nothing after the attention mixes slots: project only heads 1..3
        def heads():
            wq, bq, wkv, bkv = self.W(a + ".q"), self.Bv(a + ".q"), self.W(a + ".kv"), self.Bv(a + ".kv")
            sl = lambda t: None if t is None else torch.cat([t[C:4 * C], t[5 * C:8 * C]], dim=0).contiguous()
            return wq[C:].contiguous(), None if bq is None else bq[C:].contiguous(), sl(wkv), sl(bkv)
        wq, bq, wkv, bkv = self.derived(("cfa_heads", prefix), heads)
        o = self.attend(F.linear(xn, wq, bq).reshape(B * Fr, N, 3 * C), F.linear(xr, wkv, bkv).reshape(B * Fr, N, 6 * C), 3,
                        scale=C ** -0.5)                                          # (B*2,N,3C): 3 head slots, each C wide
        o = self.dropout(self.lin(o.reshape(B, Fr, N, 3, C).sum(dim=1).transpose(1, 2), a + ".proj"), drop)   # (B,3,N,C)
        t = prefix + ".trans_block_2"
        if not feats and drop <= 0.0:  # only the flows are read (MultiFrameEstimatier.forward never uses cross_block3's features)
            return None, self.mlp_t(t, o, tail=prefix + ".mapping_xyz")
        xa = self.mlp_t(t, o, drop=drop)
        frames = self.lin(xa, prefix + ".mapping_xyz")
        return xa, frames                                                         # (B,3,N,C), (B,3,N,3)

    def cross_frame_att_pair(self, prefix, new):
        """cross_block3 on both decoder directions at once, inference, flows only.  new (2B,N,C): [:B] = feat1_new, [B:] = feat2_new.
        The reference calls the block with the frame pair (feat1_new, feat2_new) for the forward direction and with the swapped
        pair for the backward one (mocopci.py:853-856).  The block is symmetric in its two frames: its attention pairs frame f
        with frame 1-f and SUMS over f (mocopci.py:619-621), so both calls compute A[b] + A[b+B] with A[i] = attention(q = y[i],
        kv = y[(i+B) mod 2B]), y = norm1(new) -- the same sum, bit for bit (fp addition commutes).  So: one normalisation and one
        projection of the 2B feature sets, ONE attention launch over 2B batch elements with the key / value batch rotated by B
        (half the attention, projection and MLP work of running the block on the stacked (2B,2,N,C) pairs), and the three flow
        frames serve both directions.  Returns (2B,3,N,3)."""
        B2, N, C = new.shape
        B = B2 // 2
        a = prefix + ".attn_feats"
        P = self._params()
        def heads():  # head slot 0 is dropped and nothing after the attention mixes slots: project only slots 1..3; the eval-mode
            # BatchNorm in front (norm1: y = g x + h per channel) folds into both projections, W (g x + h) + b = (W diag g) x + (W h + b)
            wq, bq, wkv, bkv = self.W(a + ".q"), self.Bv(a + ".q"), self.W(a + ".kv"), self.Bv(a + ".kv")
            sl = lambda t: None if t is None else torch.cat([t[C:4 * C], t[5 * C:8 * C]], dim=0).contiguous()
            wq, bq, wkv, bkv = wq[C:].contiguous(), None if bq is None else bq[C:].contiguous(), sl(wkv), sl(bkv)
            g = P[prefix + ".norm1.weight"] * torch.rsqrt(P[prefix + ".norm1.running_var"] + 1e-5)
            hsh = P[prefix + ".norm1.bias"] - P[prefix + ".norm1.running_mean"] * g
            fold = lambda w, b: ((w * g[None, :]).contiguous(), (w @ hsh + (0 if b is None else b)).contiguous())
            return (*fold(wq, bq), *fold(wkv, bkv))
        wq, bq, wkv, bkv = self.derived(("cfa_heads_folded", prefix), heads)
        q, kv = F.linear(new, wq, bq), F.linear(new, wkv, bkv)                    # (2B,N,3C), (2B,N,6C) = [k | v]
        att = ops.backend().attention_rot(q, kv[..., :3 * C], kv[..., 3 * C:], 3, B, scale=C ** -0.5)     # (2B,N,3C)
        # the sum over the two frames; proj and the MLP act on the last axis only, so the (N, slot) order of the rows is kept as the
        # attention wrote it and only the small (B,N,3,3) result is rearranged (transposing o first cost two copies of it)
        o = (att[:B] + att[B:]).reshape(B, N, 3, C)
        o = self.lin(o, a + ".proj")
        frames = self.mlp_t(prefix + ".trans_block_2", o, tail=prefix + ".mapping_xyz").transpose(1, 2)   # (B,3,N,3)
        return torch.cat([frames, frames], dim=0)

    def mlp_t(self, prefix, x, tail=None, res=None, bn=None, drop=0.0):
        """Mlp_T.forward (mocopci.py:1558-1565): fc1, depthwise 1x1 conv, PReLU, fc2 -- as ONE fused kernel (ops.mlp2).  Everything
        affine around the first layer is folded into it once: the depthwise k=1 conv is a per-channel scale + bias,
        (W x + b) * s + t = (s W) x + (s b + t); bn = (name, eps) is an eval-mode BatchNorm applied to x first,
        W (g x + h) + b = (W diag g) x + (W h + b).  tail: a Linear applied to the result (mapping_xyz), folded into fc2 where
        only the 3-channel flow is read.  res: residual added to the result (inside the kernel).  drop > 0 (net.train()): the
        reference's dropout after the activation and after fc2 (mocopci.py:1561-1564, :1592-1595) -- the layers then run unfused."""
        P = self._params()
        be = ops.backend()

        def fold():
            w1, b1 = self.W(prefix + ".fc1"), self.Bv(prefix + ".fc1")
            if prefix + ".dwconv.dwconv.weight" in P:
                sc = P[prefix + ".dwconv.dwconv.weight"].reshape(-1)
                w1, b1 = w1 * sc[:, None], b1 * sc + P[prefix + ".dwconv.dwconv.bias"]
            if bn is not None:
                g = P[bn[0] + ".weight"] * torch.rsqrt(P[bn[0] + ".running_var"] + bn[1])
                hsh = P[bn[0] + ".bias"] - P[bn[0] + ".running_mean"] * g
                w1, b1 = w1 * g[None, :], b1 + w1 @ hsh
            w2, b2 = self.folded_tail(prefix + ".fc2", tail) if tail is not None else (self.W(prefix + ".fc2"), self.Bv(prefix + ".fc2"))
            return w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        w1, b1, w2, b2 = self.derived(("mlp_t", prefix, tail, bn), fold)
        live = self._live is not None
        if drop > 0.0 or not be.mlp2_supported(w1.shape[1], w1.shape[0], w2.shape[0]):  # dropout, or widths the kernel is not built for
            if drop <= 0.0 and res is None and be.linear_narrow_supported(x.numel() // x.shape[-1], w1.shape[0], w2.shape[0]):
                # flow tail at widths mlp2 is not built for (C = 256): library GEMM, then PReLU + the (4C -> 3) map as one kernel
                slope = P[prefix + ".act.weight"] if live else self.derived(("slope", prefix), lambda: float(P[prefix + ".act.weight"]))
                return be.linear_narrow(F.linear(x, w1, b1), w2, b2, slope)
            def tall(t, w, b):   # as lin(): the streaming kernel where it takes the shape, else the library with the streaming backward
                if be.linear_supported(t, w.shape[0], few_rows=not live):
                    return be.linear(t, w, b, 1.0, None)
                if live and t.is_cuda and torch.is_grad_enabled() and t.numel() // t.shape[-1] >= 16384:
                    return ops.plain_linear(t, w, b)
                return F.linear(t, w, b)
            if drop > 0.0 and self._mode is not None:   # act + drop as one pass each way (mocopci.py:1561-1562)
                hid = be.prelu_dropout(tall(x, w1, b1), P[prefix + ".act.weight"], drop)
            else:
                hid = F.prelu(tall(x, w1, b1), P[prefix + ".act.weight"])
            out = self.dropout(tall(hid, w2, b2), drop)
            return out if res is None else out + res
        # the PReLU slope: the live parameter in a training forward (it gets its gradient), a cached float otherwise
        slope = P[prefix + ".act.weight"] if live else self.derived(("slope", prefix), lambda: float(P[prefix + ".act.weight"]))
        packed = None if live else self.derived(("mlp_t_pack", be.name, prefix, tail, bn), lambda: be.mlp2_pack(w1, b1, w2, b2))
        return be.mlp2(x, w1, b1, w2, b2, slope, res=res, packed=packed)

    def pred_head(self, x, prefix):
        """pred (mocopci.py:790-791 in :1033,:1044): Linear(64 -> 32), ReLU, Linear(32 -> 3) -- inference: ONE launch of the two-layer
        kernel (ReLU = PReLU with slope 0; the 32-wide hidden activation of 196608 rows never exists); a training forward keeps
        the two Linears (their explicit backward)."""
        be = ops.backend()
        w1, b1, w2, b2 = self.W(prefix + ".0"), self.Bv(prefix + ".0"), self.W(prefix + ".2"), self.Bv(prefix + ".2")
        if self._live is None and be.mlp2_supported(w1.shape[1], w1.shape[0], w2.shape[0]):
            packed = self.derived(("pred_pack", be.name, prefix), lambda: be.mlp2_pack(w1, b1, w2, b2))
            return be.mlp2(x, w1, b1, w2, b2, 0.0, packed=packed)
        return self.lin(self.lin(x, prefix + ".0", slope=0.0), prefix + ".2")

    def multi_frame_att(self, prefix, x, heads=8, rows=None, feats=True):
        """Multi_Frame_Att.forward (mocopci.py:551-575) batched, on the INNER frames only.  The reference runs 5 frames
        and returns frames[:, 1:-1].  Every operator in between is per frame and per point (eval-mode BatchNorm, 1x1
        convs, PReLU) except the attention, which pairs frame f with frame 4-f of the flipped stack: the inner three pair
        among themselves (1<->3, 2<->2), so frames 0 and 4 never reach the output and x holds frames 1..3, (B,3,N,C).
        rows: optional int64 indices into the flattened (sample, frame) axis; the result then has shape (len(rows),1,N,.)."""
        B, Fr, N, C = x.shape
        xn = self.bn_eval(x, prefix + ".norm1", 1e-5)
        xr = torch.flip(xn, dims=[1])
        if rows is not None:  # only these (sample, frame) rows are wanted: queries, residual path and MLPs shrink with them
            x, xn, xr = (t.reshape(B * Fr, 1, N, C)[rows] for t in (x, xn, xr))
        return self._mfa_core(prefix, x, xn, xr, heads, feats)

    def _mfa_core(self, prefix, x, xn, xr, heads=8, feats=True):
        """Multi_Frame_Att from its three inputs on: x (the stack + time codes), xn = norm1(x), xr = xn of the attention partners."""
        B, Fr, N, C = x.shape
        a = prefix + ".attn_feats"
        o = ops.backend().attention(self.lin(xn, a + ".q").reshape(B * Fr, N, C), self.lin(xr, a + ".kv").reshape(B * Fr, N, 2 * C),
                                    heads)                                         # (B*3,N,C)
        xn = self.lin(o.reshape(B, Fr, N, C), a + ".proj", res=xn)                 # xn + proj(attention), mocopci.py:559
        x = self.mlp_t(prefix + ".mlp", xn, res=x, bn=(prefix + ".norm2", 1e-5))   # x + mlp(norm2(xn)), mocopci.py:561-563
        if not feats:  # flows only: trans_block.fc2 and mapping_xyz collapse into one (4C -> 3) map
            return None, self.mlp_t(prefix + ".trans_block", x, tail=prefix + ".mapping_xyz")
        xf = self.mlp_t(prefix + ".trans_block", x)                               # (B,3,N,latent)
        frames = self.lin(xf, prefix + ".mapping_xyz")                            # (B,3,N,3)
        return xf, frames

    def multi_frame_att_full(self, prefix, x, heads=8):
        """Multi_Frame_Att.forward (mocopci.py:551-575) on all five frames, for a net.train() training forward: BatchNorm with
        each sample's own statistics over its five frames (so the outer two cannot be skipped), attention / projection /
        MLP dropout and stochastic depth.  x (B,5,N,C) -> (feature frames, flow frames) of the inner three."""
        B, Fr, N, C = x.shape
        drop = self._mode[0]
        xn = self.bn_batch(x, prefix + ".norm1", 1e-5)
        xr = torch.flip(xn, dims=[1])
        a = prefix + ".attn_feats"
        o = self.attend(self.lin(xn, a + ".q").reshape(B * Fr, N, C), self.lin(xr, a + ".kv").reshape(B * Fr, N, 2 * C), heads)
        o = self.dropout(self.lin(o.reshape(B, Fr, N, C), a + ".proj"), drop)
        xn = xn + self.drop_path(o)                                               # mocopci.py:559
        xb = self.bn_batch(xn, prefix + ".norm2", 1e-5)
        x = x + self.drop_path(self.mlp_t(prefix + ".mlp", xb, drop=drop))        # mocopci.py:561-563
        xf = self.mlp_t(prefix + ".trans_block", x, drop=drop)
        frames = self.lin(xf, prefix + ".mapping_xyz")
        return xf[:, 1:-1], frames[:, 1:-1]

    def index_tensor(self, values, device):
        """int64 device tensor of a small static index list, built once (no host-to-device copy inside the forward)."""
        key = ("index", tuple(values), str(device))
        self.__dict__.setdefault("_time_cache", {})
        if key not in self._time_cache:
            self._time_cache[key] = torch.tensor(values, device=device)
        return self._time_cache[key]

    def batch_map(self, values, device):
        """int32 device tensor of a static batch map (see ops.cross_volume), built once."""
        key = ("bmap", values, str(device))
        self.__dict__.setdefault("_time_cache", {})
        if key not in self._time_cache:
            self._time_cache[key] = torch.tensor(values, dtype=torch.int32, device=device)
        return self._time_cache[key]

    def area_matrix(self, n_in, n_out, device):
        """F.interpolate(mode="area") = adaptive average pooling along the last axis, as an (n_in, n_out) matrix: output j
        averages inputs floor(j*n_in/n_out) .. ceil((j+1)*n_in/n_out)-1.  (3 -> 32: one or two inputs per output, so the
        product is the same sum of the same rounded terms.)"""
        key = ("area", n_in, n_out, str(device))
        self.__dict__.setdefault("_time_cache", {})
        if key not in self._time_cache:
            m = torch.zeros(n_in, n_out)
            for j in range(n_out):
                lo, hi = (j * n_in) // n_out, -((-(j + 1) * n_in) // n_out)
                m[lo:hi, j] = 1.0 / (hi - lo)
            self._time_cache[key] = m.to(device)
        return self._time_cache[key]

    def time_code(self, ts, dim, device):
        """Multiframe_Attention.time_embedding (mocopci.py:172-180): float64 python math, stored fp32."""
        key = (tuple(ts), dim, str(device))
        self.__dict__.setdefault("_time_cache", {})
        if key not in self._time_cache:
            enc = torch.zeros(len(ts), dim)
            for i, t in enumerate(ts):
                for j in range(0, dim, 2):
                    enc[i, j] = math.sin(t * math.pow(10000, -j / dim))
                    if j + 1 < dim:
                        enc[i, j + 1] = math.cos(t * math.pow(10000, -(j + 1) / dim))
            self._time_cache[key] = enc.to(device)
        return self._time_cache[key]

    def time_pair(self, B, dim, device):
        """Time codes of the forward (rows [:B]) and backward (rows [B:]) decoder directions, (2B,5,1,dim); built once per shape."""
        key = ("time_pair", B, dim, str(device))
        self.__dict__.setdefault("_time_cache", {})
        if key not in self._time_cache:
            self._time_cache[key] = torch.cat([self.time_code(self.T_F, dim, device).expand(B, -1, -1),
                                               self.time_code(self.T_B, dim, device).expand(B, -1, -1)], dim=0).unsqueeze(2).contiguous()
        return self._time_cache[key]

    def mfa_projections(self, prefix, f1_new, f2_new, f1_0, f1_1, f2_0, f2_1):
        """The four cross_t11 / cross_t22 projections of Multiframe_Attention's concatenated features (mocopci.py:186-190):
        (t11 of set 1, t22 of set 2, t11 of set 2, t22 of set 1).  They read features only."""
        c1, c2 = (f1_0, f1_1, f1_new), (f2_0, f2_1, f2_new)
        if not ops.backend().linear_supported(list(c1), self.W(prefix + ".bid.cross_t11").shape[0], few_rows=self._live is None):
            c1, c2 = torch.cat(c1, dim=-1), torch.cat(c2, dim=-1)  # library path: concatenate once, both projections read it
        b = prefix + ".bid"
        return self.lin(c1, b + ".cross_t11"), self.lin(c2, b + ".cross_t22"), self.lin(c2, b + ".cross_t11"), self.lin(c1, b + ".cross_t22")

    def multiframe_attention(self, prefix, pc1, pc2, f1_new, f2_new, f1_0, f1_1, f2_0, f2_1, up_frames, time_enc, rows=None, idx_c12=None,
                             projections=None):
        """Multiframe_Attention.forward (mocopci.py:182-212).  time_enc (B,5,1,C).  projections: schedule node whose result's last four
        members are mfa_projections(...) -- the decoder then ran that feature-only chain on a side lane beside the warp / search chain
        below, which reads only coordinates and flows (f1_new / f2_new are not read here in that case)."""
        b, fe = prefix + ".bid", prefix + ".fe"
        if projections is None:
            t11_1, t22_2, t11_2, t22_1 = self.mfa_projections(prefix, f1_new, f2_new, f1_0, f1_1, f2_0, f2_1)
        bid_mlp = [b + ".mlp.0"]
        fe_mlp = [fe + ".mlp.0"]
        fes = []
        # The 16 feature-cosine neighbours depend only on the encoder features (f1_0, f2_0), not on the warped
        # coordinates: one search serves all nine cross() calls of this level.  The batch holds both decoder
        # directions, so the (f2_0 -> f1_0) search is the same result with its halves swapped.
        half = f1_0.shape[0] // 2
        if idx_c12 is None:
            idx_c12 = ops.backend().knn_cosine(f1_0, f2_0, 16)
        if isinstance(idx_c12, (tuple, list)):  # the encoder's lane produced both arrangements
            idx_c12, idx_c21 = idx_c12
        else:
            idx_c21 = torch.cat([idx_c12[half:], idx_c12[:half]], dim=0)
        # The loop over the 3 upsampled flows (mocopci.py:191-197) has no carried dependency -- the bid/fe layers always
        # see the original c_feat1/c_feat2 -- so the three iterations run as one batch of 3 x (2B); feat1_new/feat2_new
        # after the loop are those of the last iteration.
        B2 = pc1.shape[0]
        R = up_frames.shape[0] // B2                                               # up_frames: (R*B2,N,3), frame-major
        dev = pc1.device
        sel = None
        rows_py = None if rows is None else list(rows)
        if rows is not None:
            # flow-embedding features are read only at the selected (sample, frame) rows and at the rows their attention
            # pairs with, (b, f) <-> (b, R-1-f); the other members of the 3 x (2B) batch are dropped
            need = sorted({f * B2 + b for b, f in (divmod(i, R) for i in rows)} | {(R - 1 - f) * B2 + b for b, f in (divmod(i, R) for i in rows)})
            if len(need) < R * B2:
                sel = self.index_tensor(need, dev)
            rows = self.index_tensor(rows, dev)
        expand = lambda t: t.unsqueeze(0).expand(R, *t.shape).reshape(R * t.shape[0], *t.shape[1:])
        rep = expand if sel is None else (lambda t: expand(t)[sel])
        pick = (lambda t: t) if sel is None else (lambda t: t[sel])
        pc1r, pc2r = rep(pc1), rep(pc2)
        pc2w = self.warp(pc1r, pc2r, pick(up_frames))
        if self._live is None:
            # inference: the features and the feature-space neighbour lists of the R iterations are the same tensors; the
            # kernel reads batch element b of them from element b mod 2B through a batch map instead of R copies
            members = need if sel is not None else list(range(R * B2))
            bmap = self.batch_map(tuple(i % B2 for i in members), dev)
            if projections is not None:
                t11_1, t22_2, t11_2, t22_1 = self._sched.get(projections)[-4:]
            n1a = self.cross(pc1r, pc2w, t11_1, t22_2, None, None, b + ".pos", bid_mlp, True, idx_c12, bmap=bmap, shared=7)
            n2a = self.cross(pc2w, pc1r, t11_2, t22_1, None, None, b + ".pos", bid_mlp, True, idx_c21, bmap=bmap, shared=7)
            fea = self.cross(pc1r, pc2w, self.lin(n1a, fe + ".conv1"), self.lin(n2a, fe + ".conv2"), None, None, fe + ".pos", fe_mlp,
                             False, idx_c12, bmap=bmap, shared=4)
        else:
            ic12, ic21 = rep(idx_c12), rep(idx_c21)
            n1a = self.cross(pc1r, pc2w, rep(t11_1), rep(t22_2), None, None, b + ".pos", bid_mlp, True, ic12)
            n2a = self.cross(pc2w, pc1r, rep(t11_2), rep(t22_1), None, None, b + ".pos", bid_mlp, True, ic21)
            fea = self.cross(pc1r, pc2w, self.lin(n1a, fe + ".conv1"), self.lin(n2a, fe + ".conv2"), None, None, fe + ".pos", fe_mlp,
                             False, ic12)
        if self._live is not None:
            if sel is not None:  # back to the 3 x (2B) layout; the dropped members are never read
                full = fea.new_zeros((R * B2, *fea.shape[1:]))
                full[sel] = fea
                fea = full
            fes = list(fea.reshape(R, B2, *fea.shape[1:]).unbind(0))
        n1, n2 = n1a[-B2:], n2a[-B2:]  # last iteration's (unused when rows are selected)
        # mocopci.py:203 stacks [feat1_new, fe_0..2, feat2_new] + time codes as 5 frames; Multi_Frame_Att keeps only the
        # inner three (see multi_frame_att), so the two outer frames are never built here
        if self._mode is not None:  # net.train(): the block's BatchNorms see all five frames of a sample (mocopci.py:200-208)
            _, frames = self.multi_frame_att_full(prefix + ".cross_block", torch.stack([n1, *fes, n2], dim=1) + time_enc)
            return frames, n1, n2
        if self._live is None:
            # ONE launch (ops.mfa_prepare) for what were eight: scatter of the computed members into the 3 x (2B) layout, time codes,
            # norm1, the flip that pairs frame f with frame R-1-f, and the selection of the rows that are read downstream
            cb = prefix + ".cross_block"
            want = list(rows_py) if rows_py is not None else list(range(B2 * R))           # output rows r = b * R + f
            store = {mem: i for i, mem in enumerate(need)} if sel is not None else None     # member f * B2 + b -> position in fea
            def maps():
                pos = lambda bb, f: (store[f * B2 + bb] if store is not None else f * B2 + bb)
                bf = [divmod(r, R) for r in want]
                ss = torch.tensor([pos(bb, f) for bb, f in bf], dtype=torch.int32, device=dev)
                sp = torch.tensor([pos(bb, R - 1 - f) for bb, f in bf], dtype=torch.int32, device=dev)
                te = time_enc[:, 1:-1, 0]                                                                     # (B2,R,C)
                ts = torch.stack([te[bb, f] for bb, f in bf]).contiguous()
                tp = torch.stack([te[bb, R - 1 - f] for bb, f in bf]).contiguous()
                return ss, sp, ts, tp
            ss, sp, ts, tp = self.derived(("mfa_maps", cb, B2, R, tuple(want), None if store is None else tuple(need)), maps)
            P = self._params()
            def fold():
                scale = P[cb + ".norm1.weight"] * torch.rsqrt(P[cb + ".norm1.running_var"] + 1e-5)
                return scale.contiguous(), (P[cb + ".norm1.bias"] - P[cb + ".norm1.running_mean"] * scale).contiguous()
            scale, shift = self.derived(("bn", cb + ".norm1", 1e-5), fold)
            x, xn, xr = ops.backend().mfa_prepare(fea.contiguous(), ss, sp, ts, tp, scale, shift)
            shape = (len(want), 1) if rows_py is not None else (B2, R)
            _, frames = self._mfa_core(cb, *(t.reshape(*shape, *t.shape[1:]) for t in (x, xn, xr)), feats=False)
            return (frames[:, 0] if rows_py is not None else frames), n1a[-B2:], n2a[-B2:]
        x = torch.stack(fes, dim=1) + time_enc[:, 1:-1]                            # (B,3,N,C): a training forward (eval graph)
        # (the block's third output, downsample(x_f), is never read by MultiFrameEstimatier.forward in inference)
        if rows is not None:  # only some (sample, frame) flows are read downstream
            _, frames = self.multi_frame_att(prefix + ".cross_block", x, rows=rows, feats=False)
            return frames[:, 0], n1, n2                                            # (len(rows),N,3)
        _, frames = self.multi_frame_att(prefix + ".cross_block", x, feats=False)  # (B,3,N,3)
        return frames, n1, n2

    def qkv_projection(self, prefix, feats, like_rows=None):
        """TransformerBlock's x = fc1(features); q, k, v = w_qs(x), w_ks(x), w_vs(x) (pointT_layer2.py:62-66; the three
        projections have no bias and nothing else reads x) as ONE (C -> 3C) affine map: W = [Wq; Wk; Wv] W1, b = [..] b1.
        Returns the packed (B,N,3C) tensor [q | k | v]."""
        def fold():
            w3 = torch.cat([self.W(prefix + w) for w in (".w_qs", ".w_ks", ".w_vs")], dim=0)
            return (w3 @ self.W(prefix + ".fc1")).contiguous(), (w3 @ self.Bv(prefix + ".fc1")).contiguous()
        w, b = self.derived(("qkv_fold", prefix), fold)
        be = ops.backend()
        like = {} if like_rows is None or self._live is not None else {"policy_rows": like_rows}
        if be.linear_supported(feats, w.shape[0], few_rows=self._live is None, **like):  # tall inputs: the fused per-point Linear (67 vs 86 us at 196608 rows)
            packed = None if self._live is not None else self.derived(("qkv_pack", be.name, prefix), lambda: be.linear_pack(w, b, [feats.shape[-1]]))
            return be.linear(feats, w, b, 1.0, None, packed=packed, **like)
        return F.linear(feats, w, b)

    def transformer_block(self, prefix, feats, xyz, k=16, qkv=None, idx=None, like_rows=None):
        """TransformerBlock.forward (pointT_layer2.py:58-77): vector attention over the 16 nearest
        neighbours (direct squared distance; the reference's full argsort is replaced by the KNN kernel).
        qkv: the packed (B,N,3C) projections when the caller already has them; idx: likewise the neighbour lists."""
        be = ops.backend()
        if idx is None:
            idx = be.knn(xyz, xyz, k, mode=ops.MCP_DIST_DIRECT)
        if qkv is None:
            qkv = self.qkv_projection(prefix, feats, like_rows=like_rows)
        C = feats.shape[-1]
        w = [t for n in (".fc_delta.0", ".fc_delta.2", ".fc_gamma.0", ".fc_gamma.2") for t in (self.W(prefix + n), self.Bv(prefix + n))]
        packed = None if self._live is not None else self.derived(("ptblock_pack", be.name, prefix), lambda: be.ptblock_pack(*w))
        res = be.ptblock_layer(xyz, qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], idx, w, packed=packed)
        return self.lin(res, prefix + ".fc2", res=feats)   # the residual inside the Linear's epilogue where the fused kernel takes the shape

    def folded_conv_bn(self, conv, bn, eps):
        """1x1 conv followed by eval-mode BatchNorm as one affine map (cached)."""
        P = self._params()

        def fold():
            scale = P[bn + ".weight"] * torch.rsqrt(P[bn + ".running_var"] + eps)
            return (self.W(conv) * scale[:, None]).contiguous(), ((self.Bv(conv) - P[bn + ".running_mean"]) * scale + P[bn + ".bias"]).contiguous()
        return self.derived(("fold", conv, bn), fold)

    def fusion(self, p1, p2, k=32, idx_self=None, calls=1):
        """MultiFrameEstimatier.knn_group + fusion (mocopci.py:798-819).  p1, p2 (B,N,3)."""
        be = ops.backend()
        m = "multi_frame_inference.conv."
        if idx_self is None:
            idx_self = be.knn(p1, p1, k)
        idx = (idx_self, be.knn(p1, p2, k))                                        # 2 x (B,N,k), both index p2
        if self._mode is not None:
            return self.fusion_batch_stats(p1, p2.contiguous(), idx, calls)
        wb = [t for ci, bi in ((0, 1), (3, 4), (6, 7)) for t in self.folded_conv_bn(m + str(ci), m + str(bi), 1e-3)]
        return be.fusion_mlp(p1, p2.contiguous(), idx, *wb)

    def fusion_batch_stats(self, p1, p2, idx, calls):
        """fusion (mocopci.py:810-819) in a net.train() forward: the three Conv2d + BatchNorm2d(eps 1e-3) + ReLU layers with BATCH
        statistics, so nothing can be folded and the layers run unfused on the gathered (B,N,64,.) tensor.  The batch holds
        `calls` consecutive reference calls (the three interpolated frames, mocopci.py:1046-1051), each normalised with its own
        statistics, running estimates updated call by call.  Large inputs (the B = 8, N = 8192 step keeps 37 GiB of layer
        outputs alive otherwise) run call by call under torch.utils.checkpoint: a call's layers are rebuilt in the backward, one
        call at a time; the running statistics are updated here, outside the recomputed function, from the statistics it returns."""
        m = "multi_frame_inference.conv."
        layers = ((0, 1), (3, 4), (6, 7))
        be = ops.backend()
        if hasattr(be, "fusion_bn"):
            # the multi-pass kernels of csrc/fusion_bn.hip: statistics passes + the layer, and a backward with the BatchNorm mean
            # terms; nothing of size rows x channels is kept (the unfused form below is what the oracle backend runs)
            P = self._params()
            conv = [t for ci, _ in layers for t in (self.W(m + str(ci)), self.Bv(m + str(ci)))]
            aff = [t for _, bi in layers for t in (P[m + f"{bi}.weight"], P[m + f"{bi}.bias"])]
            per = p1.shape[0] // calls
            rows = per * p1.shape[1] * 64
            outs, stats = [], []
            for c in range(calls):
                sl = slice(c * per, (c + 1) * per)
                idx_c = tuple(i[sl] for i in idx) if isinstance(idx, (tuple, list)) else idx[sl]
                out, bn, var = be.fusion_bn(p1[sl], p2[sl], idx_c, conv, aff, 1e-3)
                outs.append(out)
                stats.append((bn, var))
            with torch.no_grad():
                # the calls' momentum updates in sequence, in closed form (bn_batch): one product per layer and statistic
                mom = self.BN_MOMENTUM
                coef, keep = self.ema_coefficients(calls, mom, p1.device)
                bns, vrs = torch.stack([b for b, _ in stats]), torch.stack([v for _, v in stats])     # (calls, 1024), (calls, 256)
                at_bn = at_var = 0
                for (_, bi), ch in zip(layers, (64, 64, 128)):
                    P[m + f"{bi}.running_mean"].mul_(keep).add_(coef @ bns[:, at_bn:at_bn + ch])
                    P[m + f"{bi}.running_var"].mul_(keep).add_(coef @ vrs[:, at_var:at_var + ch], alpha=rows / (rows - 1))
                    P[m + f"{bi}.num_batches_tracked"].add_(calls)
                    at_bn += 4 * ch
                    at_var += ch
            return torch.cat(outs, dim=0)
        idx = grad.whole(idx)
        if p1.shape[0] * p1.shape[1] * idx.shape[-1] * 128 * 4 <= self.CHECKPOINT_BYTES or not torch.is_grad_enabled():
            nb = ops.backend().group_rows(p2, idx)                                    # (B,N,64,3)
            resi = nb - p1.unsqueeze(2)
            x = torch.cat([resi, torch.norm(resi, dim=-1, keepdim=True)], dim=-1)
            for ci, bi in layers:
                x = F.linear(x, self.W(m + str(ci)), self.Bv(m + str(ci)))
                x = torch.relu(self.bn_batch(x.reshape(calls, -1, x.shape[-1]), m + str(bi), 1e-3).reshape(x.shape))
            wgt = torch.softmax(x.max(dim=-1)[0], dim=-1)
            return torch.sum(wgt.unsqueeze(-1) * nb, dim=2)
        from torch.utils.checkpoint import checkpoint
        P = self._params()
        G = ops.backend().group_rows
        params = [t for ci, bi in layers for t in (self.W(m + str(ci)), self.Bv(m + str(ci)), P[m + f"{bi}.weight"], P[m + f"{bi}.bias"])]

        def one_call(p1c, p2c, idxc, *wb):  # pure: reads nothing but its arguments, so the backward can run it again
            nb = G(p2c, idxc)
            resi = nb - p1c.unsqueeze(2)
            x = torch.cat([resi, torch.norm(resi, dim=-1, keepdim=True)], dim=-1)
            stats = []
            for i in range(3):
                w, b, gamma, beta = wb[4 * i:4 * i + 4]
                x = F.linear(x, w, b)
                flat = x.reshape(-1, x.shape[-1])
                mean, var = flat.mean(dim=0), flat.var(dim=0, unbiased=False)
                stats += [mean.detach(), var.detach()]
                x = torch.relu((x - mean) * (gamma * torch.rsqrt(var + 1e-3)) + beta)
            wgt = torch.softmax(x.max(dim=-1)[0], dim=-1)
            return (torch.sum(wgt.unsqueeze(-1) * nb, dim=2), *stats)
        per = p1.shape[0] // calls
        outs = []
        for c in range(calls):
            sl = slice(c * per, (c + 1) * per)
            out, *stats = checkpoint(one_call, p1[sl], p2[sl], idx[sl], *params, use_reentrant=False)
            outs.append(out)
            n = per * p1.shape[1] * idx.shape[-1]
            with torch.no_grad():
                for i, (ci, bi) in enumerate(layers):
                    rm, rv, mom = P[m + f"{bi}.running_mean"], P[m + f"{bi}.running_var"], self.BN_MOMENTUM
                    rm.mul_(1.0 - mom).add_(stats[2 * i], alpha=mom)
                    rv.mul_(1.0 - mom).add_(stats[2 * i + 1], alpha=mom * n / (n - 1))
                    P[m + f"{bi}.num_batches_tracked"].add_(1)
        return torch.cat(outs, dim=0)

    # ---- decoder ------------------------------------------------------------------------
    def run_decoder(self, pcs, feats, B, train=False):
        """_decoder run to completion (no deferred tail)."""
        gen = self._decoder(pcs, feats, B, train=train)
        try:
            next(gen)
        except StopIteration as stop:
            return stop.value
        raise RuntimeError("decoder yielded without defer")

    def _decoder(self, pcs, feats, B, train=False, defer=False):
        """MultiFrameEstimatier.forward (mocopci.py:821-1059) as a generator: with defer=True it yields ONCE, right after the
        refinement stage's furthest point sampling has been launched on its side stream -- the caller may enqueue other work on
        this stream there (begin() / finish(): the next batch's encoder and decoder) -- and finishes when resumed; without defer it
        never yields.  The result is the generator's return value.  pcs/feats hold both frames stacked on
        the batch axis (frame 1 = [:B], frame 2 = [B:]).  Returns out_lst: 3 x (B,N,3); with train=True
        (flows_lst_f, flows_lst_b, out_lst) as the reference does (mocopci.py:1056-1059), every (direction, frame) flow of every
        level being computed then (the training loss reads them all)."""
        m = "multi_frame_inference."
        dev = pcs[0].device
        sw = lambda t: torch.roll(t, B, 0) if t.requires_grad else torch.cat([t[B:], t[:B]], dim=0)   # swap the two frames (training graph: a roll, see run_encoder)
        sched = self._sched
        # "other" frame, same order (levels 1..3 are the ones read): the encoder issued these copies off the main stream as soon as
        # their sources existed -- the clouds with the sampling pyramid, the level-1 / 2 features on their lanes; level 3's features are
        # needed right here (a lane result would make this stream wait for everything queued in front of it on the shared hardware
        # queue), so that one copy is made on this stream.  Fetched where a level is first read.
        class _Other:
            def __init__(self, fetch):
                self.fetch, self.have = fetch, {}

            def __getitem__(self, i):
                if i not in self.have:
                    self.have[i] = self.fetch(i)
                return self.have[i]
        pcs_o = _Other(lambda i: sched.get("swap_pc")[i])
        feats_o = _Other(lambda i: sched.get(("swap_f", i)) if sched.has(("swap_f", i)) else sw(feats[i]))
        cache = {}
        if not train:
            # level 0's stacked inputs (frames 1, 1, 2 for the three interpolated frames) read encoder outputs only: four copies made
            # beside the level-0 interpolation search, long before this stream wants them
            def rep0():
                i3_, w3_ = sched.get("i3_01")
                r3 = lambda t: (lambda a, b: torch.cat([a, a, b], dim=0))(*self.halves(t))
                return r3(pcs[0]), r3(feats[0]), r3(i3_), r3(w3_)
            sched.run("rep0", rep0)

        # EI cross-formers (mocopci.py:830-836; fusion features shared by both frames), the feature-cosine searches and the level-0
        # interpolation search were issued by the encoder as soon as their inputs existed; level 3's own are needed right here
        self._mark("dec start")
        up43 = lambda: self.conv1d_block(self.interp(pcs[3], pcs[4], feats[4], cache, "43"), m + "deconv4_3")   # l4 -> l3 (mocopci.py:842-845)
        # beside the EI cross-former of level 3 (a row of small launches on this stream): the upsampled level-4 features and cross3's
        # feature-cosine search -- both read encoder outputs only
        f3o = feats_o[3]
        sched.run("up43", up43)
        sched.run(("cos", 3), lambda: ops.backend().knn_cosine(feats[3], f3o, 16))
        f3 = self.ei_crossformer(m + "ei3", *self.halves(feats[3]), stacked=feats[3])
        fus = [None, None, None, None]

        self._mark("ei3 done")
        f_l4_3 = sched.get("up43")
        # (2B,256,576) = [encoder | fusion (the same for both frames) | upsampled level 4]: ONE concatenation, the shared fusion features
        # enter as a broadcast view instead of a stacked copy
        N3, C3 = feats[3].shape[1], feats[3].shape[2]
        c3 = torch.cat([feats[3].view(2, B, N3, C3), f3.unsqueeze(0).expand(2, B, N3, f3.shape[-1]), f_l4_3.view(2, B, N3, -1)],
                       dim=-1).view(2 * B, N3, -1)
        # cross3 (pointconv_util.py:783-791): rows [:B] give feat1_new, rows [B:] give feat2_new
        x = m + "cross3"
        if self._live is None:
            # points2 = cross_t22 of the OTHER frame's concatenation: the kernel reads batch element b of the projection from element
            # (b + B) mod 2B through its batch map instead of a swapped copy of the 576-wide input
            swap_map = self.batch_map(tuple((i + B) % (2 * B) for i in range(2 * B)), dev)
            # the two 576 -> 256 projections (library GEMMs, 60 us each at 4096 rows) side by side
            sched.run(("t22", 3), lambda: self.lin(c3, x + ".cross_t22"))
            t11_3 = self.lin(c3, x + ".cross_t11")
            new3 = self.cross(pcs[3], pcs_o[3], t11_3, sched.get(("t22", 3)), feats[3], feats_o[3], x + ".pos1", [x + ".mlp1.0"], False,
                              idx_c=sched.get(("cos", 3)), bmap=swap_map, shared=2)
        else:
            new3 = self.cross(pcs[3], pcs_o[3], self.lin(c3, x + ".cross_t11"), self.lin(sw(c3), x + ".cross_t22"), feats[3],
                              feats_o[3], x + ".pos1", [x + ".mlp1.0"], False, idx_c=sched.get(("cos", 3)))
        if self._live is None:  # the two per-frame Linears as ONE batched product written straight into the stacked layout
            wt, bt = self.derived(("cross3_t12", x), lambda: (torch.stack([self.W(x + ".cross_t1").t(), self.W(x + ".cross_t2").t()]).contiguous(),
                                                               torch.stack([self.Bv(x + ".cross_t1"), self.Bv(x + ".cross_t2")]).unsqueeze(1).contiguous()))
            new3 = torch.baddbmm(bt, new3.reshape(2, -1, new3.shape[-1]), wt).reshape(new3.shape[0], new3.shape[1], -1)
        else:
            new3 = (lambda a, b: torch.cat([self.lin(a, x + ".cross_t1"), self.lin(b, x + ".cross_t2")], dim=0))(*self.halves(new3))
        # Feature-only chains (upsampled features -> deconv -> the four projections of the next Multiframe_Attention) are schedule nodes
        # of their own -- their lane has been idle since the encoder -- beside this stream's coordinate chain (cross_block3 / flow
        # upsampling / warp / searches): both are rows of small launches that leave the chip mostly empty.  The 3-NN search both
        # sides interpolate with is made first, on this stream.
        proj = self._live is None   # inference: the projections come from the node; a training forward keeps them in its autograd graph here
        if proj:
            self.interp_prepare(pcs[2], pcs[3], cache, "32")
            fus[2] = sched.get(("fus", 2))
            f2o = feats_o[2]
            def chain2():
                f = self.conv1d_block(self.interp(pcs[2], pcs[3], new3, cache, "32"), m + "deconv3_2")
                return self.mfa_projections(m + "multi_frame_up_2", f, sw(f), feats[2], fus[2], f2o, fus[2])
            sched.run(("mfa_proj", 2), chain2)
            f_l3_2 = None
        else:
            f_l3_2 = self.conv1d_block(self.interp(pcs[2], pcs[3], new3, cache, "32"), m + "deconv3_2")
        # cross_block3, both directions at once (mocopci.py:853-856)
        self._mark("cross3 done")
        if self._live is None and self.PAIR_CFA:
            frame3s = self.cross_frame_att_pair(m + "cross_block3", new3)          # (2B,3,N3,3), both directions from one evaluation
        else:
            xs = torch.stack([new3, sw(new3)], dim=1)                              # (2B,2,N3,C)
            _, frame3s = self.cross_frame_att(m + "cross_block3", xs, feats=False)  # (2B,3,N3,3)

        # Which level-1 flows are read: l0 (below) uses, of the 2B samples x 3 frames, the forward branch's frames 0,1 and
        # the backward branch's frame 0.  (Their flow embeddings also need each frame's attention partner f <-> 2-f, so
        # only (backward, frame 1) is dead at level 1; the same selection one level up measured no gain and is not made.)
        rows1 = None if train else [3 * i for i in range(B)] + [3 * i + 1 for i in range(B)] + [3 * (i + B) for i in range(B)]

        self._mark("cross_block3 + deconv done")
        # l2 (mocopci.py:870-911): rows [:B] = forward direction, rows [B:] = backward direction
        ups = self.interp_flows(pcs[2], pcs[3], frame3s, cache, "32")
        C = feats[2].shape[-1]
        te = self.time_pair(B, C, dev)                                              # (2B,5,1,C)
        fus[2] = sched.get(("fus", 2))
        self._mark("got early fus2/cos2")
        frame2s, n1_2, n2_2 = self.multiframe_attention(m + "multi_frame_up_2", pcs[2], pcs_o[2], f_l3_2, None if proj else sw(f_l3_2), feats[2],
                                                        fus[2], feats_o[2], fus[2], ups, te, idx_c12=sched.get(("cos", 2)),
                                                        projections=("mfa_proj", 2) if proj else None)  # (2B,3,N2,3)
        # l2 -> l1 (mocopci.py:920-927): the forward branch upsamples (feat1_new_f -> pc1, feat2_new_f -> pc2),
        # the backward branch (feat1_new_b -> pc1, feat2_new_b -> pc2) where *_b come from the swapped call.
        def upsampled1():
            (n1f, n1b), (n2f, n2b) = self.halves(n1_2), self.halves(n2_2)
            f1_up = self.conv1d_block(self.interp(pcs[1], pcs[2], torch.cat([n1f, n2f], 0), cache, "21"), m + "deconv2_1")
            f2_up = self.conv1d_block(self.interp(pcs[1], pcs[2], torch.cat([n2b, n1b], 0), cache, "21"), m + "deconv2_1")
            # forward call gets (feat1_l2_1_f, feat2_l2_1_f); backward call gets (feat2_l2_1_b, feat1_l2_1_b)
            (a0, a1), (b0, b1) = (t.reshape(2, B, *t.shape[1:]).unbind(0) for t in (f1_up, f2_up))   # halves by unbind (backward: a stack)
            return torch.cat([a0, b1], dim=0), torch.cat([a1, b0], dim=0)
        fus[1] = sched.get(("fus", 1))
        if proj:
            self.interp_prepare(pcs[1], pcs[2], cache, "21")
            f1o = feats_o[1]
            sched.run(("mfa_proj", 1), lambda: self.mfa_projections(m + "multi_frame_up_1", *upsampled1(), feats[1], fus[1], f1o, fus[1]))
            f_up_1 = f_up_1_o = None
        else:
            f_up_1, f_up_1_o = upsampled1()
        ups = self.interp_flows(pcs[1], pcs[2], frame2s, cache, "21")
        C = feats[1].shape[-1]
        te = self.time_pair(B, C, dev)
        # l0 (mocopci.py:997-1053).  Output frames 0,1 use the forward branch (flow index i on frame 1);
        # frame 2 uses the backward branch: up_frame0_lst_b[2] = upsample(frame1s_b[:, 3-2-1]).  Of the level-1 flows
        # (2B samples x 3 frames) only these 3B are read: [:B] frame 0, [:B] frame 1, [B:] frame 0.
        self._mark("level2 done, got early fus1")
        frame1s = self.multiframe_attention(m + "multi_frame_up_1", pcs[1], pcs_o[1], f_up_1, f_up_1_o, feats[1], fus[1],
                                            feats_o[1], fus[1], ups, te, rows=rows1, idx_c12=sched.get(("cos", 1)),
                                            projections=("mfa_proj", 1) if proj else None)[0].contiguous()
        rep3 = lambda t: (lambda a, b: torch.cat([a, a, b], dim=0))(*self.halves(t))
        # the three refinements interpolate on (pc1->pc1, pc1->pc1, pc2->pc2): one 3-NN search on the stacked frames
        # (2B rows), its rows repeated for the 3B arrangement
        if train:
            pc0, f0 = rep3(pcs[0]), rep3(feats[0])
            i3, w3 = sched.get("i3_01")
        else:
            pc0, f0, i3r, w3r = sched.get("rep0")
        if train:
            # all six upsampled level-1 flows (mocopci.py:1011-1019): up_f[i] = upsample(frame1s_f[:, i]) on frame 1's points,
            # up_b[i] = upsample(frame1s_b[:, 2 - i]) on frame 2's points
            # (unbind, not six slices: its backward is one stack, a slice's is a zero-filled tensor plus a copy)
            lv1 = [d.unbind(1) for d in frame1s.reshape(2, B, 3, frame1s.shape[2], 3).unbind(0)]   # [direction][frame] -> (B,N1,3)
            src6 = torch.cat([lv1[0][0], lv1[0][1], lv1[0][2], lv1[1][2], lv1[1][1], lv1[1][0]], dim=0)
            rep6 = lambda t: (lambda a, b: torch.cat([a] * 3 + [b] * 3, dim=0))(*self.halves(t))
            up6 = ops.backend().interp3_apply(src6.contiguous(), rep6(i3), rep6(w3))  # (6B,N,3)
            up6 = list(up6.split(B))
            up_f, up_b = up6[:3], up6[3:]
            up_flow = torch.cat([up_f[0], up_f[1], up_b[2]], dim=0)
        else:
            flow_src = frame1s                                                     # (3B,N1,3): the three flows read below
            up_flow = ops.backend().interp3_apply(flow_src, i3r, w3r)              # (3B,N,3)
        self._mark("level1 done")
        warped = pc0 + up_flow
        # F.interpolate(size=32, mode="area") over the 3 flow components (mocopci.py:1021-1022), then rlevel0: read next by the
        # refinement stage's PointConvD, after the sampling -- a node beside the warped clouds' self search
        sched.run("wf", lambda: self.conv1d_block(f0 + up_flow @ self.area_matrix(3, f0.shape[-1], dev), m + "rlevel0"), reads=(f0,))
        # the refinement stage's sampling: a 1.2-1.4 ms latency chain on 24 CUs
        if train:   # the index chain alone goes to the lane (no gradient); the gather that carries the gradient to `warped` stays here
            sched.run("refine_fps", lambda: ops.backend().fps(warped.detach(), 2048))
        else:
            sched.run("refine_fps", lambda: self.fps_gather(warped, 2048, return_idx=True))
        idx_self = ops.backend().knn(warped, warped, 32)      # fusion's self search: independent of the refine branch
        be = ops.backend()
        like = warped.shape[0] * warped.shape[1]   # kernels as for every candidate centre (what the speculative form computes): same bits
        if defer and sched.lane("refine_fps") is not None:
            # The sampling is on its way; whatever the caller enqueues on this stream before resuming runs beside it, so neither the
            # wait for it nor the 4x speculative PointConvD below is paid.  The tensors the tail reads go to the caller (finish() may
            # run the tail on another stream and must record them there).
            self._mark("refine FPS launched (tail deferred)")
            yield (warped, idx_self)
            down, sel = sched.get("refine_fps")
            sched.run("i3_refine", lambda: be.interp3_search(warped, down))
            # the Point-Transformer's 16-NN search needs the sampled cloud only: beside PointConvD
            sched.run("knn_down", lambda: be.knn(down, down, 16, mode=ops.MCP_DIST_DIRECT), reads=(down,))
            dfeat = self.pointconv(m + "level1", warped, down, sched.get("wf"), idx=self.sampled_neighbours(idx_self, sel), like_rows=like)
            shape = self.transformer_block(m + "shape1", dfeat, down, idx=sched.get("knn_down"), like_rows=like)
            upf = be.interp3_apply(shape, *sched.get("i3_refine"))
        elif sched.lane("refine_fps") is not None and not train:
            # same speculation as in the encoder: PointConvD of EVERY candidate centre and the Point-Transformer's four
            # per-point projections are computed while the sampling runs, the sampled rows are gathered afterwards
            t = m + "shape1"
            dfeat_all = self.pointconv(m + "level1", warped, warped, sched.get("wf"), idx=idx_self)
            qkv_all = self.qkv_projection(t, dfeat_all)                           # (3B,N,192)
            self._mark("refine speculation done (before FPS wait)")
            down, sel = sched.get("refine_fps")
            # the 3-NN search of the upsampling below needs only (warped, down): it runs beside the Point-Transformer kernel
            sched.run("i3_refine", lambda: be.interp3_search(warped, down))
            dfeat = be.group_rows(dfeat_all, sel)
            shape = self.transformer_block(t, dfeat, down, qkv=be.group_rows(qkv_all, sel))
            upf = be.interp3_apply(shape, *sched.get("i3_refine"))
        else:
            # down = warped[sel]: its 32 nearest in warped are rows of the self search the fusion stage needs anyway
            if train:
                sel = sched.get("refine_fps")
                down = be.group_rows(warped, sel)
            else:
                down, sel = sched.get("refine_fps")
            dfeat = self.pointconv(m + "level1", warped, down, sched.get("wf"), idx=self.sampled_neighbours(idx_self, sel))
            shape = self.transformer_block(m + "shape1", dfeat, down)
            upf = ops.backend().interp3(warped, down, shape)
        self._mark("ptblock + interp done")
        refine = self.pred_head(upf, m + "pred")                                   # (3B,N,3): Linear, ReLU, Linear
        self._mark("refine coords done")
        final = self.fusion(warped, refine, idx_self=idx_self, calls=3)
        self._mark("fusion done")
        out_lst = list(final.split(B))
        if not train:
            return out_lst
        # the lists the training loss reads (mocopci.py:1011-1059), all (B,n,3): index i = interpolated frame
        p1, p2 = [p[:B] for p in pcs], [p[B:] for p in pcs]
        per = lambda t: [d.unbind(1) for d in t.reshape(2, B, 3, t.shape[2], 3).unbind(0)]   # (2B,3,N_l,3) -> [direction][frame] (B,N_l,3)
        lv = {1: lv1, 2: per(frame2s), 3: per(frame3s)}                            # direction 0 forward, 1 backward
        flows_f = [[p1[0] + up_f[i] for i in range(3)], [p1[0] + up_b[2 - i] for i in range(3)]]
        flows_b = [[p2[0] + up_b[i] for i in range(3)], [p2[0] + up_f[2 - i] for i in range(3)]]
        for l in (1, 2, 3):
            flows_f.append([p1[l] + lv[l][0][i] for i in range(3)])
            flows_b.append([p2[l] + lv[l][1][2 - i] for i in range(3)])
        return flows_f, flows_b, out_lst

    def prefetch(self, xyz1, xyz2, inputs_ready=None):
        """Issue NOW everything of an inference forward on (xyz1, xyz2) that depends on nothing but the inputs -- the channel-last
        layout, the encoder's furthest-point-sampling pyramid (a ~2.5 ms chain of latency-bound kernels on 16 CUs) and the level-0
        self search -- on side lanes (issue_inputs_only), and return a handle for forward(..., prefetched=handle).  A serving loop
        calls it for batch k+1 while batch k is still being computed (forward(then_prefetch=...) does so right after batch k's encoder
        is enqueued), so the sampling chains run under batch k's decoder instead of stalling batch k+1's encoder.  inputs_ready: an
        event after which the inputs are complete (a loader's copy-stream event); without it the work is ordered behind the current
        stream.  The inputs must stay unmodified until the consuming forward has run: an in-place refill of the same buffers (copy_, any
        in-place op) is detected through the tensors' version counters and the handle is then refused; writes through raw pointers
        are not seen.  Returns None on backends without streams."""
        dev = xyz1.device
        sched = Schedule(self, dev)
        if sched.lane("xyz") is None:
            return None
        main = torch.cuda.current_stream(dev)
        if inputs_ready is None:   # no event: behind the caller's stream as it stands now
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(main)
        scope = {}
        with torch.no_grad(), ops.backend().cloud_scope(scope):
            self.issue_inputs_only(sched, lambda: torch.cat([xyz1, xyz2], dim=0).transpose(1, 2).contiguous(), after=(inputs_ready,))
        return {"inputs": (xyz1.data_ptr(), xyz2.data_ptr(), tuple(xyz1.shape)), "versions": (xyz1._version, xyz2._version),
                "stream": main.stream_id, "sched": sched, "scope": scope}

    def begin(self, xyz1, xyz2, prefetched=None, then_prefetch=None, inputs_ready=None):
        """First part of an inference forward, for a loop that keeps two batches in flight (software pipelining of consecutive
        batches on ONE stream): everything up to and including the launch of the refinement stage's furthest point sampling
        (encoder, decoder levels 3..1, warped clouds, their self search).  Returns a handle; finish(handle) enqueues the rest
        (Point-Transformer refinement, fusion) and returns out_lst.  A serving loop calls begin(batch k+1) BEFORE finish(batch k):
        the sampling of batch k -- a serial 1.2-1.4 ms chain that leaves 90 % of the chip idle, with nothing of batch k left to
        run beside it -- then overlaps the encoder of batch k+1, and the PointConvD speculation that otherwise fills the wait (4x
        the work) is not needed.  Same results as forward(), bit for bit and at every batch size: the deferred tail computes only the
        sampled rows of PointConvD / the Point-Transformer projection, with the kernels the speculative form runs on every candidate
        row (lin(like_rows=)); every batch's work is enqueued exactly once.  Arguments as forward()."""
        B = xyz1.shape[0]
        self._check_cache()
        be = ops.backend()
        h = prefetched
        if h is None and inputs_ready is not None:
            h = self.prefetch(xyz1, xyz2, inputs_ready)
        scope = {} if h is None else h["scope"]
        with torch.no_grad(), be.cloud_scope(scope):
            xyz, sched = self._consume_prefetched(h, xyz1, xyz2)
            self._sched = sched
            pcs, feats = self.run_encoder(xyz, sched)
            if then_prefetch is not None:
                self._next = self.prefetch(*then_prefetch)
            gen = self._decoder(pcs, feats, B, defer=True)
            try:
                reads = next(gen)
                out = None
            except StopIteration as stop:  # backends without streams never defer
                gen, out, reads = None, stop.value, ()
        ready = None
        if gen is not None:
            ready = torch.cuda.Event()
            ready.record()
        return {"gen": gen, "scope": scope, "out": out, "ready": ready, "sched": sched, "reads": reads}

    def finish(self, pending, tail_stream=None):
        """Second part of the forward begun with begin(): returns out_lst, 3 x (B,N,3).
        tail_stream (optional): enqueue this part on that stream instead of the current one, behind the first part's last kernel; it
        then runs CONCURRENTLY with whatever the current stream was given after begin() (the next batch's first part) -- the tail's
        chip-filling kernels (32-NN search, fusion) beside the next encoder's launch-bound ones.  The current stream is made to wait
        for the tail before it continues, so the returned tensors are safe to use on it."""
        if pending["gen"] is None:
            return pending["out"]
        cur = torch.cuda.current_stream()
        run_on = cur if tail_stream is None else tail_stream
        if run_on != cur:
            run_on.wait_event(pending["ready"])
            for t in pending["reads"]:   # produced on the first part's stream, read by the tail on another one
                t.record_stream(run_on)
        self._sched = pending["sched"]   # (another batch's begin() has run in between)
        with torch.no_grad(), ops.backend().cloud_scope(pending["scope"]), torch.cuda.stream(run_on):
            try:
                next(pending["gen"])
            except StopIteration as stop:
                pending["gen"] = None
                out = stop.value
            else:
                raise RuntimeError("decoder yielded twice")
            if run_on != cur:
                tail_done = torch.cuda.Event()
                tail_done.record(run_on)
        if run_on != cur:
            cur.wait_event(tail_done)
            for t in out:
                t.record_stream(cur)
        return out

    def _consume_prefetched(self, h, xyz1, xyz2):
        """(xyz, schedule) of a forward: from the prefetch handle (the stream waits for its layout), or laid out here."""
        if h is None:
            return torch.cat([xyz1, xyz2], dim=0).transpose(1, 2).contiguous(), Schedule(self, xyz1.device)
        main = torch.cuda.current_stream(xyz1.device)
        if h["inputs"] != (xyz1.data_ptr(), xyz2.data_ptr(), tuple(xyz1.shape)) or h["stream"] != main.stream_id:
            raise RuntimeError("prefetched handle belongs to other inputs or another stream")
        if h["versions"] != (xyz1._version, xyz2._version):
            # a loader refilled the buffers in place (copy_ / in-place ops) after prefetch() laid them out: the handle holds the OLD
            # contents.  (Writes through raw pointers are invisible to the version counters and remain the caller's responsibility.)
            raise RuntimeError("the inputs were modified in place after prefetch(): prefetch them again")
        return h["sched"].get("xyz"), h["sched"]

    def take_prefetched(self):
        """The handle forward(then_prefetch=...) produced (None if it did not); hands it over once."""
        h, self._next = self.__dict__.get("_next"), None
        return h

    def forward(self, xyz1, xyz2, gt=None, t=None, train=False, inputs_ready=None, prefetched=None, then_prefetch=None):
        """MoCoPCI.forward (mocopci.py:1069-1097): xyz1, xyz2 (B,3,N) -> out_lst, 3 x (B,N,3).
        train=True: (frames_lst_f, frames_lst_b, gt_frame, out_lst) as the reference returns, computed with autograd enabled so
        that train.py:135-160's loss can be back-propagated: gradients reach every parameter through the fused kernels
        (mocopci_amd.grad).  gt: 3 x (B,3,N) as train.py:125-126 passes it.  As in the reference, what the normalisation and
        dropout layers do then follows the MODULE's mode: after net.train() (train.py:130) the BatchNorms of the fusion MLP,
        Multi_Frame_Att and Cross_Frame_Att normalise with batch statistics -- per sample where the reference loops over
        samples -- and update their running estimates, and dropout / stochastic depth are drawn at drop_rate / attn_drop_rate /
        drop_path_rate (torch's device RNG; the fused MLP / attention kernels give way to their unfused forms where a mask sits
        between their stages); after net.eval() (the constructor's state) the same call differentiates the inference graph
        (running statistics, no dropout).  A forward with train=False is always the inference graph.
        Pipelining of consecutive batches (inference, optional; a single isolated call is unchanged without them):
        inputs_ready: a torch.cuda.Event after which xyz1 / xyz2 are complete: the input-only work (see prefetch) is issued behind
        THAT event instead of behind the caller's whole stream, so in a loop of forwards it overlaps the previous call's tail.
        prefetched: the handle of an earlier prefetch(xyz1, xyz2) -- that work is then not issued again.
        then_prefetch: (next_xyz1, next_xyz2[, ready event]) -- once this call's encoder is enqueued, prefetch() the NEXT batch, so
        its sampling chains run under this call's decoder; the handle is collected with take_prefetched()."""
        B = xyz1.shape[0]
        self._check_cache()
        if not train:
            h = prefetched
            if h is None and inputs_ready is not None:
                h = self.prefetch(xyz1, xyz2, inputs_ready)
            with torch.no_grad(), ops.backend().cloud_scope(None if h is None else h["scope"]):
                xyz, self._sched = self._consume_prefetched(h, xyz1, xyz2)
                pcs, feats = self.run_encoder(xyz, self._sched)
                if then_prefetch is not None:
                    self._next = self.prefetch(*then_prefetch)
                return self.run_decoder(pcs, feats, B)
        xyz = torch.cat([xyz1, xyz2], dim=0).transpose(1, 2).contiguous()
        self._live = {**dict(self.named_parameters()), **dict(self.named_buffers())}
        self._mode = (float(self.drop_rate), float(self.attn_drop_rate), float(self.drop_path_rate)) if self.training else None
        self._train_lane0 = self.TRAIN_PYRAMID_LANE and not (xyz1.requires_grad or xyz2.requires_grad)
        N = xyz1.shape[2]
        try:
            with torch.enable_grad(), ops.backend().cloud_scope():
                self._sched = Schedule(self, xyz.device)
                if gt is not None:
                    # downsampling(), mocopci.py:1099-1104: every ground-truth frame sampled to N/4, N/16 and N/32 points, each from the
                    # full cloud.  Furthest point sampling is sequential from index 0, so the N/16 and N/32 samples are the first points
                    # of the N/4 sample (bit for bit), and the frames are independent clouds: one launch instead of nine.  It reads the
                    # ground truth only and carries no gradient: a schedule node on a lane of its own (a 1.2 ms latency chain on 24 CUs
                    # that used to run after the decoder, on the critical path), fetched where the lists are assembled below
                    def gt_down():
                        with torch.no_grad():
                            return self.fps_gather(torch.cat([g.transpose(1, 2) for g in gt], dim=0).contiguous(), N // 4)
                    self._sched.run("gt_down", gt_down)
                pcs, feats = self.run_encoder(xyz, self._sched)
                flows_f, flows_b, out_lst = self.run_decoder(pcs, feats, B, train=True)
                pts = self._sched.get("gt_down") if gt is not None else None
        finally:
            self._live = None
            self._mode = None
        gt_frame = []
        if gt is not None:
            with torch.no_grad():
                Bg = gt[0].shape[0]
                for i, g in enumerate(gt):
                    p = pts[i * Bg:(i + 1) * Bg]
                    gt_frame.append([g] + [p[:, :N // d].transpose(1, 2).contiguous() for d in (4, 16, 32)])
        frames_lst_f = [[lst[i] for lst in flows_f] for i in range(3)]
        frames_lst_b = [[lst[i] for lst in flows_b] for i in range(3)]
        return frames_lst_f, frames_lst_b, gt_frame, out_lst
