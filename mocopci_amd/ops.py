"""Layer-level point-set operators of the hot path on channel-last tensors, backed by
libmocopci_hip.so.  These are the fused counterparts of the reference's Python helpers
(models/m_models/mocopci.py:1130-1266,1456-1502; models/pointconv_util.py:67-192) that the
model harness (mocopci_amd/model.py) is written against.

Layout convention: xyz (B,N,3), features (B,N,C) -- one neighbour = one contiguous row --
instead of the reference's (B,C,N); indices are int32 end to end.

`HipBackend` is the product path.  The harness resolves operators through `backend()` so
that tests and bench.py's cpu_baseline leg can run the SAME graph on the CPU oracle by
calling `set_backend(...)`; nothing in this package imports the oracle, and HipBackend has
no CPU fallback (a CPU tensor raises).
"""
import contextlib
import ctypes
import threading

import torch

from . import _lib, grad

MCP_DIST_EXPANSION = 0
MCP_DIST_DIRECT = 1


_raw_stream = torch._C._cuda_getCurrentRawStream  # (device index) -> hipStream_t of torch's current stream, without the Python
_cur_device = torch._C._cuda_getDevice            # wrappers of torch.cuda.current_stream(): ~10 us per call, 90 calls per step


def _call(name, ref_tensor, *args):
    """One library call on torch's current stream of the tensor's device.  The device guard is taken only when the tensor lives
    on another device than the current one (the guard costs more host time than the launch itself)."""
    fn = getattr(_lib.load(), name)
    idx = ref_tensor.device.index
    if idx is None or idx == _cur_device():
        rc = fn(*args, _raw_stream(_cur_device() if idx is None else idx))
    else:
        with torch.cuda.device(idx):
            rc = fn(*args, _raw_stream(idx))
    if rc:
        _lib.check(rc)


def _group_rows_fwd(points, idx):
    B, N, C = points.shape
    idx = idx.contiguous()
    T = idx[0].numel()
    out = torch.empty((*idx.shape, C), dtype=torch.float32, device=points.device)
    _call("mcp_group_rows", points, B, N, C, T, _lib.fptr(points), _lib.iptr(idx), _lib.fptr(out))
    return out


def _interp3_apply_fwd(feat, idx3, w3):
    B, N, _ = idx3.shape
    S, C = feat.shape[1], feat.shape[2]
    out = torch.empty((B, N, C), dtype=torch.float32, device=feat.device)
    _call("mcp_interp3_apply", feat, B, N, S, C, _lib.fptr(feat), _lib.iptr(idx3), _lib.fptr(w3), _lib.fptr(out))
    return out


class _Interp3ApplyFn(torch.autograd.Function):
    """mcp_interp3_apply with its deterministic backward (mcp_interp3_apply_grad_sorted): the weights' gradient as a gather-dot, the
    features' as a weighted segmented reduction over the CSR form of idx3 -- no (B,N,3,C) tensor (autograd over the unfused blend
    builds three)."""

    @staticmethod
    def forward(ctx, feat, idx3, w3):
        feat, idx3, w3 = feat.detach().contiguous(), idx3.contiguous(), w3.detach().contiguous()
        ctx.save_for_backward(feat, idx3, w3)
        return _interp3_apply_fwd(feat, idx3, w3)

    @staticmethod
    def backward(ctx, grad_out):
        feat, idx3, w3 = ctx.saved_tensors
        B, N, _ = idx3.shape
        S, C = feat.shape[1], feat.shape[2]
        grad_out = grad_out.contiguous()
        need_f, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        d_feat = torch.empty_like(feat) if need_f else None
        d_w3 = torch.empty_like(w3) if need_w else None
        order, seg = _scatter_segments(idx3, S) if need_f else (None, None)
        _call("mcp_interp3_apply_grad_sorted", feat, B, N, S, C, _lib.fptr(feat), _lib.iptr(idx3), _lib.fptr(w3), _lib.fptr(grad_out),
              None if order is None else _lib.iptr(order), None if seg is None else _lib.iptr(seg),
              None if d_feat is None else _lib.fptr(d_feat), None if d_w3 is None else _lib.fptr(d_w3))
        return d_feat, None, d_w3


class _GroupRowsFn(torch.autograd.Function):
    """Row gather with its scatter-add backward (channel-last counterpart of GroupingOperation, pointnet2_utils.py:156-198)."""

    @staticmethod
    def forward(ctx, points, idx):
        idx = idx.contiguous()
        ctx.save_for_backward(idx)
        ctx.n = points.shape[1]
        return _group_rows_fwd(points.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _group_rows_grad(grad_out, idx, ctx.n), None


_segments_memo = None   # {key: (idx, (order, seg))} while a segments_memo() block is open (backward runs on autograd's own thread,
                        # so this is process-wide, not thread-local: one backward pass at a time, as in train.py)


@contextlib.contextmanager
def segments_memo():
    """Around one backward pass: the (order, seg) pair of a gather list is built once and serves every gradient scattered with
    that list -- the same neighbour list gathers coordinates, features and interpolation operands in several nodes of the graph.
    The memo holds the list itself, so its storage cannot be handed to another tensor while the block is open."""
    global _segments_memo
    outer, _segments_memo = _segments_memo, ({} if _segments_memo is None else _segments_memo)
    try:
        yield
    finally:
        _segments_memo = outer


def _scatter_segments(idx, n):
    """(order, seg) of a gather list (B,...) into rows 0..n-1: the gather positions sorted by destination row (stable) and the row
    boundaries in that order -- the operands of mcp_group_rows_grad_sorted; one pair serves every tensor gathered with the list.
    mcp_scatter_segments: a counting sort on the small keys (csrc/scatter_csr.hip) instead of a general stable key-value sort."""
    memo = _segments_memo
    if memo is None:
        return _build_segments(idx, n)
    key = (idx.data_ptr(), tuple(idx.shape), tuple(idx.stride()), idx.dtype, idx._version, n, idx.device.index)
    hit = memo.get(key)
    if hit is None:
        hit = memo[key] = (idx, _build_segments(idx, n), torch.cuda.current_stream(idx.device))
    elif hit[2] != torch.cuda.current_stream(idx.device):   # built on another stream of this backward pass
        done = torch.cuda.Event()
        done.record(hit[2])
        torch.cuda.current_stream(idx.device).wait_event(done)
        for t in hit[1]:
            t.record_stream(torch.cuda.current_stream(idx.device))
    return hit[1]


def _build_segments(idx, n):
    B = idx.shape[0]
    T = idx[0].numel()
    flat = idx.reshape(B, T)
    if flat.dtype != torch.int32:
        flat = flat.int()
    flat = flat.contiguous()
    order = torch.empty((B, T), dtype=torch.int32, device=idx.device)
    seg = torch.empty((B, n + 1), dtype=torch.int32, device=idx.device)
    need = _lib.load().mcp_scatter_segments_workspace_bytes(B, T, n)
    if need == 0:   # more destinations than the kernel's LDS histograms hold: a general stable sort
        keys, order64 = torch.sort(flat, dim=1, stable=True)
        bounds = torch.arange(n + 1, device=idx.device, dtype=keys.dtype).expand(B, n + 1).contiguous()
        return order64.int().contiguous(), torch.searchsorted(keys.contiguous(), bounds).int().contiguous()
    ws = torch.empty((need,), dtype=torch.uint8, device=idx.device)
    _call("mcp_scatter_segments", flat, B, T, n, _lib.iptr(flat), _lib.iptr(order), _lib.iptr(seg), ws.data_ptr(), need)
    return order, seg


def _group_rows_grad(grad_out, idx, n, segments=None):
    """Scatter-add of gathered rows' gradients (B,...,C) back to (B,n,C) as a deterministic segmented reduction: stable sort of the
    gather positions by destination row, in-order sums per row."""
    B, C = grad_out.shape[0], grad_out.shape[-1]
    T = idx[0].numel()
    grad_out = grad_out.contiguous()
    order, seg = segments if segments is not None else _scatter_segments(idx, n)
    grad_rows = torch.empty((B, n, C), dtype=torch.float32, device=grad_out.device)
    _call("mcp_group_rows_grad_sorted", grad_out, B, n, C, T, _lib.fptr(grad_out), _lib.iptr(order), _lib.iptr(seg), _lib.fptr(grad_rows))
    return grad_rows


class _PointConvAggFn(torch.autograd.Function):
    """mcp_pointconv_agg with its hand-written backward (mcp_pointconv_agg_grad): WeightNet recomputed per (centre, neighbour) pair in
    the backward kernel, per-neighbour gradients through the deterministic segmented scatter, weight gradients fixed-order sums."""

    @staticmethod
    def forward(ctx, be, idx, s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2):
        args = [t.detach().contiguous() for t in (s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2)]
        ctx.save_for_backward(idx, *args)
        return be._pointconv_agg(*args[:3], idx, *args[3:])

    @staticmethod
    def backward(ctx, grad_out):
        idx, s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2 = ctx.saved_tensors
        B, N, D = s_points.shape
        S = new_xyz.shape[1]
        lib, dev = _lib.load(), s_points.device
        grad_out = grad_out.contiguous()
        d_new = torch.empty_like(new_xyz)
        d_gxyz = torch.empty((B, S, 32, 3), dtype=torch.float32, device=dev)
        d_rows = torch.empty((B, S, 32, D), dtype=torch.float32, device=dev)
        d_w = torch.empty((lib.mcp_pointconv_agg_grad_floats(),), dtype=torch.float32, device=dev)
        need = lib.mcp_pointconv_agg_grad_workspace_bytes(B, S)
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _call("mcp_pointconv_agg_grad", s_points, B, N, S, D, 32, _lib.fptr(s_xyz), _lib.fptr(new_xyz), _lib.fptr(s_points), _lib.iptr(idx),
              _lib.fptr(w0), _lib.fptr(b0), _lib.fptr(w1), _lib.fptr(b1), _lib.fptr(w2), _lib.fptr(b2), _lib.fptr(grad_out), _lib.fptr(d_new),
              _lib.fptr(d_gxyz), _lib.fptr(d_rows), _lib.fptr(d_w), ws.data_ptr(), need)
        d_sxyz = d_spoints = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[4]:
            segments = _scatter_segments(idx, N)
            if ctx.needs_input_grad[2]:
                d_sxyz = _group_rows_grad(d_gxyz, idx, N, segments)
            if ctx.needs_input_grad[4]:
                d_spoints = _group_rows_grad(d_rows, idx, N, segments)
        pieces, at = [], 0
        for t in (w0, b0, w1, b1, w2, b2):
            pieces.append(d_w[at:at + t.numel()].view(t.shape))
            at += t.numel()
        return (None, None, d_sxyz, d_new, d_spoints, *pieces)


class _CrossFn(torch.autograd.Function):
    """mcp_cross_volume with its hand-written backward (mcp_cross_grad, D = 64 / 128): recompute inside the backward kernel, the
    per-neighbour gradients through the deterministic segmented scatter (one sort serves both), weight gradients fixed-order sums."""

    @staticmethod
    def forward(ctx, be, ia, ib, xyz1, xyz2, points1, points2, wpos, bpos, wmlp, bmlp):
        args = [t.detach().contiguous() for t in (xyz1, xyz2, points1, points2, wpos, bpos, wmlp, bmlp)]
        ctx.save_for_backward(ia, ib, *args)
        return be.cross_volume(*args[:4], ia if ib is None else (ia, ib), be.cross_pack(*args[4:]))

    @staticmethod
    def backward(ctx, grad_out):
        ia, ib, xyz1, xyz2, points1, points2, wpos, bpos, wmlp, bmlp = ctx.saved_tensors
        B, N1, D = points1.shape
        N2 = xyz2.shape[1]
        lib, dev = _lib.load(), points1.device
        grad_out = grad_out.contiguous()
        d_xyz1, d_points1 = torch.empty_like(xyz1), torch.empty_like(points1)
        d_dir = torch.empty((B, N1, 32, 3), dtype=torch.float32, device=dev)
        d_rows = torch.empty((B, N1, 32, D), dtype=torch.float32, device=dev)
        d_w = torch.empty((lib.mcp_cross_grad_floats(D),), dtype=torch.float32, device=dev)
        need = lib.mcp_cross_grad_workspace_bytes(B, N1, D)
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _call("mcp_cross_grad", points1, B, N1, N2, D, 32, _lib.fptr(xyz1), _lib.fptr(xyz2), _lib.fptr(points1), _lib.fptr(points2), _lib.iptr(ia),
              None if ib is None else _lib.iptr(ib), _lib.fptr(wpos), _lib.fptr(bpos), _lib.fptr(wmlp), _lib.fptr(bmlp), _lib.fptr(grad_out),
              _lib.fptr(d_xyz1), _lib.fptr(d_dir), _lib.fptr(d_points1), _lib.fptr(d_rows), _lib.fptr(d_w), ws.data_ptr(), need)
        d_xyz2 = d_points2 = None
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[6]:
            whole = ia if ib is None else torch.cat((ia, ib), dim=-1)
            segments = _scatter_segments(whole, N2)
            if ctx.needs_input_grad[4]:
                d_xyz2 = _group_rows_grad(d_dir, whole, N2, segments)
            if ctx.needs_input_grad[6]:
                d_points2 = _group_rows_grad(d_rows, whole, N2, segments)
        pieces, at = [], 0
        for t in (wpos, bpos, wmlp, bmlp):
            pieces.append(d_w[at:at + t.numel()].view(t.shape))
            at += t.numel()
        return (None, None, None, d_xyz1, d_xyz2, d_points1, d_points2, *pieces)


class _AttentionSmallFn(torch.autograd.Function):
    """mcp_attention_small (head dims 8 / 16) with its hand-written backward (mcp_attention_small_grad: row statistics, dQ, dK/dV)."""

    @staticmethod
    def forward(ctx, be, q, kv, heads, scale, drop_p, seed):
        q, kv = q.detach().contiguous(), kv.detach().contiguous()
        BF, Nq, C = q.shape
        Nk = kv.shape[1]
        out = torch.empty((BF, Nq, C), dtype=torch.float32, device=q.device)
        lse = torch.empty((BF, heads, Nq), dtype=torch.float32, device=q.device)   # kept for the backward: no statistics pass there
        _call("mcp_attention_small_lse", q, BF, Nq, Nk, heads, C // heads, _lib.fptr(q), C, _lib.fptr(kv), 2 * C, kv.data_ptr() + 4 * C, 2 * C,
              float(scale), float(drop_p), int(seed), out.data_ptr(), lse.data_ptr())
        ctx.heads, ctx.scale, ctx.drop_p, ctx.seed = heads, scale, drop_p, seed
        ctx.save_for_backward(q, kv, out, lse)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, kv, out, lse = ctx.saved_tensors
        BF, Nq, C = q.shape
        Nk, heads = kv.shape[1], ctx.heads
        lib = _lib.load()
        grad_out = grad_out.contiguous()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        need = lib.mcp_attention_small_grad_workspace_bytes(BF, Nq, heads)
        ws = torch.empty((need,), dtype=torch.uint8, device=q.device)
        _call("mcp_attention_small_grad_lse", q, BF, Nq, Nk, heads, C // heads, q.data_ptr(), C, kv.data_ptr(), 2 * C, kv.data_ptr() + 4 * C, 2 * C,
              float(ctx.scale), float(ctx.drop_p), int(ctx.seed), _lib.fptr(out), _lib.fptr(grad_out), _lib.fptr(lse), _lib.fptr(dq), _lib.fptr(dkv),
              ws.data_ptr(), need)
        return None, dq, dkv, None, None, None, None


def _prelu_drop_fwd(z, slope, drop_p=0.0, seed=0):
    """m * prelu(z, slope): slope a 1-element device tensor; drop_p = 0: plain PReLU (mcp_prelu_dropout)."""
    out = torch.empty_like(z)
    _call("mcp_prelu_dropout", z, z.numel(), _lib.fptr(z), _lib.fptr(slope), float(drop_p), int(seed), _lib.fptr(out))
    return out


def _prelu_drop_bwd(z, slope, grad_out, drop_p=0.0, seed=0):
    """(dz, dslope (1,)) of the above in one pass over z and grad_out (mcp_prelu_dropout_grad; dslope summed in a fixed order)."""
    dz, da = torch.empty_like(z), torch.empty((1,), dtype=torch.float32, device=z.device)
    need = _lib.load().mcp_prelu_dropout_grad_workspace_bytes(z.numel())
    ws = torch.empty((need,), dtype=torch.uint8, device=z.device)
    _call("mcp_prelu_dropout_grad", z, z.numel(), _lib.fptr(z), _lib.fptr(slope), _lib.fptr(grad_out), float(drop_p), int(seed),
          _lib.fptr(dz), _lib.fptr(da), ws.data_ptr(), need)
    return dz, da


class _PreluDropFn(torch.autograd.Function):
    """mcp_prelu_dropout / mcp_prelu_dropout_grad: PReLU (one slope, the live parameter) then dropout, one pass each way; only z is
    kept for the backward, which regenerates the mask from the seed."""

    @staticmethod
    def forward(ctx, z, slope, drop_p, seed):
        z = z.detach().contiguous()
        ctx.drop_p, ctx.seed, ctx.slope_shape = drop_p, seed, slope.shape
        slope = slope.detach().reshape(1).contiguous()
        ctx.save_for_backward(z, slope)
        return _prelu_drop_fwd(z, slope, drop_p, seed)

    @staticmethod
    def backward(ctx, grad_out):
        z, slope = ctx.saved_tensors
        dz, da = _prelu_drop_bwd(z, slope, grad_out.contiguous(), ctx.drop_p, ctx.seed)
        return dz, da.reshape(ctx.slope_shape), None, None


def _leaky_grad(gy, y, slope):
    """gy * act'(z) of a one-slope activation (0 <= slope <= 1) from its OUTPUT y (y > 0 exactly where z > 0): one pass."""
    return torch.ops.aten.leaky_relu_backward(gy, y, float(slope), True)


def _tall_matmul(a, w, bias=None):
    """a (rows, n) @ w (n, k) (+ bias) for rows >> n, k (the input gradient of a per-point Linear): the streaming forward kernel on w^T
    where it is built for the shape (the library GEMM takes ~160 us for 196608 x 64 @ 64 x 32 against ~30), the library otherwise."""
    be = backend()
    n, k = w.shape
    if a.is_cuda and hasattr(be, "linear_kernel_ok") and a.shape[0] >= 8192:
        a = a.contiguous()
        if be.linear_kernel_ok(a, k):
            return be.linear(a, w.t().contiguous(), bias, 1.0, None)
    return a @ w if bias is None else torch.addmm(bias, a, w)


def _wgrad(gz, x):
    """(gz^T x, column sums of gz) for gz (rows, n), x (rows, k): mcp_linear_wgrad (rows on the MFMA's contraction axis, fixed-order
    partial sums) where it takes the shape -- n beyond 256 in column blocks of 256 -- and the library GEMM + a reduce otherwise."""
    rows, n = gz.shape
    k = x.shape[1]
    lib = _lib.load() if gz.is_cuda else None
    blocks = [(c0, min(256, n - c0)) for c0 in range(0, n, 256)]
    if lib is None or rows < 2048 or any(lib.mcp_linear_wgrad_workspace_bytes(rows, nb, k) == 0 for _, nb in blocks):
        return gz.t() @ x, gz.sum(dim=0)
    gz, x = gz.contiguous(), x.contiguous()
    dw = torch.empty((n, k), dtype=torch.float32, device=gz.device)
    db = torch.empty((n,), dtype=torch.float32, device=gz.device)
    for c0, nb in blocks:
        need = lib.mcp_linear_wgrad_workspace_bytes(rows, nb, k)
        ws = torch.empty((need,), dtype=torch.uint8, device=gz.device)
        _call("mcp_linear_wgrad", gz, rows, nb, k, gz.data_ptr() + 4 * c0, n, _lib.fptr(x), k, dw.data_ptr() + 4 * c0 * k, db.data_ptr() + 4 * c0,
              ws.data_ptr(), need)
    return dw, db


class _Mlp2Fn(torch.autograd.Function):
    """mcp_mlp2 (Linear, one-slope PReLU, Linear, + residual) with an explicit backward on the streaming kernels: the hidden
    activation is rebuilt once (forward kernel), the two input-gradient products run on the forward kernel with transposed weights,
    the two weight gradients on mcp_linear_wgrad -- instead of autograd over the unfused twin (four library GEMMs on tall, narrow
    operands: 3.6 ms per training step at the pipeline's shapes)."""

    @staticmethod
    def forward(ctx, fused, x, res, w1, b1, w2, b2, slope):
        args = [None if t is None else t.detach() for t in (x, res, w1, b1, w2, b2)]
        sl = slope.detach() if isinstance(slope, torch.Tensor) else slope
        out = fused(*args, sl)
        ctx.slope_is_tensor = isinstance(slope, torch.Tensor)
        ctx.slope = None if ctx.slope_is_tensor else float(slope)
        ctx.has_res = res is not None
        ctx.save_for_backward(x, w1, b1, w2, *( [slope] if ctx.slope_is_tensor else []))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, w1, b1, w2, *rest = ctx.saved_tensors
        slope = rest[0] if ctx.slope_is_tensor else ctx.slope
        hdim, cin = w1.shape
        cout = w2.shape[0]
        x2 = x.reshape(-1, cin)
        gy = grad_out.reshape(-1, cout).contiguous()
        hid = _tall_matmul(x2, w1.t(), b1)
        g_act = _tall_matmul(gy, w2)                                   # (rows, hidden)
        dslope = None
        if ctx.slope_is_tensor and hid.is_cuda:                        # the live PReLU slope: one pass each for act and (g_hid, dslope)
            sl = slope.detach().reshape(1).contiguous()
            hid, g_act = hid.contiguous(), g_act.contiguous()
            act = _prelu_drop_fwd(hid, sl)
            g_hid, dslope = _prelu_drop_bwd(hid, sl, g_act)
            dslope = dslope.reshape(slope.shape) if ctx.needs_input_grad[7] else None
        else:
            pos = hid > 0
            act = torch.where(pos, hid, hid * slope)
            g_hid = torch.where(pos, g_act, g_act * slope)
            if ctx.slope_is_tensor and ctx.needs_input_grad[7]:
                dslope = (g_act * torch.where(pos, torch.zeros_like(hid), hid)).sum().reshape(slope.shape)
        dw2, db2 = _wgrad(gy, act)
        dw1, db1 = _wgrad(g_hid, x2)
        dx = _tall_matmul(g_hid, w1).reshape(x.shape) if ctx.needs_input_grad[1] else None
        dres = grad_out if ctx.has_res and ctx.needs_input_grad[2] else None
        return (None, dx, dres, dw1 if ctx.needs_input_grad[3] else None, db1 if (b1 is not None and ctx.needs_input_grad[4]) else None,
                dw2 if ctx.needs_input_grad[5] else None, db2 if ctx.needs_input_grad[6] else None, dslope)


class _LinearFn(torch.autograd.Function):
    """The fused Linear (+ one-slope activation, + residual) with an explicit backward: two GEMMs and a mask taken from the saved
    activation OUTPUT (for 0 <= slope <= 1 the output of act is positive exactly where its argument is), instead of evaluating the
    layer again under autograd.  With a residual AND an activation the kernel runs without the residual and the sum is made here
    (the same single rounded addition the kernel's epilogue does, so the same bits): the mask then comes from act(z) itself -- a
    mask recovered as (y - res) > 0 loses entries with 0 < act(z) < ulp(res) / 2."""

    @staticmethod
    def forward(ctx, fused, x, w, b, slope, res):
        act_res = res is not None and slope != 1.0
        y = fused(x.detach(), w.detach(), None if b is None else b.detach(), slope, None if (res is None or act_res) else res.detach())
        ctx.slope, ctx.has_b, ctx.has_res = float(slope), b is not None, res is not None
        ctx.save_for_backward(x, w, y if slope != 1.0 else None)
        return y + res.detach() if act_res else y

    @staticmethod
    def backward(ctx, grad_out):
        x, w, act = ctx.saved_tensors
        n, k = w.shape
        gy = grad_out.reshape(-1, n)
        if ctx.slope != 1.0:
            gz = _leaky_grad(gy, act.reshape(-1, n), ctx.slope)
        else:
            gz = gy
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dx = _tall_matmul(gz, w).reshape(x.shape)                  # dx = gz W: the forward kernel again, on W^T
        if ctx.needs_input_grad[2] or (ctx.has_b and ctx.needs_input_grad[3]):
            dw, db = _wgrad(gz, x.reshape(-1, k))                      # dW = gz^T x and db in one kernel (csrc/linear_grad.hip)
            if not (ctx.has_b and ctx.needs_input_grad[3]):
                db = None
        dres = grad_out if ctx.has_res and ctx.needs_input_grad[5] else None
        return None, dx, dw, db, None, dres


class _TorchLinearFn(torch.autograd.Function):
    """A tall per-point Linear the fused kernel does not take (the 3 -> 32 lift, the 32 -> 3 head: widths that are not multiples
    of 4) -- the library's forward, but the backward of _LinearFn: the weight gradient of a (rows x 3) operand is otherwise a
    (3 x rows) x (rows x 32) library GEMM of 300-430 us (round 5 trace) against ~20 for mcp_linear_wgrad."""

    @staticmethod
    def forward(ctx, x, w, b, slope):
        y = torch.nn.functional.linear(x.detach(), w.detach(), None if b is None else b.detach())
        if slope != 1.0:
            y = torch.where(y > 0, y, y * slope)
        ctx.slope, ctx.has_b = float(slope), b is not None
        ctx.save_for_backward(x, w, y if slope != 1.0 else None)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, w, act = ctx.saved_tensors
        n, k = w.shape
        gy = grad_out.reshape(-1, n)
        gz = _leaky_grad(gy, act.reshape(-1, n), ctx.slope) if ctx.slope != 1.0 else gy
        dx = _tall_matmul(gz, w).reshape(x.shape) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dw, db = _wgrad(gz, x.reshape(-1, k))
            if not (ctx.has_b and ctx.needs_input_grad[2]):
                db = None
        return dx, dw, db, None


def plain_linear(x, w, b, slope=1.0):
    """act(x W^T + b) on the library's GEMM with the streaming backward kernels (training forwards, tall inputs only)."""
    return _TorchLinearFn.apply(x, w, b, float(slope))


class _ChamferFn(torch.autograd.Function):
    """chamfer_loss (models/utils.py:36-45, pytorch3d defaults) per sample, with an explicit backward.  Forward: the two nearest-
    neighbour searches return the squared distances themselves (direct form), so the value is two row means -- nothing is gathered or
    re-evaluated.  Backward: d/dx_i = (2 g_b / N)(x_i - y[ixy_i]) - sum over {j: iyx_j = i} of (2 g_b / M)(y_j - x_i), the second term
    as the deterministic segmented scatter of the other direction's direct term (and symmetrically for y when it asks for a gradient:
    the ground truth of the training objective does not).  Autograd over the unfused form (grad.chamfer_twin) took ~45 launches per
    call, 15 calls per training step."""

    @staticmethod
    def forward(ctx, be, x, y):
        x, y = x.detach().contiguous(), y.detach().contiguous()
        ixy, dxy = be.knn(x, y, 1, mode=MCP_DIST_DIRECT, return_dist=True)
        iyx, dyx = be.knn(y, x, 1, mode=MCP_DIST_DIRECT, return_dist=True)
        ctx.save_for_backward(x, y, ixy, iyx)      # the index lists as (B,N,1) / (B,M,1)
        return dxy[..., 0].mean(1) + dyx[..., 0].mean(1)

    @staticmethod
    def backward(ctx, gv):
        x, y, ixy, iyx = ctx.saved_tensors
        B, N, _ = x.shape
        M = y.shape[1]
        gx_d = (x - _group_rows_fwd(y, ixy).squeeze(2)) * (gv * (2.0 / N)).view(B, 1, 1)
        gy_d = (y - _group_rows_fwd(x, iyx).squeeze(2)) * (gv * (2.0 / M)).view(B, 1, 1)
        gx = gx_d - _group_rows_grad(gy_d.unsqueeze(2), iyx, N) if ctx.needs_input_grad[1] else None
        gy = gy_d - _group_rows_grad(gx_d.unsqueeze(2), ixy, M) if ctx.needs_input_grad[2] else None
        return None, gx, gy


class _PtblockFn(torch.autograd.Function):
    """mcp_ptblock_attention with its hand-written backward (mcp_ptblock_grad): the block re-evaluated in the backward kernel, the
    per-neighbour gradients through the deterministic segmented scatter (one sort serves xyz, k and v), weight gradients fixed-order."""

    @staticmethod
    def forward(ctx, be, idx, xyz, q, k, v, *weights):
        xyz = xyz.detach().contiguous()
        q, k, v = q.detach(), k.detach(), v.detach()
        if not (q.stride() == k.stride() == v.stride() and q.stride(2) == 1 and q.stride(0) == q.shape[1] * q.stride(1)):
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ws = [t.detach().contiguous() for t in weights]
        ctx.save_for_backward(idx, xyz, q, k, v, *ws)
        return be.ptblock_attention(xyz, q, k, v, idx, be.ptblock_pack(*ws))

    @staticmethod
    def backward(ctx, grad_out):
        idx, xyz, q, k, v, *ws = ctx.saved_tensors
        B, N, C = q.shape
        lib, dev = _lib.load(), q.device
        grad_out = grad_out.contiguous()
        d_xyz_c = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        d_xyz_rows = torch.empty((B, N, 16, 3), dtype=torch.float32, device=dev)
        d_q = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        d_k_rows = torch.empty((B, N, 16, C), dtype=torch.float32, device=dev)
        d_v_rows = torch.empty((B, N, 16, C), dtype=torch.float32, device=dev)
        d_w = torch.empty((lib.mcp_ptblock_grad_floats(),), dtype=torch.float32, device=dev)
        need = lib.mcp_ptblock_grad_workspace_bytes(B, N)
        wsp = torch.empty((need,), dtype=torch.uint8, device=dev)
        _call("mcp_ptblock_grad", q, B, N, C, 16, q.stride(1), _lib.fptr(xyz), q.data_ptr(), k.data_ptr(), v.data_ptr(), _lib.iptr(idx),
              *[_lib.fptr(t) for t in ws], _lib.fptr(grad_out), _lib.fptr(d_xyz_c), _lib.fptr(d_xyz_rows), _lib.fptr(d_q), _lib.fptr(d_k_rows),
              _lib.fptr(d_v_rows), _lib.fptr(d_w), wsp.data_ptr(), need)
        segments = _scatter_segments(idx, N)
        d_xyz = d_xyz_c + _group_rows_grad(d_xyz_rows, idx, N, segments) if ctx.needs_input_grad[2] else None
        d_k = _group_rows_grad(d_k_rows, idx, N, segments) if ctx.needs_input_grad[4] else None
        d_v = _group_rows_grad(d_v_rows, idx, N, segments) if ctx.needs_input_grad[5] else None
        pieces, at = [], 0
        for t in ws:
            pieces.append(d_w[at:at + t.numel()].view(t.shape))
            at += t.numel()
        return (None, None, d_xyz, d_q, d_k, d_v, *pieces)


class _FusionBNFn(torch.autograd.Function):
    """The fusion layer of one reference call on BATCH statistics (net.train()), forward and backward on the multi-pass kernels of
    csrc/fusion_bn.hip.  Returns (out, bn, var): bn = per layer mean | rstd | gamma | beta, var = the biased batch variances (both
    non-differentiable by-products for the caller's running-estimate update)."""

    @staticmethod
    def forward(ctx, be, eps, ia, ib, p1, p2, w1, b1, w2, b2, w3, b3, g1, e1, g2, e2, g3, e3):
        conv = [t.detach().contiguous() for t in (w1, b1, w2, b2, w3, b3)]
        p1, p2 = p1.detach().contiguous(), p2.detach().contiguous()
        rows = p1.shape[0] * p1.shape[1] * 64
        # kept for the backward: every neighbour's score, its layer-3 channel and zhat3 there (12 bytes per row) -- the backward's first
        # pass then has no layer to re-evaluate
        saved = (torch.empty((rows,), dtype=torch.int32, device=p1.device), torch.empty((rows,), dtype=torch.float32, device=p1.device),
                 torch.empty((rows,), dtype=torch.float32, device=p1.device))
        out, bn, var = be.fusion_bn_forward(p1, p2, ia if ib is None else (ia, ib), conv, (g1, e1, g2, e2, g3, e3), eps, saved=saved)
        ctx.save_for_backward(ia, ib, p1, p2, *conv, bn, *saved)
        ctx.mark_non_differentiable(bn, var)
        return out, bn, var

    @staticmethod
    def backward(ctx, grad_out, _gbn, _gvar):
        ia, ib, p1, p2, w1, b1, w2, b2, w3, b3, bn, save_c, save_z, save_s = ctx.saved_tensors
        B, N, _ = p1.shape
        rows = B * N * 64
        lib, dev = _lib.load(), p1.device
        grad_out = grad_out.contiguous()
        row_c = torch.empty((rows,), dtype=torch.int32, device=dev)
        row_dy, row_a = torch.empty((rows,), dtype=torch.float32, device=dev), torch.empty((rows,), dtype=torch.float32, device=dev)
        dy2, dy1 = torch.empty((rows, 64), dtype=torch.float32, device=dev), torch.empty((rows, 64), dtype=torch.float32, device=dev)
        d_p1 = torch.empty_like(p1)
        d_nb = torch.empty((B, N, 64, 3), dtype=torch.float32, device=dev)
        d_w = torch.empty((lib.mcp_fusion_grad_floats(),), dtype=torch.float32, device=dev)
        d_aff = torch.empty((512,), dtype=torch.float32, device=dev)
        need = lib.mcp_fusion_bn_grad_workspace_bytes(B, N)
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        _call("mcp_fusion_bn_backward_saved", p1, B, N, 64, _lib.fptr(p1), _lib.fptr(p2), _lib.iptr(ia), None if ib is None else _lib.iptr(ib),
              *[_lib.fptr(t) for t in (w1, b1, w2, b2, w3, b3, bn, grad_out)], _lib.iptr(save_c), _lib.fptr(save_z), _lib.fptr(save_s),
              _lib.iptr(row_c), _lib.fptr(row_dy), _lib.fptr(row_a), _lib.fptr(dy2),
              _lib.fptr(dy1), _lib.fptr(d_p1), _lib.fptr(d_nb), _lib.fptr(d_w), _lib.fptr(d_aff), ws.data_ptr(), need)
        d_p2 = None
        if ctx.needs_input_grad[5]:
            d_p2 = _group_rows_grad(d_nb, ia if ib is None else torch.cat((ia, ib), dim=-1), p2.shape[1])
        pieces, at = [], 0
        for t in (w1, b1, w2, b2, w3, b3):
            pieces.append(d_w[at:at + t.numel()].view(t.shape))
            at += t.numel()
        aff = [d_aff[0:64], d_aff[64:128], d_aff[128:192], d_aff[192:256], d_aff[256:384], d_aff[384:512]]
        return (None, None, None, None, d_p1, d_p2, *pieces, *aff)


class _FusionFn(torch.autograd.Function):
    """mcp_fusion with its hand-written backward (mcp_fusion_grad): the layer is re-evaluated inside the backward kernel, the
    neighbour gradients go through the deterministic segmented scatter, weight gradients are fixed-order sums."""

    @staticmethod
    def forward(ctx, be, ia, ib, p1, p2, w1, b1, w2, b2, w3, b3):
        args = [t.detach().contiguous() for t in (p1, p2, w1, b1, w2, b2, w3, b3)]
        ctx.save_for_backward(ia, ib, *args)
        return be._fusion_mlp(args[0], args[1], ia if ib is None else (ia, ib), *args[2:])

    @staticmethod
    def backward(ctx, grad_out):
        ia, ib, p1, p2, w1, b1, w2, b2, w3, b3 = ctx.saved_tensors
        B, N, _ = p1.shape
        lib = _lib.load()
        grad_out = grad_out.contiguous()
        d_p1 = torch.empty_like(p1)
        d_nb = torch.empty((B, N, 64, 3), dtype=torch.float32, device=p1.device)
        d_w = torch.empty((lib.mcp_fusion_grad_floats(),), dtype=torch.float32, device=p1.device)
        need = lib.mcp_fusion_grad_workspace_bytes(B, N)
        ws = torch.empty((need,), dtype=torch.uint8, device=p1.device)
        _call("mcp_fusion_grad", p1, B, N, 64, _lib.fptr(p1), _lib.fptr(p2), _lib.iptr(ia), None if ib is None else _lib.iptr(ib), _lib.fptr(w1),
              _lib.fptr(b1), _lib.fptr(w2), _lib.fptr(b2), _lib.fptr(w3), _lib.fptr(b3), _lib.fptr(grad_out), _lib.fptr(d_p1), _lib.fptr(d_nb),
              _lib.fptr(d_w), ws.data_ptr(), need)
        d_p2 = None
        if ctx.needs_input_grad[4]:
            d_p2 = _group_rows_grad(d_nb, ia if ib is None else torch.cat((ia, ib), dim=-1), p2.shape[1])
        pieces, at = [], 0
        for t in (w1, b1, w2, b2, w3, b3):
            pieces.append(d_w[at:at + t.numel()].view(t.shape))
            at += t.numel()
        return (None, None, None, d_p1, d_p2, *pieces)


class HipBackend:
    name = "hip"

    def fps(self, xyz, npoint, with_points=False):
        """furthest_point_sample (pointnet2_utils.py:10-29): xyz (B,N,3) -> (B,npoint) int32.  with_points: also the sampled
        coordinates (B,npoint,3) -- the index_points_gather that follows in every caller -- from the same launch."""
        xyz = xyz.detach()
        B, N, _ = xyz.shape
        out = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        need = _lib.load().mcp_fps_workspace_bytes(B, N, npoint)  # scratch for the tiled kernel (16384 < N <= 65536), else 0
        ws = torch.empty((need,), dtype=torch.uint8, device=xyz.device) if need else None
        if N <= 65536:  # a fresh sampling: the running distances stay inside the kernel (no (B,N) buffer to fill)
            pts = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device) if with_points else None
            try:
                _call("mcp_furthest_point_sampling_fresh", xyz, B, N, npoint, _lib.fptr(xyz), _lib.iptr(out), None if pts is None else _lib.fptr(pts),
                      ws.data_ptr() if need else None, need)
                return (out, pts) if with_points else out
            except _lib.Unsupported:  # only the streaming kernel applies to this shape: it keeps its running distances in temp
                pass
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        _call("mcp_furthest_point_sampling_ws", xyz, B, N, npoint, _lib.fptr(xyz), _lib.fptr(temp), _lib.iptr(out),
              ws.data_ptr() if need else None, need)
        return (out, self.group_rows(xyz, out)) if with_points else out

    # clouds at least this large go through the Morton-sorted, box-pruned search (same results)
    # (plain attributes: A/B tools set them on the class; the package reads no tuning variables from the environment)
    PRUNE_MIN_REFS = 2048
    PRUNE_MIN_QUERIES = 1024

    def __init__(self):
        self._tls = threading.local()  # .scope: {key: (tensor, sorted cloud, stream, built-event)} while a cloud_scope is open
        self._tile = None

    @property
    def TILE(self):
        if self._tile is None:
            self._tile = _lib.load().mcp_knn_tile_size()
        return self._tile

    @contextlib.contextmanager
    def cloud_scope(self, seed=None):
        """Within this context the Morton-sorted form of a cloud is built once per tensor and reused by every search on it
        (the model searches the same clouds many times per forward).  The caller promises not to write into a cloud tensor
        while the scope is open -- model.forward opens one per call and never does.  Entries hold a strong reference to their
        tensor, so its storage cannot be recycled under the key; the scope is per thread and dropped on exit.  Outside a scope
        nothing is cached: every search rebuilds (one extra launch), so buffers a caller reuses between calls are always safe.
        seed: a dict to use as this scope's table (MoCoPCI.prefetch fills one ahead of the forward that then opens its scope on
        it); a seeded scope replaces an enclosing one until it exits."""
        outer = getattr(self._tls, "scope", None)
        self._tls.scope = seed if seed is not None else ({} if outer is None else outer)
        try:
            yield
        finally:
            self._tls.scope = outer

    def _build_cloud(self, xyz):
        B, N, _ = xyz.shape
        tiles = (N + self.TILE - 1) // self.TILE
        boxes = torch.empty((B, tiles, 6), dtype=torch.float32, device=xyz.device)
        if N <= 16384:  # one fused launch: bbox, Morton keys, in-LDS sort, gather, tile boxes
            sorted_xyz = torch.empty_like(xyz)
            perm = torch.empty((B, N), dtype=torch.int32, device=xyz.device)
            _call("mcp_build_cloud", xyz, B, N, _lib.fptr(xyz), _lib.fptr(sorted_xyz), _lib.iptr(perm), _lib.fptr(boxes))
        else:
            box = torch.cat([xyz.amin(dim=1), xyz.amax(dim=1)], dim=-1).contiguous()
            codes = torch.empty((B, N), dtype=torch.int32, device=xyz.device)
            _call("mcp_morton_codes", xyz, B, N, _lib.fptr(xyz), _lib.fptr(box), _lib.iptr(codes))
            perm = torch.sort(codes, dim=1)[1].int()
            sorted_xyz = self.group_rows(xyz, perm)
            _call("mcp_tile_boxes", xyz, B, N, _lib.fptr(sorted_xyz), _lib.fptr(boxes))
        return sorted_xyz, perm, boxes

    def _sorted_cloud(self, xyz):
        """Morton order of a cloud: (sorted xyz, perm int32, tile boxes); cached only inside a cloud_scope."""
        scope = getattr(self._tls, "scope", None)
        if scope is None:
            return self._build_cloud(xyz)
        cur = torch.cuda.current_stream(xyz.device)
        key = (xyz.data_ptr(), tuple(xyz.shape), xyz.device.index)
        hit = scope.get(key)
        if hit is not None:
            _, cloud, stream, done = hit
            if stream != cur:  # built on another stream of this forward: order + lifetime
                cur.wait_event(done)
                for t in cloud:
                    t.record_stream(cur)
            return cloud
        cloud = self._build_cloud(xyz)
        done = torch.cuda.Event()
        done.record(cur)
        scope[key] = (xyz, cloud, cur, done)
        return cloud

    def prebuild_cloud(self, xyz):
        """Build the Morton-sorted form of a cloud now, on the current stream, when a later search of this cloud_scope will want
        it (a cloud large enough for the pruned search): lets a caller put the build on a side stream, off the critical path.
        No-op outside a scope or for small clouds."""
        if getattr(self._tls, "scope", None) is not None and self.PRUNE_MIN_REFS <= xyz.shape[1] <= 65536:
            self._sorted_cloud(xyz.detach())

    def knn(self, query, ref, k, mode=MCP_DIST_EXPANSION, return_dist=False):
        """knn_point(k, ref, query) (mocopci.py:1158-1169): (B,Q,3),(B,N,3) -> (B,Q,k) int32,
        ascending by (distance, index)."""
        query, ref = query.detach(), ref.detach()  # an index-producing search: no gradient (pointnet2_utils.py:31-33)
        B, Q, _ = query.shape
        N = ref.shape[1]
        idx = torch.empty((B, Q, k), dtype=torch.int32, device=query.device)
        dist = torch.empty((B, Q, k), dtype=torch.float32, device=query.device) if return_dist else None
        if N >= self.PRUNE_MIN_REFS and Q >= self.PRUNE_MIN_QUERIES and k <= 32 and N <= 65536:
            _lib.fptr(query), _lib.fptr(ref)  # validate before building the sorted clouds
            rs, rperm, boxes = self._sorted_cloud(ref)
            same = query.data_ptr() == ref.data_ptr() and query.shape == ref.shape
            qs, qperm, _ = (rs, rperm, boxes) if same else self._sorted_cloud(query)
            _call("mcp_knn_pruned", query, B, Q, N, k, mode, _lib.fptr(qs), _lib.iptr(qperm), _lib.fptr(rs), _lib.iptr(rperm),
                  _lib.fptr(boxes), _lib.iptr(idx), _lib.fptr(dist) if return_dist else None)
        else:
            _call("mcp_knn", query, B, Q, N, k, mode, _lib.fptr(query), _lib.fptr(ref), _lib.iptr(idx),
                  _lib.fptr(dist) if return_dist else None)
        return (idx, dist) if return_dist else idx

    def knn_bruteforce(self, query, ref, k, mode=MCP_DIST_EXPANSION, return_dist=False):
        """The exhaustive kernel regardless of size (tests compare both paths)."""
        B, Q, _ = query.shape
        N = ref.shape[1]
        idx = torch.empty((B, Q, k), dtype=torch.int32, device=query.device)
        dist = torch.empty((B, Q, k), dtype=torch.float32, device=query.device) if return_dist else None
        _call("mcp_knn", query, B, Q, N, k, mode, _lib.fptr(query), _lib.fptr(ref), _lib.iptr(idx),
              _lib.fptr(dist) if return_dist else None)
        return (idx, dist) if return_dist else idx

    def knn_cosine(self, qfeat, rfeat, k, return_dist=False):
        """knn_point_cosine(k, rfeat, qfeat) (pointconv_util.py:111-153) on channel-last features:
        (B,Q,C),(B,N,C) -> (B,Q,k) int32, ascending by (1 - cosine, index).  MFMA kernel, C in {64,128,256}."""
        qfeat, rfeat = qfeat.detach().contiguous(), rfeat.detach().contiguous()
        B, Q, C = qfeat.shape
        N = rfeat.shape[1]
        idx = torch.empty((B, Q, k), dtype=torch.int32, device=qfeat.device)
        dist = torch.empty((B, Q, k), dtype=torch.float32, device=qfeat.device) if return_dist else None
        ws = torch.empty((B * (Q + N) * C,), dtype=torch.float32, device=qfeat.device)
        _call("mcp_knn_cosine", qfeat, B, Q, N, C, k, _lib.fptr(qfeat), _lib.fptr(rfeat), _lib.iptr(idx),
              _lib.fptr(dist) if return_dist else None, _lib.fptr(ws))
        return (idx, dist) if return_dist else idx

    def group_rows(self, points, idx):
        """index_points_group / index_points_gather (mocopci.py:1190-1215): points (B,N,C),
        idx (B,...) int32 -> (B,...,C).  Differentiable w.r.t. points (deterministic segmented-reduction backward)."""
        if points.requires_grad and torch.is_grad_enabled():
            return _GroupRowsFn.apply(points, idx)
        return _group_rows_fwd(points.contiguous(), idx)

    def group_rows_add_leaky(self, points, idx, centre, slope=0.1):
        """leaky(points[idx] + centre[:, :, None, :]): points (B,N,C), idx (B,S,K), centre (B,S,C) -> (B,S,K,C)
        (first layer of the unfused cross(), pointconv_util.py:762-770)."""
        B, N, C = points.shape
        _, S, K = idx.shape
        out = torch.empty((B, S, K, C), dtype=torch.float32, device=points.device)
        _call("mcp_group_rows_add_leaky", points, B, N, C, S, K, ctypes.c_float(slope), _lib.fptr(points), _lib.iptr(idx.contiguous()),
              _lib.fptr(centre), _lib.fptr(out))
        return out

    def _interp3_weights(self, dense, sparse, idx3):
        B, N, _ = dense.shape
        w3 = torch.empty((B, N, 3), dtype=torch.float32, device=dense.device)
        _call("mcp_interp3_weights", dense, B, N, sparse.shape[1], _lib.fptr(dense), _lib.fptr(sparse), _lib.iptr(idx3), _lib.fptr(w3))
        return w3

    def interp3_search(self, dense, sparse):
        """3-NN search + inverse-distance weights of UpsampleFlow (mocopci.py:1494-1498).  The weights are differentiable
        w.r.t. both coordinate sets (the reference's are: warped coordinates depend on predicted flows)."""
        dense, sparse = dense.contiguous(), sparse.contiguous()
        idx3 = self.knn(dense.detach(), sparse.detach(), 3)  # spatially pruned for the large levels, exhaustive below
        w3 = grad.run(self._interp3_weights, lambda d, s_, i: grad.interp3_weights_twin(self.group_rows, d, s_, i), dense, sparse, idx3)
        return idx3, w3

    def interp3_apply(self, feat, idx3, w3):
        """Blend of the three neighbours' rows; differentiable w.r.t. feat and the weights."""
        if grad.wants_grad(feat, w3):
            if not self.EXPLICIT_INTERP3_GRAD:   # A/B switch: autograd over the unfused blend
                return grad.run(lambda f, i, w: _interp3_apply_fwd(f.contiguous(), i.contiguous(), w.contiguous()),
                                lambda f, i, w: grad.interp3_apply_twin(self.group_rows, f, i, w), feat, idx3, w3)
            return _Interp3ApplyFn.apply(feat, idx3, w3)
        return _interp3_apply_fwd(feat.contiguous(), idx3.contiguous(), w3.contiguous())

    EXPLICIT_INTERP3_GRAD = True

    def interp3(self, dense, sparse, feat):
        """UpsampleFlow.forward (mocopci.py:1485-1502): dense (B,N,3), sparse (B,S,3), feat (B,S,C) -> (B,N,C)."""
        B, N, _ = dense.shape
        S, C = feat.shape[1], feat.shape[2]
        # large levels: spatially pruned 3-NN search.  The same two-step route carries the gradients (the fused small-level
        # kernel below has no autograd node), so it is also taken whenever one is wanted.
        if (S >= self.PRUNE_MIN_REFS and N >= self.PRUNE_MIN_QUERIES) or grad.wants_grad(dense, sparse, feat):
            idx3, w3 = self.interp3_search(dense, sparse)
            return self.interp3_apply(feat, idx3, w3)
        idx3 = torch.empty((B, N, 3), dtype=torch.int32, device=dense.device)
        w3 = torch.empty((B, N, 3), dtype=torch.float32, device=dense.device)
        out = torch.empty((B, N, C), dtype=torch.float32, device=dense.device)
        _call("mcp_interp3", dense, B, N, S, C, _lib.fptr(dense), _lib.fptr(sparse), _lib.fptr(feat), _lib.fptr(out),
              _lib.iptr(idx3), _lib.fptr(w3))
        return out

    def fusion_mlp(self, p1, p2, idx, w1, b1, w2, b2, w3, b3):
        """fusion after the neighbour searches (mocopci.py:803-819), BN folded into (w,b): -> (B,N,3).  Differentiable."""
        if not grad.wants_grad(p1, p2, w1, b1, w2, b2, w3, b3):
            return self._fusion_mlp(p1, p2, idx, w1, b1, w2, b2, w3, b3)
        ia, ib = idx if isinstance(idx, (tuple, list)) else (idx, None)
        if ia.shape[-1] + (0 if ib is None else ib.shape[-1]) != 64:   # the kernels take the 2 x 32 lists only: autograd over the unfused twin
            return grad.fusion_twin(self.group_rows, p1, p2, idx, w1, b1, w2, b2, w3, b3)
        return _FusionFn.apply(self, ia.contiguous(), None if ib is None else ib.contiguous(), p1, p2, w1, b1, w2, b2, w3, b3)

    def fusion_bn(self, p1, p2, idx, conv, affine, eps):
        """The fusion layer of ONE reference call on batch statistics, differentiable w.r.t. p1, p2, the conv weights and the
        BatchNorm weight / bias: -> (out (B,N,3), bn, var) -- see _FusionBNFn."""
        ia, ib = idx if isinstance(idx, (tuple, list)) else (idx, None)
        if ia.shape[-1] + (0 if ib is None else ib.shape[-1]) != 64:
            raise ValueError("fusion_bn: the batch-statistics kernels take 64 neighbours per point (2 x 32 or one list of 64)")
        return _FusionBNFn.apply(self, float(eps), ia.contiguous(), None if ib is None else ib.contiguous(), p1, p2, *conv, *affine)

    def fusion_bn_forward(self, p1, p2, idx, conv, affine, eps, saved=None):
        """The fusion layer of ONE reference call on batch statistics (net.train(); mocopci.py:810-819 with nn.BatchNorm2d in training
        mode): conv = (w1, b1, w2, b2, w3, b3) raw conv weights, affine = (gamma1, beta1, gamma2, beta2, gamma3, beta3).
        -> out (B,N,3), bn (1024: per layer mean | rstd | gamma | beta), var (256: biased batch variances 64 | 64 | 128).  No autograd.
        saved: (int32, float32, float32) tensors of B*N*64 elements that receive every neighbour's arg-max channel, zhat3 there and score
        (mcp_fusion_bn_forward_save; what mcp_fusion_bn_backward_saved reads)."""
        lib = _lib.load()
        p1, p2 = p1.contiguous(), p2.contiguous()
        conv = [t.detach().contiguous() for t in conv]
        B, N, _ = p1.shape
        ia, ib = idx if isinstance(idx, (tuple, list)) else (idx, None)
        bn = torch.empty((lib.mcp_fusion_bn_floats(),), dtype=torch.float32, device=p1.device)
        at = 0
        for c, (g, b) in zip((64, 64, 128), zip(affine[0::2], affine[1::2])):
            bn[at + 2 * c:at + 3 * c] = g.detach()
            bn[at + 3 * c:at + 4 * c] = b.detach()
            at += 4 * c
        var = torch.empty((256,), dtype=torch.float32, device=p1.device)
        out = torch.empty((B, N, 3), dtype=torch.float32, device=p1.device)
        need = lib.mcp_fusion_bn_workspace_bytes(B, N)
        ws = torch.empty((need,), dtype=torch.uint8, device=p1.device)
        if saved is None:
            _call("mcp_fusion_bn_forward", p1, B, N, 64, _lib.fptr(p1), _lib.fptr(p2), _lib.iptr(ia), None if ib is None else _lib.iptr(ib),
                  *[_lib.fptr(t) for t in conv], float(eps), _lib.fptr(bn), _lib.fptr(var), _lib.fptr(out), ws.data_ptr(), need)
        else:
            _call("mcp_fusion_bn_forward_save", p1, B, N, 64, _lib.fptr(p1), _lib.fptr(p2), _lib.iptr(ia), None if ib is None else _lib.iptr(ib),
                  *[_lib.fptr(t) for t in conv], float(eps), _lib.fptr(bn), _lib.fptr(var), _lib.fptr(out), _lib.iptr(saved[0]), _lib.fptr(saved[1]),
                  _lib.fptr(saved[2]), ws.data_ptr(), need)
        return out, bn, var

    def _fusion_mlp(self, p1, p2, idx, w1, b1, w2, b2, w3, b3):
        p1, p2, w1, b1, w2, b2, w3, b3 = (t.contiguous() for t in (p1, p2, w1, b1, w2, b2, w3, b3))
        B, N, _ = p1.shape
        out = torch.empty((B, N, 3), dtype=torch.float32, device=p1.device)
        ia, ib = idx if isinstance(idx, (tuple, list)) else (idx, None)   # the two searches' lists as they are, or one (B,N,64) list
        nb = ia.shape[-1] if ib is None else ia.shape[-1] + ib.shape[-1]
        _call("mcp_fusion", p1, B, N, nb, _lib.fptr(p1), _lib.fptr(p2), _lib.iptr(ia), None if ib is None else _lib.iptr(ib), _lib.fptr(w1), _lib.fptr(b1),
              _lib.fptr(w2), _lib.fptr(b2), _lib.fptr(w3), _lib.fptr(b3), _lib.fptr(out))
        return out

    def cross_pack(self, wpos, bpos, wmlp, bmlp):
        """Pack one cross layer's weights into the kernel's MFMA-operand image (do this once per layer)."""
        D = wmlp.shape[0]
        lib = _lib.load()
        n = lib.mcp_cross_packed_floats(D)
        if n == 0:
            raise RuntimeError(f"cross_volume supports D in (64, 128, 256), got {D}")
        packed = torch.empty((n,), dtype=torch.float32, device=wmlp.device)
        _call("mcp_cross_pack", wmlp, D, _lib.fptr(wpos.contiguous()), _lib.fptr(bpos.contiguous()), _lib.fptr(wmlp.contiguous()),
              _lib.fptr(bmlp.contiguous()), _lib.fptr(packed))
        return packed

    def cross_volume(self, xyz1, xyz2, points1, points2, idx, packed, bmap=None, shared=0):
        """cross() after its neighbour searches (pointconv_util.py:750-781): -> (B,N1,D); D in {64,128,256};
        packed = cross_pack(wpos, bpos, wmlp, bmlp).  bmap (B int32) + shared (1 points1, 2 points2, 4 first index list): those
        tensors hold a smaller batch and element b is read from their element bmap[b] (replicated inputs are not copied)."""
        B, N1 = xyz1.shape[0], xyz1.shape[1]
        D = points1.shape[2]
        N2 = xyz2.shape[1]
        out = torch.empty((B, N1, D), dtype=torch.float32, device=points1.device)
        ia, ib = idx if isinstance(idx, (tuple, list)) else (idx, None)   # the 16 + 16 halves as the searches produce them, or one list
        k = ia.shape[-1] if ib is None else ia.shape[-1] + ib.shape[-1]
        if bmap is not None and (shared & 4) and ib is None:
            raise RuntimeError("a shared index list needs the two-list form")
        _call("mcp_cross_volume", points1, B, N1, N2, D, k, _lib.fptr(xyz1), _lib.fptr(xyz2), _lib.fptr(points1),
              _lib.fptr(points2), _lib.iptr(ia), None if ib is None else _lib.iptr(ib), None if bmap is None else _lib.iptr(bmap), int(shared),
              _lib.fptr(packed), _lib.fptr(out))
        return out

    def cross_layer(self, xyz1, xyz2, points1, points2, idx, wpos, bpos, wmlp, bmlp, packed=None, bmap=None, shared=0):
        """cross() from the layer's own weights; differentiable w.r.t. coordinates, features and weights.  packed: the
        cross_pack image of these weights when the caller keeps one (inference); built on the fly otherwise.  bmap / shared:
        see cross_volume (inference only: a training forward passes the replicated tensors)."""
        def fused(x1, x2, f1, f2, i, wp, bp, wm, bm):
            pk = packed if packed is not None else self.cross_pack(wp, bp, wm, bm)
            return self.cross_volume(x1.contiguous(), x2.contiguous(), f1.contiguous(), f2.contiguous(), i, pk, bmap=bmap, shared=shared)
        if bmap is not None:
            return fused(xyz1, xyz2, points1, points2, idx, wpos, bpos, wmlp, bmlp)
        ia, ib = idx if isinstance(idx, (tuple, list)) else (idx, None)
        k_total = ia.shape[-1] + (0 if ib is None else ib.shape[-1])   # the backward kernel is built for the 32-neighbour lists
        if grad.wants_grad(xyz1, xyz2, points1, points2, wpos, bpos, wmlp, bmlp) and k_total == 32 and _lib.load().mcp_cross_grad_floats(wmlp.shape[0]):
            return _CrossFn.apply(self, ia.contiguous(), None if ib is None else ib.contiguous(), xyz1, xyz2, points1, points2, wpos, bpos, wmlp, bmlp)
        return grad.run(fused, lambda *a: grad.cross_twin(self.group_rows, *a), xyz1, xyz2, points1, points2, idx, wpos, bpos, wmlp, bmlp)

    def ptblock_layer(self, xyz, q, k, v, idx, weights, packed=None):
        """TransformerBlock vector attention from the block's own weights (wd1,bd1,wd2,bd2,wg1,bg1,wg2,bg2); differentiable."""
        def fused(x, q_, k_, v_, i, *w):
            pk = packed if packed is not None else self.ptblock_pack(*w)
            return self.ptblock_attention(x.contiguous(), q_, k_, v_, i, pk)
        if grad.wants_grad(xyz, q, k, v, *weights) and q.shape[-1] == 64 and idx.shape[-1] == 16:
            return _PtblockFn.apply(self, idx.contiguous(), xyz, q, k, v, *weights)
        return grad.run(fused, lambda *a: grad.ptblock_twin(self.group_rows, *a), xyz, q, k, v, idx, *weights)

    def ptblock_pack(self, wd1, bd1, wd2, bd2, wg1, bg1, wg2, bg2):
        """Pack fc_delta / fc_gamma of a TransformerBlock (pointT_layer2.py:42-51) into the kernel's operand image."""
        packed = torch.empty((_lib.load().mcp_ptblock_packed_floats(),), dtype=torch.float32, device=wd2.device)
        args = [t.contiguous() for t in (wd1, bd1, wd2, bd2, wg1, bg1, wg2, bg2)]
        _call("mcp_ptblock_pack", wd2, *[_lib.fptr(t) for t in args], _lib.fptr(packed))
        return packed

    def ptblock_attention(self, xyz, q, k, v, idx, packed):
        """Vector attention of TransformerBlock.forward (pointT_layer2.py:68-75) after knn + projections: -> (B,N,64).
        q, k, v: (B,N,64) tensors, or last-axis slices of one packed (B,N,192) projection (read in place through the row stride)."""
        B, N, C = q.shape
        rs = q.stride(1)
        for t in (q, k, v):
            if not (t.is_cuda and t.dtype == torch.float32 and t.shape == (B, N, C) and t.stride() == (N * rs, rs, 1)):
                raise RuntimeError("q, k, v must be float32 CUDA tensors of one shape with unit channel stride and a common row stride")
        out = torch.empty((B, N, C), dtype=torch.float32, device=q.device)
        _call("mcp_ptblock_attention", q, B, N, C, idx.shape[-1], rs, _lib.fptr(xyz), q.data_ptr(), k.data_ptr(), v.data_ptr(), _lib.iptr(idx),
              _lib.fptr(packed), _lib.fptr(out))
        return out

    def pointconv_agg(self, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2):
        """PointConv grouping + WeightNet + aggregation (mocopci.py:1330-1335): -> (B,S,(3+D)*8).  Differentiable."""
        D = s_points.shape[-1]
        if not grad.wants_grad(s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2):
            return self._pointconv_agg(s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2)
        if idx.shape[-1] != 32 or D % 4 or D > 256 or tuple(w2.shape) != (8, 8):   # shapes the backward kernel is not built for
            return grad.run(self._pointconv_agg, lambda *a: grad.pointconv_agg_twin(self.group_rows, *a), s_xyz, new_xyz, s_points, idx,
                            w0, b0, w1, b1, w2, b2)
        return _PointConvAggFn.apply(self, idx.contiguous(), s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2)

    def _pointconv_agg(self, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2):
        s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2 = (t.contiguous() for t in (s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2))
        B, N, D = s_points.shape
        S = new_xyz.shape[1]
        out = torch.empty((B, S, (D + 3) * 8), dtype=torch.float32, device=s_points.device)
        _call("mcp_pointconv_agg", s_points, B, N, S, D, idx.shape[-1], _lib.fptr(s_xyz), _lib.fptr(new_xyz), _lib.fptr(s_points),
              _lib.iptr(idx), _lib.fptr(w0), _lib.fptr(b0), _lib.fptr(w1), _lib.fptr(b1), _lib.fptr(w2), _lib.fptr(b2), _lib.fptr(out))
        return out

    _PCL_MIN_ROWS = 16384   # from here mcp_linear runs linear_kernel, whose arithmetic the one-launch form reproduces bit for bit

    def pointconv_linear_supported(self, d, c_out, k=32, rows=None):
        """Shapes mcp_pointconv_linear is built for (levels 0 / 1 of the encoder and the refinement stage's PointConvD).  With
        `rows` (= B * S centres): whether the model should use it -- only where its output is bit-identical to pointconv_agg +
        linear, i.e. where that Linear would run the full-K kernel and not the split-K / library forms of the few-row launches."""
        return k == 32 and (d, c_out) in ((32, 32), (64, 64)) and (rows is None or rows >= self._PCL_MIN_ROWS)

    def pointconv_linear_pack(self, w, b):
        """Operand image of PointConv's Linear: the same image linear() uses for the (B,S,(3+D)*8) aggregate."""
        return self.linear_pack(w, b, [w.shape[1]])

    def pointconv_linear(self, s_xyz, new_xyz, s_points, idx, w0, b0, w1, b1, w2, b2, w, b, slope, packed=None):
        """PointConv after the sampling in one launch (mocopci.py:1330-1342): grouping, WeightNet, aggregation, Linear, LeakyReLU ->
        (B,S,C_out).  Inference only (a training forward uses pointconv_agg + linear, which carry gradients).
        packed: pointconv_linear_pack(w, b) if kept."""
        s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2 = (t.contiguous() for t in (s_xyz, new_xyz, s_points, w0, b0, w1, b1, w2, b2))
        B, N, D = s_points.shape
        S, n = new_xyz.shape[1], w.shape[0]
        pk = packed if packed is not None else self.pointconv_linear_pack(w, b)
        out = torch.empty((B, S, n), dtype=torch.float32, device=s_points.device)
        _call("mcp_pointconv_linear", s_points, B, N, S, D, idx.shape[-1], _lib.fptr(s_xyz), _lib.fptr(new_xyz), _lib.fptr(s_points),
              _lib.iptr(idx), _lib.fptr(w0), _lib.fptr(b0), _lib.fptr(w1), _lib.fptr(b1), _lib.fptr(w2), _lib.fptr(b2), _lib.fptr(pk), n,
              float(slope), _lib.fptr(out))
        return out

    def attention(self, q, kv, heads, scale=None, dropout_p=0.0):
        """softmax(q k^T * scale) v per head, reading the projection outputs in place: q (BF,Nq,C), kv (BF,Nk,2C)
        laid out [k | v] as the reference's kv Linear produces (mocopci.py:74-75, :653-654) -> (BF,Nq,C).  dropout_p > 0 (head dims 8 /
        16 only): attention dropout on the softmax matrix inside the kernel, the mask a counter-based hash seeded from torch's CPU
        generator (reproducible under torch.manual_seed; see mcp_attention_small_dropout)."""
        BF, Nq, C = q.shape
        Nk = kv.shape[1]
        hd = C // heads
        if scale is None:
            scale = hd ** -0.5
        if dropout_p > 0.0:
            if hd not in (8, 16):
                raise RuntimeError("attention dropout inside the kernel: head dims 8 / 16")
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())  # CPU generator: no device synchronisation
            if grad.wants_grad(q, kv):
                return _AttentionSmallFn.apply(self, q, kv, heads, float(scale), float(dropout_p), seed)
            return self._attention(q, kv, heads, float(scale), float(dropout_p), seed)
        if hd not in (8, 16, 32, 64, 256):  # head dims neither kernel is built for: the dense formulation (same arithmetic as the twin)
            return grad.attention_twin(q, kv, heads, float(scale))
        if hd in (8, 16) and grad.wants_grad(q, kv):
            return _AttentionSmallFn.apply(self, q, kv, heads, float(scale), 0.0, 0)
        return grad.run(self._attention, grad.attention_twin, q, kv, heads, float(scale))

    def _attention(self, q, kv, heads, scale, drop_p=0.0, seed=0):
        q, kv = q.contiguous(), kv.contiguous()
        BF, Nq, C = q.shape
        Nk, hd = kv.shape[1], C // heads
        out = torch.empty((BF, Nq, C), dtype=torch.float32, device=q.device)
        if drop_p > 0.0:
            _call("mcp_attention_small_dropout", q, BF, Nq, Nk, heads, hd, q.data_ptr(), C, kv.data_ptr(), 2 * C, kv.data_ptr() + 4 * C, 2 * C,
                  float(scale), float(drop_p), int(seed), out.data_ptr())
            return out
        # head dims 8/16: S on MFMA, P.V on packed FMAs; 32/64/256 (ei3, Cross_Frame_Att): both products on MFMA
        name = "mcp_attention_small" if hd in (8, 16) else "mcp_attention_wide"
        _call(name, q, BF, Nq, Nk, heads, hd, q.data_ptr(), C, kv.data_ptr(), 2 * C, kv.data_ptr() + 4 * C, 2 * C,
              float(scale), out.data_ptr(), C)
        return out

    def prelu_dropout(self, z, slope, drop_p):
        """dropout(prelu(z, slope), drop_p) of a training forward (Mlp_T's act + drop, mocopci.py:1561-1562) as one pass, with a
        one-pass backward; slope: the layer's 1-element parameter.  The mask is a counter-based hash seeded from torch's CPU
        generator per call (reproducible under torch.manual_seed), as in attention()."""
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())  # CPU generator: no device synchronisation
        if slope.numel() != 1:
            raise RuntimeError("prelu_dropout: one slope for all channels (nn.PReLU())")
        return _PreluDropFn.apply(z, slope, float(drop_p), seed)

    def attention_rot(self, q, k, v, heads, kv_shift, scale=None):
        """Attention over one stacked batch whose keys / values come from batch element (b + kv_shift) mod BF: q, k, v are
        (BF,N,C) views with unit channel stride (e.g. column slices of one packed projection, read in place).  Inference only."""
        BF, Nq, C = q.shape
        Nk, hd = k.shape[1], C // heads
        for t, n in ((q, Nq), (k, Nk), (v, Nk)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(0) == n * t.stride(1)):
                raise RuntimeError("attention_rot: (BF,N,C) float32 CUDA views with unit channel stride and dense batches")
        out = torch.empty((BF, Nq, C), dtype=torch.float32, device=q.device)
        _call("mcp_attention", q, BF, Nq, Nk, heads, hd, q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), v.data_ptr(), v.stride(1),
              int(kv_shift), float(hd ** -0.5 if scale is None else scale), out.data_ptr(), C)
        return out

    def add_layernorm(self, x, y=None, bias=None, eps=1e-6):
        """(z - mean z) * rsqrt(var z + eps) over the last axis, z = x (+ y) (+ bias): one kernel (csrc/norm.hip).  Inference only."""
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        y2 = None if y is None else y.reshape(-1, C)
        if x2.stride(1) != 1 or (y2 is not None and y2.stride(1) != 1):
            raise RuntimeError("add_layernorm: unit channel stride")
        out = torch.empty((x2.shape[0], C), dtype=torch.float32, device=x.device)
        _call("mcp_add_layernorm", x, x2.shape[0], C, _lib.fptr(x2) if x2.is_contiguous() else x2.data_ptr(), x2.stride(0),
              None if y2 is None else y2.data_ptr(), 0 if y2 is None else y2.stride(0), None if bias is None else _lib.fptr(bias), None, None,
              float(eps), _lib.fptr(out), C)
        return out.reshape(x.shape)

    def mfa_prepare(self, fea, src_self, src_partner, te_self, te_partner, scale, shift):
        """(x, xn, xr) of Multi_Frame_Att from the computed flow embeddings in one launch (csrc/norm.hip): fea (M,N,C), index lists
        (R,) int32, time codes (R,C), norm1's eval scale / shift (C) -> three (R,N,C) tensors.  Inference only."""
        M, N, C = fea.shape
        R = src_self.shape[0]
        x, xn, xr = (torch.empty((R, N, C), dtype=torch.float32, device=fea.device) for _ in range(3))
        _call("mcp_mfa_prepare", fea, R, N, C, _lib.fptr(fea), _lib.iptr(src_self), _lib.iptr(src_partner), _lib.fptr(te_self), _lib.fptr(te_partner),
              _lib.fptr(scale), _lib.fptr(shift), _lib.fptr(x), _lib.fptr(xn), _lib.fptr(xr))
        return x, xn, xr

    # ---- per-point Linear with fused epilogue (csrc/linear.hip) ----
    @staticmethod
    def _pieces(xs):
        """The input pieces as 2-D row views (rows, k_i) with unit channel stride, 16-byte aligned rows; None if one of them
        cannot be read in place."""
        out = []
        for t in (xs if isinstance(xs, (tuple, list)) else (xs,)):
            k = t.shape[-1]
            v = t.reshape(-1, k) if t.is_contiguous() else (t if t.dim() == 2 else None)
            if v is None and t.stride(-1) == 1:  # a last-axis slice of a contiguous tensor: rows keep the parent's stride
                lead = t.shape[:-1]
                st = t.stride()
                ok = all(st[i] == st[i + 1] * lead[i + 1] for i in range(len(lead) - 1))
                v = t.as_strided((t.numel() // k, k), (st[-2], 1)) if ok else None
            if v is None or v.stride(1) != 1 or v.stride(0) % 4 or v.data_ptr() % 16 or k % 4 or v.dtype != torch.float32 or not v.is_cuda:
                return None
            out.append(v)
        return out

    def linear_supported(self, xs, n, few_rows=True, policy_rows=None):
        """Whether linear() takes this call.  policy_rows: decide (and run, see linear()) as for that many rows -- a caller that
        computes a subset of the rows of a tall product asks for the tall product's kernel, so that both give the same bits.  Beyond what the kernel can do, a shape policy: the fused kernel beats the BLAS
        chain for tall inputs with moderate K (tools/linear_ab.py, MI355X: 1.3-1.9x at rows >= 16384, K <= 320, e.g. 58 -> 31 us
        for 196608 x 32 -> 64 + LeakyReLU) and, since its K loop is staged four chunks per barrier, for the K = 536 -> 64 PointConv
        projections (171 -> 146 us at 196608 rows); it loses for few rows (one 256-row workgroup per 8 waves: 2048-8192 rows do
        not cover the chip) and for long K at wide N (2072 -> 256), which stay on the library.  few_rows=False: without the
        split-K few-row kernel (a training forward keeps the library's fp32 GEMMs there, as the gradient tests were pinned)."""
        if self._NO_LINEAR:
            return False
        ps = self._pieces(xs)
        if ps is None or len(ps) > 3 or len({p.shape[0] for p in ps}) != 1:
            return False
        k = sum(p.shape[1] for p in ps)
        rows = ps[0].shape[0] if policy_rows is None else int(policy_rows)
        if few_rows and self._LIN_FEW_MIN_ROWS <= rows < self._LIN_MIN_ROWS:
            # few rows (the lower pyramid levels): the split-K kernel -- four waves share a 32-row tile and split its K chunks -- beats
            # the library GEMM + activation pair up to 256 output columns (tools/linear_fewrows_ab.py: 33 -> 26 us at 4096 x 1048 ->
            # 256, 25 -> 12 us at 8192 x 128 -> 128); wider outputs re-read and re-split x per 64-column block and lose
            if n > 256 or k < 64 or k > self._LIN_FEW_MAX_K:
                return False
        elif rows < self._LIN_MIN_ROWS or n > self._LIN_MAX_N or (k > self._LIN_MAX_K and not (k <= 2 * self._LIN_MAX_K and n <= 64)):
            return False
        ks = (ctypes.c_int * len(ps))(*[p.shape[1] for p in ps])
        return _lib.load().mcp_linear_packed_floats(n, len(ps), ks) != 0

    def linear_kernel_ok(self, x2d, n):
        """Whether the fused Linear KERNEL can take (rows, K) -> n at all (capability, not the inference shape policy above): used by
        the backward products, where the alternative is a library GEMM on a tall, narrow operand."""
        ps = self._pieces(x2d)
        if self._NO_LINEAR or ps is None or len(ps) != 1:
            return False
        ks = (ctypes.c_int * 1)(ps[0].shape[1])
        return _lib.load().mcp_linear_packed_floats(n, 1, ks) != 0

    # shape policy of the fused Linear (measured optimum; tools/linear_ab.py overrides them on the class)
    _NO_LINEAR = False
    _NO_NARROW = False
    _LIN_MIN_ROWS = 16384
    _LIN_FEW_MIN_ROWS = 2048   # rows from which the split-K few-row kernel is used (0 < rows < _LIN_MIN_ROWS); 1 << 30 turns it off
    _LIN_FEW_MAX_K = 1100
    _LIN_MAX_K = 320
    _LIN_MAX_N = 192

    def linear_pack(self, w, b, ks):
        """Operand image of one Linear whose K axis is the concatenation of pieces of widths ks (split once per layer)."""
        n = w.shape[0]
        kk = (ctypes.c_int * len(ks))(*ks)
        nf = _lib.load().mcp_linear_packed_floats(n, len(ks), kk)
        if nf == 0:
            raise RuntimeError(f"linear does not support n={n}, pieces={list(ks)}")
        packed = torch.empty((nf,), dtype=torch.float32, device=w.device)
        _call("mcp_linear_pack", w, n, len(ks), kk, _lib.fptr(w.contiguous()), None if b is None else _lib.fptr(b.contiguous()), _lib.fptr(packed))
        return packed

    def linear(self, xs, w, b=None, slope=1.0, res=None, packed=None, policy_rows=None):
        """act(W [x_0 | x_1 | x_2] + b) [+ res] over the last axis (policy_rows: the kernel of a product with that many rows, see
        linear_supported): xs one tensor or up to three pieces of a concatenation (read in
        place, column slices allowed); act(v) = v > 0 ? v : slope v (1.0: none, 0.1: LeakyReLU, 0: ReLU, a PReLU slope).
        One kernel instead of cat + GEMM + activation + add.  Differentiable.  packed: linear_pack(w, b, widths) if kept."""
        def fused(xs_, w_, b_, slope_, res_):
            ps = self._pieces(xs_)
            ks = [p.shape[1] for p in ps]
            pk = packed if packed is not None else self.linear_pack(w_, b_, ks)
            rows, n = ps[0].shape[0], w_.shape[0]
            first = xs_[0] if isinstance(xs_, (tuple, list)) else xs_
            out = torch.empty((rows, n), dtype=torch.float32, device=first.device)
            r2 = None if res_ is None else res_.reshape(rows, n).contiguous()
            xp = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
            st = (ctypes.c_int * len(ps))(*[p.stride(0) for p in ps])
            kk = (ctypes.c_int * len(ps))(*ks)
            _call("mcp_linear_as", first, rows, rows if policy_rows is None else int(policy_rows), n, len(ps), xp, st, kk, float(slope_), _lib.fptr(pk),
                  None if r2 is None else _lib.fptr(r2), n, _lib.fptr(out), n)
            return out.reshape(*first.shape[:-1], n)
        if isinstance(xs, (tuple, list)) and grad.wants_grad(*xs, w, b, res):
            # training: the pieces are concatenated (autograd splits the gradient again) so that the layer takes the explicit
            # backward below -- the streaming dx / dW kernels -- instead of plain autograd over library GEMMs
            cat = torch.cat(list(xs), dim=-1)
            if isinstance(slope, (int, float)) and 0.0 <= slope <= 1.0 and self.linear_supported(cat, w.shape[0], few_rows=False):
                return _LinearFn.apply(fused, cat, w, b, float(slope), res)
            return grad.linear_twin(cat, w, b, slope, res)
        if grad.wants_grad(xs, w, b, res) and isinstance(slope, (int, float)) and 0.0 <= slope <= 1.0:
            return _LinearFn.apply(fused, xs, w, b, float(slope), res)
        return grad.run(fused, grad.linear_twin, xs, w, b, slope, res)

    def linear_narrow_supported(self, rows, k, n):
        """act-then-Linear with at most 4 outputs over K in {256, 512, 1024} (see linear_narrow)."""
        return not (self._NO_LINEAR or self._NO_NARROW) and n <= 4 and k in (256, 512, 1024) and rows >= 1024

    def linear_narrow(self, x, w, b, in_slope):
        """b + W . act(x) over the last axis, act(v) = v > 0 ? v : in_slope v, W (n <= 4, K): the flow tail of Mlp_T (PReLU, then
        fc2 and mapping_xyz as one 4C -> 3 map) as one streaming kernel instead of an activation launch and a 3-column GEMM."""
        def fused(x_, w_, b_, slope_):
            x2 = x_.reshape(-1, x_.shape[-1])
            x2 = x2 if x2.stride(1) == 1 and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0 else x2.contiguous()
            rows, k, n = x2.shape[0], x2.shape[1], w_.shape[0]
            out = torch.empty((rows, n), dtype=torch.float32, device=x_.device)
            _call("mcp_linear_narrow", x_, rows, k, n, _lib.fptr(x2) if x2.is_contiguous() else x2.data_ptr(), x2.stride(0), _lib.fptr(w_.contiguous()),
                  None if b_ is None else _lib.fptr(b_.contiguous()), float(slope_), _lib.fptr(out), n)
            return out.reshape(*x_.shape[:-1], n)
        twin = lambda x_, w_, b_, slope_: torch.nn.functional.linear(torch.where(x_ > 0, x_, x_ * slope_), w_, b_)
        return grad.run(fused, twin, x, w, b, in_slope)

    def mlp2_pack(self, w1, b1, w2, b2):
        """Operand image of one two-layer MLP for mlp2 (split once; do this once per block)."""
        hidden, cin = w1.shape
        cout = w2.shape[0]
        n = _lib.load().mcp_mlp2_packed_floats(cin, hidden, cout)
        if n == 0:
            raise RuntimeError(f"mlp2 does not support cin={cin}, hidden={hidden}, cout={cout}")
        packed = torch.empty((n,), dtype=torch.float32, device=w1.device)
        _call("mcp_mlp2_pack", w1, cin, hidden, cout, _lib.fptr(w1.contiguous()), _lib.fptr(b1.contiguous()), _lib.fptr(w2.contiguous()),
              _lib.fptr(b2.contiguous()), _lib.fptr(packed))
        return packed

    def mlp2(self, x, w1, b1, w2, b2, slope, res=None, packed=None):
        """out = [res +] W2 act(W1 x + b1) + b2 over the last axis, act = PReLU with one slope: Mlp_T and the flow heads of the
        frame-attention blocks (mocopci.py:1558-1565, :566-567, :510-511) as ONE kernel; the hidden activation is never written.
        x (..., cin) -> (..., cout); slope: a float, or the 1-element PReLU parameter (then it receives a gradient too).
        Differentiable w.r.t. x, res and the weights.  packed: mlp2_pack(w1, b1, w2, b2) when the caller keeps one."""
        def fused(x_, res_, w1_, b1_, w2_, b2_, slope_):
            pk = packed if packed is not None else self.mlp2_pack(w1_, b1_, w2_, b2_)
            cin, hidden, cout = w1_.shape[1], w1_.shape[0], w2_.shape[0]
            x2 = x_.reshape(-1, cin)
            if not (x2.stride(1) == 1 and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0):
                x2 = x2.contiguous()
            r2 = None if res_ is None else res_.reshape(-1, cout).contiguous()
            out = torch.empty((x2.shape[0], cout), dtype=torch.float32, device=x_.device)
            _call("mcp_mlp2", x_, x2.shape[0], cin, hidden, cout, float(slope_), x2.data_ptr(), x2.stride(0), None if r2 is None else _lib.fptr(r2), cout,
                  _lib.fptr(pk), _lib.fptr(out), cout)
            return out.reshape(*x_.shape[:-1], cout)
        if grad.wants_grad(x, res, w1, b1, w2, b2, slope) and b2 is not None:
            return _Mlp2Fn.apply(fused, x, res, w1, b1, w2, b2, slope)
        return grad.run(fused, grad.mlp2_twin, x, res, w1, b1, w2, b2, slope)

    def mlp2_supported(self, cin, hidden, cout):
        """Shapes the fused kernel is built for (the ones where it beats the BLAS chain); A/B tools set _NO_MLP2 on the class."""
        return not self._NO_MLP2 and _lib.load().mcp_mlp2_packed_floats(cin, hidden, cout) != 0

    _NO_MLP2 = False

    EXPLICIT_CHAMFER_GRAD = True

    def chamfer(self, x, y, per_sample=False):
        """chamfer_loss (models/utils.py:36-45; pytorch3d defaults): x (B,N,3), y (B,M,3) -> 0-dim tensor.  As a training loss
        (train.py:135-160) it is differentiable w.r.t. both clouds: the nearest neighbours come from the search kernel and the
        squared distances to them are re-evaluated differentiably.  per_sample: the (B,) values whose mean that is -- several terms
        of the objective that share a ground-truth cloud are then ONE call on a stacked batch (training.multiscale_loss)."""
        if grad.wants_grad(x, y):
            if self.EXPLICIT_CHAMFER_GRAD:
                v = _ChamferFn.apply(self, x, y)
                return v if per_sample else v.mean()
            x, y = x.contiguous(), y.contiguous()   # A/B: autograd over the unfused form
            ixy = self.knn(x.detach(), y.detach(), 1, mode=MCP_DIST_DIRECT)[..., 0].contiguous()
            iyx = self.knn(y.detach(), x.detach(), 1, mode=MCP_DIST_DIRECT)[..., 0].contiguous()
            return grad.chamfer_twin(self.group_rows, x, y, ixy, iyx, per_sample)
        B, N, _ = x.shape
        M = y.shape[1]
        dxy = torch.empty((B, N), dtype=torch.float32, device=x.device)
        dyx = torch.empty((B, M), dtype=torch.float32, device=x.device)
        _call("mcp_chamfer_nn", x, B, N, M, _lib.fptr(x), _lib.fptr(y), _lib.fptr(dxy), _lib.fptr(dyx))
        v = dxy.mean(1) + dyx.mean(1)
        return v if per_sample else v.mean()


_backend = HipBackend()


def backend():
    return _backend


def set_backend(b):
    """Swap the operator backend (tests / cpu_baseline only). Returns the previous one."""
    global _backend
    prev, _backend = _backend, b
    return prev


# ---- instrumentation passthrough (bench.py) ----
KERNEL_IDS = {"fps": 1, "knn": 2, "group_rows": 3, "interp3": 4, "knn_cosine": 5, "fusion": 6, "cross": 7, "pointconv": 8, "attention": 9, "ptblock": 10, "mlp": 11, "linear": 12}


def prof_enable(kernel_names):
    """Time every launch of the named kernels (iterable of KERNEL_IDS keys) with hipEvents; None/() disables."""
    mask = 0
    for name in (kernel_names or ()):
        mask |= 1 << KERNEL_IDS[name]
    _lib.check(_lib.load().mcp_prof_enable(mask))


def prof_collect(kernel_name):
    n = ctypes.c_int(0)
    ms = ctypes.c_float(0.0)
    _lib.check(_lib.load().mcp_prof_collect(KERNEL_IDS[kernel_name], ctypes.byref(n), ctypes.byref(ms)))
    return n.value, ms.value
