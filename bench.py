#!/usr/bin/env python3
"""bench.py -- interpolated frames/s of the MoCoPCI point-set hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1 invoked plainly: this process spawns `python -m torch.distributed.run --nproc-per-node N ... bench.py ...`, one rank per
     GPU, and relays rank 0's JSON line; under torch.distributed.run already: reads RANK/LOCAL_RANK/WORLD_SIZE from the env)

A step = one forward of the interpolation graph (mocopci_amd.model.MoCoPCI, eval mode) over one batch of synthetic 4-frame
sequences already resident in HBM; it yields 3 interpolated frames per sequence (mocopci.py:822,1053), so
frames/s = 3 * B_total * K / t.  Workload at every N: BASELINE.json configs[1]/[2] -- N=8192 points, 8 sequences per GPU
(weak scaling; configs[2] is 8 GPUs x 8).  Multi-GPU: sequences shard across ranks with no data-path collective; the only
exchange is the final all_gather of the output frames over RCCL (inside the timed step).

Extra objects on the JSON line:
  roofline          the dominant hand-written kernel of the step's critical (main) stream, timed live with hipEvents around every
                    launch on its launch stream inside the timed region; achieved = algorithmic flops or bytes per launch
                    (SURVEY.md 8(d) per-unit figure x units per launch) / average launch duration
  roofline_others   the same for every other hand-written kernel family (FPS runs on a side stream, beside the main stream)
  parity            sequence 0 of this very run against the REFERENCE'S stored forward (tests/golden/forward_c2_n8192.npz)
  cpu_baseline      the same graph on the CPU oracle backend (oracle/, "port") for one full batch, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (~6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3    # f32-input MFMA (v_mfma_f32_32x32x2_f32): 64 FLOP/clk/SIMD
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
SPLIT_PRODUCTS = 6              # bf16 partial products per fp32 product on the split path (mocopci_amd/csrc/mfma_split.h)
# What a pure v_mfma_f32_32x32x16_bf16 stream sustains on this chip once the operand bits toggle (tools/mfma_peak.py --random,
# profiles/r02_mfma_sustained.txt: power management holds the clock near 1.8 GHz; 0.99 of the datasheet figure with all-ones operands).
# Reported beside `peak`, never instead of it.
MFMA_BF16_SUSTAINED_TFLOPS = 1865.0
NPOINTS = 8192     # the default workload (--config c2); main() rebinds both for --config c4
B_PER_GPU = 8
HBM_COPY_PEAK_GBS = 6290.0      # measured float4 copy (MI355X_MICROARCH.md; SURVEY 8(d) prices the point-set path against it)
# --config: which BASELINE.json configuration the run measures.  c2 (default) is the one the metric is quoted on (configs[1]; configs[2]
# is the same per-GPU workload on 8 GPUs); c4 / c5 are the "stress" and "roofline" configurations (configs[3], configs[4]).
CONFIGS = {
    "c2": dict(config_id=2, npoints=8192, batch=8, kw={}, golden="forward_c2_n8192",
               workload="KITTI-o-like NL-Drive synthetic, N=8192, batch=8 per GPU, 3 interp frames (BASELINE configs[1]; configs[2] at 8 GPUs)"),
    "c4": dict(config_id=4, npoints=16384, batch=8, kw=dict(extent=50.0, zlo=-5.0, zhi=3.0), golden=None,
               workload="NuScenes-like NL-Drive synthetic, N=16384 (x,y in +-50, z in [-5,3]), batch=8, 3 interp frames (BASELINE configs[3]: LDS-tile / "
                        "ball-query radius stress) + the standalone ball_query (M=2048, r in {0.5,1,2,4}) and three_nn launches of SURVEY 8(d)"),
}
FAMILIES = ("fps", "knn", "knn_cosine", "fusion", "cross", "pointconv", "attention", "ptblock", "mlp", "linear")
HEADLINE = "fusion"  # the single kernel symbol with the most time on a step's critical (main) stream
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc.json")   # tools/profile_round.sh output of this round (counters per launch)
PMC_STAMP = os.path.join(ROOT, "profiles", "r05_pmc_sources.json")  # sha256 of every csrc file the profiled library was built from

NAMES = {
    "fps": "fps_spatial_kernel / fps_resident_kernel (mcp_furthest_point_sampling_ws)",
    "knn": "knn_walk (K = 32) / knn_pruned (K <= 16) / knn_queue / knn_small kernels (mcp_knn, mcp_knn_pruned)",
    "knn_cosine": "knn_cosine_kernel (mcp_knn_cosine)",
    "fusion": "fusion_split_kernel (mcp_fusion)",
    "cross": "cross_kernel<64|128> / cross256_stream_kernel (mcp_cross_volume)",
    "pointconv": "pointconv_linear_kernel / pointconv_agg_kernel (mcp_pointconv_linear, mcp_pointconv_agg)",
    "attention": "attention_small_kernel<8|16> / attention_wide_kernel<32|256> (mcp_attention_small, mcp_attention_wide)",
    "ptblock": "ptblock_kernel (mcp_ptblock_attention)",
    "mlp": "mlp2_kernel (mcp_mlp2)",
    "linear": "linear_kernel (mcp_linear)",
}
NOTES = {
    "fps": "achieved/frac = SURVEY 8(d)'s figure: bytes of the REFERENCE'S streaming formulation B*(M-1)*20*N over kernel time -- NOT a "
           "utilisation: the kernel keeps points and running distances in VGPRs/LDS, its real HBM traffic is the compulsory "
           "B*(16N+4M) (compulsory_GBs, ~0.1 % of peak; PMC traffic agrees) and it is bound by the latency of M-1 dependent "
           "iterations (us_per_iteration), one workgroup per batch element; it runs on a side stream beside the main stream",
    "knn": "achieved = compulsory bytes B*(12Q+12N+4QK) per search (SURVEY 8d); with the distance matrix gone the search is bound by "
           "VALU issue (list maintenance), see valu_busy_frac_of_chip from the PMC pass",
    "knn_cosine": "2*B*Q*N*C flop on the f32-input MFMA (exact fp32 fma chains: the neighbour indices are compared bit for bit)",
    "fusion": "4->64->64->128 MLP per neighbour, 1.6 MFLOP/point; fp32 products on the bf16 matrix pipe through an exact 3-way operand "
              "split: peak = bf16 dense MFMA peak / 6 partial products per fp32 product",
    "cross": "B*N1*K*(8C+2C^2) flop; every D on the split-bf16 path (peak as for fusion)",
    "pointconv": "gather + WeightNet + aggregation on the VALU; achieved = gathered rows + aggregate written, B*S*(K*4*(D+3) + 32*(D+3)) bytes "
                 "(the gathers hit L2 / Infinity Cache).  The D = 32 / 64 launches (levels 0, 1, refinement) also run the Linear + LeakyReLU behind "
                 "the aggregation (mcp_pointconv_linear): their aggregate stays in LDS, so they move LESS than this figure and do more",
    "attention": "4*BF*H*Nq*Nk*hd flop; S = QK^T on the f32-input MFMA, softmax and (head dims 8/16) P.V on the VALU",
    "ptblock": "B*N*16*2*(3*64 + 3*64^2) flop; the three 64x64 layers on the split-bf16 path",
    "mlp": "2*rows*(C*H + H*C_out) flop of the fused Mlp_T / flow-head blocks on the split-bf16 path, weights streamed through LDS",
    "linear": "tall per-point Linear layers with fused activation / residual: bytes read + written, 4*rows*(K+n) (memory-bound by design)",
}


def kernel_sources_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile): tools/profile_round.sh stores it beside the counters it collects,
    and a bench run whose sources differ flags the counters it quotes as stale instead of passing them off as this build's."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "mocopci_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h")) + [os.path.join(base, "Makefile")]):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


def pmc_is_stale():
    try:
        return json.load(open(PMC_STAMP))["csrc_sha256"] != kernel_sources_digest()
    except (OSError, KeyError, ValueError):
        return True


PMC_IS_STALE = pmc_is_stale()


def algorithmic_work(kernel, calls):
    """SURVEY.md 8(d) per-unit figures x the units each launch processes (call shapes logged from one untimed step)."""
    if kernel == "fps":
        return sum(b * (m - 1) * 20 * n for (b, n, m) in calls), "bytes"
    if kernel == "knn":
        return sum(b * (12 * q + 12 * n + 4 * q * k) for (b, q, n, k) in calls), "bytes"
    if kernel == "knn_cosine":
        return sum(2 * b * q * n * c for (b, q, n, c) in calls), "flops"
    if kernel == "fusion":
        return sum(b * n * 2 * 64 * (4 * 64 + 64 * 64 + 64 * 128) for (b, n) in calls), "flops"
    if kernel == "cross":
        return sum(b * n1 * 32 * (8 * d + 2 * d * d) for (b, n1, d) in calls), "flops"
    if kernel == "pointconv":
        return sum(b * s * (32 * 4 * (d + 3) + 32 * (d + 3)) for (b, s, d) in calls), "bytes"
    if kernel == "attention":
        return sum(4 * bf * h * nq * nk * hd for (bf, h, nq, nk, hd) in calls), "flops"
    if kernel == "ptblock":
        return sum(b * n * 16 * 2 * (3 * 64 + 3 * 64 * 64) for (b, n) in calls), "flops"
    if kernel == "mlp":
        return sum(2 * rows * (c * h + h * co) for (rows, c, h, co) in calls), "flops"
    if kernel == "linear":
        return sum(4 * rows * (k + n) for (rows, k, n) in calls), "bytes"
    raise KeyError(kernel)


def compulsory_bytes(kernel, calls):
    """SURVEY.md 8(d) COMPULSORY bytes of the point-set families (inputs read once + outputs written once), for the end-to-end line."""
    if kernel == "fps":
        return sum(b * (12 * n + 4 * m) for (b, n, m) in calls)
    if kernel == "knn":
        return sum(b * (12 * q + 12 * n + 4 * q * k) for (b, q, n, k) in calls)
    if kernel == "knn_cosine":
        return sum(b * (4 * c * (q + n) + 4 * q * 16) for (b, q, n, c) in calls)
    if kernel == "cross":   # 4B(2C N1 + C N2 + 3(N1+N2) + K N1), N2 = N1 at every call of this graph
        return sum(4 * b * (2 * d * n1 + d * n1 + 3 * 2 * n1 + 32 * n1) for (b, n1, d) in calls)
    if kernel == "pointconv":   # group(C=D+3,S,K=32): B(4SK + 8CSK) is the REFERENCE's materialised group; compulsory: neighbour lists + features read once + output
        return sum(b * s * (4 * 32 + 4 * (d + 3) + 4 * d) for (b, s, d) in calls)
    if kernel == "fusion":      # two clouds + 64 neighbour indices in, one cloud out
        return sum(b * n * (12 + 12 + 4 * 64 + 12) for (b, n) in calls)
    if kernel == "ptblock":     # xyz + q,k,v rows + 16 neighbour indices in, 64 channels out
        return sum(b * n * (12 + 3 * 256 + 4 * 16 + 256) for (b, n) in calls)
    raise KeyError(kernel)


POINTSET_FAMILIES = ("fps", "knn", "knn_cosine", "cross", "pointconv", "fusion", "ptblock")


def pointset_end_to_end(timed, calls, steps):
    """SURVEY 8(d)'s end-to-end line: (sum of compulsory bytes of all point-set kernels per forward) / (time in those kernels),
    against the measured HBM copy peak.  It prices the whole point-set path as if it were a streaming pass; it is far below the peak
    because none of these kernels is HBM-bound once the N x N matrices are gone (their own bounds are in the per-family entries)."""
    total_b, total_ms, per = 0.0, 0.0, {}
    for k in POINTSET_FAMILIES:
        launches, kms = timed.get(k, (0, 0.0))
        if not launches:
            continue
        b = compulsory_bytes(k, calls[k])
        per[k] = {"compulsory_MB_per_step": b / 1e6, "kernel_ms_per_step": kms / steps}
        total_b += b
        total_ms += kms / steps
    ach = total_b / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0
    return {"what": "sum of SURVEY 8(d) compulsory bytes of the point-set kernel families per step / time in those kernels (instrumented pass)",
            "compulsory_MB_per_step": total_b / 1e6, "kernel_ms_per_step": total_ms, "achieved_GBs": ach, "peak_GBs": HBM_COPY_PEAK_GBS,
            "frac_of_measured_copy_peak": ach / HBM_COPY_PEAK_GBS, "frac_of_spec_peak": ach / HBM_PEAK_GBS, "families": per}


def log_call_shapes(be, step):
    """One untimed step with shape-recording wrappers on the backend's entry points.  Exactly one record per timed launch: the
    3-NN searches inside interp3 / interp3_search reach the timer through be.knn, so only be.knn records them; the fused
    small-level mcp_interp3 call (its own KNN launch inside the library) records here."""
    calls = {k: [] for k in FAMILIES}
    names = ("fps", "knn", "interp3", "knn_cosine", "fusion_mlp", "cross_volume", "pointconv_agg", "pointconv_linear", "attention", "attention_rot", "ptblock_attention", "mlp2", "linear",
             "linear_narrow")
    orig = {n: getattr(be, n) for n in names}

    def wrap(name, rec):
        def f(*a, **k):
            rec(*a, **k)
            return orig[name](*a, **k)
        return f

    def rec_interp3(d, s_, f):
        if not (s_.shape[1] >= be.PRUNE_MIN_REFS and d.shape[1] >= be.PRUNE_MIN_QUERIES):
            calls["knn"].append((d.shape[0], d.shape[1], s_.shape[1], 3))

    be.fps = wrap("fps", lambda xyz, m, **kw: calls["fps"].append((xyz.shape[0], xyz.shape[1], m)))
    be.knn = wrap("knn", lambda q, r, k, **kw: calls["knn"].append((q.shape[0], q.shape[1], r.shape[1], k)))
    be.interp3 = wrap("interp3", rec_interp3)
    be.knn_cosine = wrap("knn_cosine", lambda q, r, k, **kw: calls["knn_cosine"].append((q.shape[0], q.shape[1], r.shape[1], q.shape[2])))
    be.fusion_mlp = wrap("fusion_mlp", lambda p1, *a: calls["fusion"].append((p1.shape[0], p1.shape[1])))
    be.cross_volume = wrap("cross_volume", lambda x1, x2, f1, *a, **k: calls["cross"].append((x1.shape[0], x1.shape[1], f1.shape[2])))
    be.pointconv_agg = wrap("pointconv_agg", lambda sx, nx, sp, *a: calls["pointconv"].append((nx.shape[0], nx.shape[1], sp.shape[2])))
    be.pointconv_linear = wrap("pointconv_linear", lambda sx, nx, sp, *a, **k: calls["pointconv"].append((nx.shape[0], nx.shape[1], sp.shape[2])))
    be.attention = wrap("attention", lambda q, kv, h, **kw: calls["attention"].append((q.shape[0], h, q.shape[1], kv.shape[1], q.shape[2] // h)))
    be.attention_rot = wrap("attention_rot", lambda q, k, v, h, *a, **kw: calls["attention"].append((q.shape[0], h, q.shape[1], k.shape[1], q.shape[2] // h)))
    be.ptblock_attention = wrap("ptblock_attention", lambda xyz, q, *a: calls["ptblock"].append((q.shape[0], q.shape[1])))
    be.linear = wrap("linear", lambda xs, w, *a, **k: calls["linear"].append(((xs[0] if isinstance(xs, (tuple, list)) else xs).numel()
                                                                               // (xs[0] if isinstance(xs, (tuple, list)) else xs).shape[-1],
                                                                               w.shape[1], w.shape[0])))
    be.linear_narrow = wrap("linear_narrow", lambda x, w, *a, **k: calls["linear"].append((x.numel() // x.shape[-1], w.shape[1], w.shape[0])))
    be.mlp2 = wrap("mlp2", lambda x, w1, b1, w2, *a, **k: calls["mlp"].append((x.numel() // x.shape[-1], w1.shape[1], w1.shape[0], w2.shape[0])))
    try:
        step()
    finally:
        for n in orig:
            delattr(be, n)  # back to the class methods
    return calls


def roofline_entries(timed, calls, steps, pmc):
    entries = []
    for kname in FAMILIES:
        launches, kms = timed[kname]
        if not launches or kms <= 0:
            continue
        # one record per timed launch: anything else would inflate the algorithmic work
        assert len(calls[kname]) * steps == launches, (kname, len(calls[kname]), steps, launches)
        work, unit = algorithmic_work(kname, calls[kname])
        per_launch = work * steps / launches
        avg_s = kms * 1e-3 / launches
        if unit == "bytes":
            ach, peak, u, bound = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
        else:
            peak = MFMA_BF16_PEAK_TFLOPS / SPLIT_PRODUCTS if kname in ("fusion", "cross", "ptblock", "mlp") else MFMA_F32_PEAK_TFLOPS
            ach, u, bound = per_launch / avg_s / 1e12, "TFLOP/s", "mfma"
        e = {"kernel": NAMES[kname], "bound": bound, "achieved": ach, "peak": peak, "unit": u, "frac": ach / peak,
             "traffic": None, "launches": launches, "launches_per_step": launches // steps, "avg_launch_us": 1e6 * avg_s,
             "kernel_ms_per_step": kms / steps, "note": NOTES[kname]}
        if unit != "bytes" and kname in ("fusion", "cross", "ptblock", "mlp"):
            e["sustained_peak"] = MFMA_BF16_SUSTAINED_TFLOPS / SPLIT_PRODUCTS
            e["frac_of_sustained_peak"] = ach / e["sustained_peak"]
            e["sustained_peak_source"] = "profiles/r02_mfma_sustained.txt (measured bf16 MFMA stream with random operand bits: 0.75 of the datasheet peak)"
        p = pmc.get(kname, {})
        if "hbm_bytes_per_launch" in p:
            e["traffic"] = p["hbm_bytes_per_launch"]
            e["traffic_source"] = "profiles/r05_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this round's build, read side doubled per the gfx950 rule)"
            e["traffic_stale"] = PMC_IS_STALE
        for key in ("valu_busy_frac_of_chip", "mfma_busy_frac_of_chip", "mfma_valu_coexec_frac_of_chip", "simd_issue_busy_frac_of_chip",
                    "mean_resident_waves_per_simd"):
            if key in p:
                e[key] = p[key]
        if "simd_issue_busy_frac_of_chip" in p and unit != "bytes":
            e["simd_issue_note"] = ("matrix pipe and VALU of a SIMD do not overlap on this chip, within or across waves (profiles/r05_overlap2_probe.txt): a fused "
                                    "MFMA + VALU kernel is bound by the SUM of its two instruction streams; simd_issue_busy = mfma + valu - coexec is its utilisation")
        if kname == "fps":
            comp = sum(b * (16 * n + 4 * m) for (b, n, m) in calls["fps"]) * steps / launches
            iters = sum(m - 1 for (b, n, m) in calls["fps"]) * steps / launches
            e["bound_in_practice"] = "latency of the M-1 dependent iterations"
            e["compulsory_GBs"] = comp / avg_s / 1e9
            e["compulsory_frac_of_hbm_peak"] = comp / avg_s / 1e9 / HBM_PEAK_GBS
            e["us_per_iteration"] = 1e6 * avg_s / iters
            e["reference_streaming_equivalent_GBs"] = ach
        entries.append(e)
    return entries


def parity_vs_reference(frames, rank):
    """Sequence 0 of rank 0's batch is the seeded cloud of tests/golden/forward_c2_n8192.npz (config 2, sample 0): compare this
    run's frames with the reference's stored forward (oracle/make_golden.py).  Data file only; the checker code lives in tests/."""
    path = os.path.join(ROOT, "tests", "golden", "forward_c2_n8192.npz")
    if rank != 0 or not os.path.exists(path):
        return None
    import numpy as np
    from mocopci_amd import ops
    g = np.load(path)
    out = []
    for j in range(3):
        want = np.ascontiguousarray(g["out%d" % j])
        got = frames[0:1, j].detach().cpu().numpy()
        spread2 = float(((want - want.mean(axis=1, keepdims=True)) ** 2).sum(-1).mean())
        disp2 = ((got - want) ** 2).sum(-1)
        cd = float(ops.backend().chamfer(frames[0:1, j].contiguous(), torch.from_numpy(want).to(frames.device)))
        out.append({"points_moved_over_2pct_of_spread": float((disp2 > 4e-4 * spread2).mean()), "mse_over_spread2": float(disp2.mean() / spread2),
                    "chamfer_to_reference_frame": cd, "chamfer_to_reference_over_spread2": cd / spread2})
    return {"what": "sequence 0 of this run vs the REFERENCE'S stored forward at N=8192 (tests/golden/forward_c2_n8192.npz)", "frames": out}


def quality_on_scan_weights(dev):
    """Chamfer / EMD against the synthetic ground truth under the SECOND weight set (synth.weights_on_scan: a coordinate-carrying
    path through the refinement branch keeps the predicted frames on the scan), sequence 0 of config 2 at N = 8192, outside the
    timed region.  With the stress weights of the timed run these metrics are E|gt|^2 whatever the kernels do; here they are
    bounded by the smoothing radius of the 32-neighbourhoods and move when a kernel is wrong (tests/test_harness_cpu.py)."""
    from mocopci_amd import emd as emd_mod, ops, synth
    from mocopci_amd.model import MoCoPCI
    net = MoCoPCI()
    net.load_state_dict(synth.weights_on_scan(net._spec), strict=True)
    net = net.to(dev)
    x1, x2, gt = synth.make_batch(2, 1, NPOINTS, device=dev)
    out = net(x1, x2)
    cd = [float(ops.backend().chamfer(out[j].contiguous(), gt[j])) for j in range(3)]
    emd = [float(emd_mod.EMD(out[j].permute(0, 2, 1).contiguous(), gt[j].permute(0, 2, 1).contiguous())) for j in range(3)]
    e_gt2 = float((gt[0] ** 2).sum(-1).mean())
    ident = [float(ops.backend().chamfer(x1.transpose(1, 2).contiguous(), gt[j])) for j in range(3)]
    return {"weights": "synth.weights_on_scan (deterministic, hand-built coordinate-carrying refinement path; no trained checkpoint exists offline)",
            "workload": f"sequence 0 of config 2, N={NPOINTS}, B=1, outside the timed region",
            "chamfer_vs_gt": cd, "emd_vs_gt": emd, "chamfer_of_copying_frame_1": ident, "mean_sq_norm_of_gt": e_gt2}


def run_train_step(args):
    """One JSON line: milliseconds per training step on one GPU."""
    from mocopci_amd import synth, training
    from mocopci_amd.model import MoCoPCI
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    net = MoCoPCI()
    net.load_state_dict(synth.weights_by_name(net._spec), strict=True)
    net = net.to(dev)
    net.train(args.train_step == "train")
    opt = torch.optim.Adam(net.parameters(), lr=1e-5)
    x1, x2, gt = synth.make_batch(2, CONFIGS["c2"]["batch"], CONFIGS["c2"]["npoints"], device=dev)
    gtc = [g.transpose(1, 2).contiguous() for g in gt]
    losses = []
    for _ in range(max(1, args.warmup)):
        losses.append(training.train_step(net, opt, x1, x2, gtc)[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(training.train_step(net, opt, x1, x2, gtc)[0])   # returns the loss as a float: one host sync per step, as train.py has
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    print(json.dumps({"metric": "training step time", "value": ms, "unit": "ms", "higher_is_better": False, "n_gpus": 1, "steps": args.steps,
                      "warmup": max(1, args.warmup), "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"MoCoPCI training step, N={CONFIGS['c2']['npoints']}, B={CONFIGS['c2']['batch']}, module mode {args.train_step}",
                                 "objective": "train.py:135-160 multi-scale Chamfer, clip 2.0, Adam"},
                      "loss_first_last": [losses[0], losses[-1]], "peak_memory_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
                      "vs_baseline": None}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=("c2", "c4", "c5"), default="c2",
                    help="BASELINE.json configuration: c2 = configs[1]/[2] (N=8192, B=8 per GPU: the one the metric is quoted on, default); "
                         "c4 = configs[3] (N=16384, B=8, model forward + ball_query / three_nn launches); c5 = configs[4] (N=65536, B=8: "
                         "FPS 65536->2048 + 32-NN with Q=2048 and Q=N, kernels only -- the model is not run at this size, SURVEY 8(d))")
    ap.add_argument("--serial", action="store_true", help="do not pipeline the input-only sampling pyramid across consecutive steps")
    ap.add_argument("--pipeline", choices=("two-batch", "prefetch", "event"), default="two-batch",
                    help="two-batch: prefetch + the tail of batch k-1 (after its refinement-stage sampling) is enqueued behind the first part of "
                         "batch k; prefetch: the next step's input-only work is issued under this step's decoder; event: issued at the next call (round 2)")
    ap.add_argument("--streams", type=int, default=1, help="steps in flight: consecutive steps issued round-robin on this many HIP streams "
                                                            "(measured: 2-3 give 0-10 %% depending on how the runtime maps the ~15 streams onto its 4 "
                                                            "hardware queues, not reproducibly; more hardware queues make it worse)")
    ap.add_argument("--train-step", choices=("eval", "train"), default=None,
                    help="NOT the BASELINE metric: time one training step (forward(train=True) + train.py's multi-scale Chamfer objective + "
                         "backward + clipped Adam, mocopci_amd/training.py) at the configs[1] shape instead; eval = the inference graph "
                         "differentiated, train = after net.train() (batch-statistics BatchNorm, dropout: the reference's mode, train.py:130)")
    args = ap.parse_args()
    if args.train_step is not None:
        if args.gpus != 1:
            raise SystemExit("--train-step is a single-GPU measurement")
        return run_train_step(args)

    global NPOINTS, B_PER_GPU
    if args.config == "c5":
        if args.gpus != 1:
            raise SystemExit("--config c5 is a single-GPU kernel run")
        return run_c5(args)
    cfg = CONFIGS[args.config]
    NPOINTS, B_PER_GPU = cfg["npoints"], cfg["batch"]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Invoked plainly (`python bench.py --gpus N`): fan out to one fresh rank process per GPU, as the reference's only
        # multi-GPU mechanism does with threads (nn.DataParallel, train.py:73-80).  This process has not touched the GPU
        # (no torch.cuda call yet) and never will: it only relays rank 0's JSON line and the launcher's exit code.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)  # one rank per GPU on a real node; ranks wrap only in single-GPU rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("MCP_DIST_BACKEND", "nccl"), device_id=dev if ndev >= world else None)

    from mocopci_amd import ops, synth, shard
    from mocopci_amd.model import MoCoPCI

    net = MoCoPCI()
    net.load_state_dict(synth.weights_by_name(net._spec), strict=True)
    net = net.to(dev)
    # config 2 (N=8192, B=8 per GPU); rank r holds sequences [8r, 8r+8) of the global batch
    x1, x2, gt = synth.make_batch(cfg["config_id"], B_PER_GPU, NPOINTS, device=dev, first_sample=rank * B_PER_GPU, **cfg["kw"])

    # The inputs are resident and complete: an event recorded here lets the input-only sampling pyramid of step k+1 be issued
    # behind it instead of behind step k's tail (MoCoPCI.forward(inputs_ready=...): pipelining of consecutive batches, as a
    # loader's copy-stream event would allow in serving).  --serial keeps every step strictly behind the previous one.
    inputs_ready = None
    if not args.serial:
        inputs_ready = torch.cuda.Event()
        inputs_ready.record()

    # Steps in flight (--streams, default 1).  Consecutive steps are independent and a single step leaves much of the chip idle
    # for stretches (FPS chains on 16-24 CUs, the small-kernel stages of the lower pyramid levels), so they can be issued
    # round-robin on a few HIP streams, each with its own side streams.  Measured gain 0-10 %, not reproducible from run to run
    # (it depends on how the runtime maps the streams onto its hardware queues), hence not the default.
    n_streams = 1 if args.serial else max(1, args.streams)
    # how consecutive steps overlap: "prefetch" (default), "event" (round 2: the input-only work of step k+1 is issued when step k+1
    # is called, behind the inputs' event), "serial"
    pipeline = "serial" if args.serial else ("event" if n_streams > 1 else args.pipeline)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream(dev)]
    for ln in lanes:
        ln.wait_stream(torch.cuda.current_stream(dev))
    counter = [0]

    handle = [None]
    pending = [None]

    def step(last=False):
        """One step of the serving loop.  Default pipelining ("two-batch"): consecutive batches are software-pipelined on one stream --
        this call enqueues batch k's forward up to the launch of its refinement-stage furthest point sampling (MoCoPCI.begin),
        then the rest of batch k-1 (MoCoPCI.finish), so that sampling chain runs beside real work instead of stalling the stream; the
        input-only work of batch k+1 is issued right after batch k's encoder (then_prefetch).  `last`: the phase ends here -- batch
        k is finished too and nothing is prefetched, so a phase (warm-up, timed region, instrumented pass) contains exactly one
        pyramid, one first part and one second part per step of its own: nothing is borrowed from the phase before, nothing is
        left over.  Returns the frames of the most recently finished batch."""
        ln = lanes[counter[0] % n_streams]
        counter[0] += 1
        with torch.cuda.stream(ln):
            if pipeline in ("two-batch", "prefetch"):
                h = handle[0] if handle[0] is not None else net.prefetch(x1, x2, inputs_ready)
                nxt = None if last else (x1, x2, inputs_ready)
                if pipeline == "two-batch":
                    cur = net.begin(x1, x2, prefetched=h, then_prefetch=nxt)
                    out = net.finish(pending[0]) if pending[0] is not None else None
                    pending[0] = cur
                    if last:
                        out = net.finish(cur)
                        pending[0] = None
                else:
                    out = net(x1, x2, prefetched=h, then_prefetch=nxt)
                handle[0] = net.take_prefetched()
            else:
                out = net(x1, x2, inputs_ready=inputs_ready)   # 3 x (B,N,3)
            return None if out is None else shard.gather_frames(out, world)  # (world*B,3,N,3) on every rank; no-op view for world == 1

    for i in range(args.warmup):
        step(last=i == args.warmup - 1)
    calls = log_call_shapes(ops.backend(), lambda: step(last=True))
    for _ in range(n_streams - 1):  # every lane has run at least once before the clock starts
        step(last=True)

    # Inside the timed region only the headline kernel is bracketed by hipEvents (two records per step).  Bracketing all nine
    # families (~45 launches per step) costs ~0.4 ms per step of stream time and serialises the steps in flight, so the other
    # families are timed in a second, instrumented pass right after the timed region (same process, same inputs, one stream).
    ops.prof_enable(() if os.environ.get("MCP_BENCH_NOPROF") == "1" else (HEADLINE,))  # MCP_BENCH_NOPROF: A/B of the event overhead only
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frames = step(last=i == args.steps - 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    headline_timed = ops.prof_collect(HEADLINE)
    inst_steps = min(args.steps, 10)
    ops.prof_enable(FAMILIES)
    for i in range(inst_steps):
        step(last=i == inst_steps - 1)   # the same calls as in the timed region (same kernels and shapes)
    torch.cuda.synchronize()
    timed = {k: ops.prof_collect(k) for k in FAMILIES}
    ops.prof_enable(None)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    local = frames[rank * B_PER_GPU:(rank + 1) * B_PER_GPU]
    chamfer = [float(ops.backend().chamfer(local[:, j].contiguous(), gt[j])) for j in range(3)]
    # second metric of test.py:90 (approximate EMD, per-point normalised as models/utils.py:223-235), outside the timed region
    from mocopci_amd import emd as emd_mod
    emd = [float(emd_mod.EMD(local[:, j].permute(0, 2, 1).contiguous(), gt[j].permute(0, 2, 1).contiguous())) for j in range(3)]

    total_frames = 3 * B_PER_GPU * world * args.steps
    result = {
        "metric": "interpolated frames/sec",
        "value": total_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # fp32 values end to end; where a fused MLP layer runs on the matrix pipe, each fp32 product is the sum of six bf16 x bf16 partial
        # products of an exact 3-way operand split (fp32 accumulation): fp32-class accuracy (~4 ulp from the f32-input MFMA build), and
        # the neighbour-index paths (FPS, KNN, cosine KNN) stay on exact f32 arithmetic
        "dtype": "f32 (fused MLP layers: fp32 products as 6 bf16-MFMA partials of an exact 3-way split, fp32 accumulate; index paths exact f32)",
        "data": "synthetic",
        "config": {"workload": cfg["workload"], "bench_config": args.config,
                   "npoints": NPOINTS, "batch_per_gpu": B_PER_GPU, "global_batch": B_PER_GPU * world,
                   "parallelism": f"sequence-sharded x{world}, final all_gather over RCCL" if world > 1 else "single GPU",
                   "weights": "deterministic by-name synthetic, eval mode",
                   "steps_in_flight": n_streams,
                   "step_pipelining": {"serial": "off (--serial)",
                                       "event": "the furthest-point-sampling pyramid and the level-0 self search of step k+1 (input-only, side streams) are issued "
                                                "when step k+1 is called, behind the inputs' ready event, and overlap the tail of step k",
                                       "two-batch": "consecutive batches are software-pipelined on one stream: step k enqueues batch k up to the launch of its "
                                                    "refinement-stage furthest point sampling, then the rest of batch k-1 (Point-Transformer refinement, fusion), so that "
                                                    "serial sampling chain overlaps the next batch's encoder; the input-only work of batch k+1 (sampling pyramid, level-0 "
                                                    "self search) is issued right after batch k's encoder.  The last step of the timed region finishes its own batch too: "
                                                    "exactly `steps` complete forwards (pyramids, first parts, second parts) are enqueued and completed inside the timed "
                                                    "region; results are bit-identical to isolated forwards",
                                       "prefetch": "the furthest-point-sampling pyramid and the level-0 self search of step k+1 (input-only, side streams) are issued "
                                                   "right after step k's encoder is enqueued and run under step k's decoder; the first step of the timed region issues "
                                                   "its own and the last step prefetches nothing: exactly `steps` pyramids run inside the timed region"}[pipeline],
                   "arithmetic": "fp32 values throughout; the fused MLP layers form each fp32 product from six bf16 MFMA partial products of an "
                                 "exact 3-way operand split (fp32 accumulation, ~4 ulp from the f32-input MFMA build)"},
        # Random (untrained) weights: the network's frames are nowhere near the scan, so these two are NOT quality numbers -- they only
        # pin the metric code path (test.py:88-90).  Parity of the run itself is the `parity` object below.
        "chamfer_vs_gt_untrained_weights": chamfer,
        "emd_vs_gt_untrained_weights": emd,
    }
    if rank == 0 and args.config == "c2":
        result["quality"] = quality_on_scan_weights(dev)
    par = parity_vs_reference(local, rank) if cfg["golden"] else None
    if par is not None:
        result["parity"] = par

    pmc = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else {}
    entries = roofline_entries(timed, calls, inst_steps, pmc)
    for e in entries:
        e["timed_in"] = f"instrumented pass of {inst_steps} steps right after the timed region (same calls, one step in flight)"
    # Headline = the single kernel symbol with the most time on the step's critical (main) stream: the fusion kernel (HEADLINE;
    # profiles/r02_step_by_queue.txt).  The FPS chains run beside the main stream on a side stream and the KNN family is a dozen
    # launches of several kernel symbols (pruned / queue / small x K variants, none above 0.35 ms): both under roofline_others.
    alone = next((e for e in entries if e["kernel"] == NAMES[HEADLINE]), None)
    if alone is not None:
        # the headline kernel as measured INSIDE the timed region (beside whatever else is on the chip: the next step's FPS
        # chains); the same kernel alone on the chip (instrumented pass) is kept beside it
        live = (roofline_entries({k: (headline_timed if k == HEADLINE else (0, 0.0)) for k in FAMILIES}, calls, args.steps, pmc) or [dict(alone)])[0]
        live["timed_in"] = f"the timed region ({n_streams} step(s) in flight)"
        live["alone"] = {k: alone[k] for k in ("achieved", "frac", "avg_launch_us", "kernel_ms_per_step", "timed_in")}
        result["roofline"] = live
        result["roofline_others"] = sorted((e for e in entries if e is not alone), key=lambda e: -e["kernel_ms_per_step"])
    result["pointset_end_to_end"] = pointset_end_to_end(timed, calls, inst_steps)
    if args.config == "c4" and rank == 0:
        result["roofline_others"] = result.get("roofline_others", []) + c4_pointnet2_launches(x1, dev)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, batch=B_PER_GPU if args.config == "c2" else 2)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(n, argv):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <argv> as a child process; returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def cpu_baseline(cfg, batch):
    """The same graph on host cores: oracle backend ("port"), a bounded sample of the same workload (about 10-20 s): the full batch of
    config 2, two sequences of config 4."""
    from mocopci_amd import ops, synth
    from mocopci_amd.model import MoCoPCI
    from oracle.backend import OracleBackend

    # the GPU box gives one GPU a 16-core share; more threads than that only oversubscribe
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    net = MoCoPCI()
    net.load_state_dict(synth.weights_by_name(net._spec), strict=True)
    n = cfg["npoints"]
    x1, x2, _ = synth.make_batch(cfg["config_id"], batch, n, **cfg["kw"])
    from oracle import pointset as orc
    orc.lib().orc_set_threads(cores)
    prev = ops.set_backend(OracleBackend())
    try:
        t0 = time.perf_counter()
        net(x1, x2)
        dt = time.perf_counter() - t0
    finally:
        ops.set_backend(prev)
    return {"value": 3.0 * batch / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"one forward of the same workload over {batch} sequences (N={n}, {3 * batch} frames), {dt:.1f} s; "
                      "C oracle point-set ops (OpenMP) + torch-CPU dense ops"}


def _event_time(fn, reps):
    """Average duration of fn() in seconds, HIP events on the stream the launches go to (torch's current stream)."""
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def c4_pointnet2_launches(x1, dev, reps=20):
    """the standalone pointnet2 launches of the legacy PointNet++ layers (models/models.py:18-22) on the
    N=16384 clouds of this run -- ball_query at M=2048 centres, r in {0.5, 1, 2, 4}, nsample in {16, 16, 8, 8}, and three_nn
    (16384 unknown, 2048 known) -- through the reference-API functions (mocopci_amd.pointnet2_utils -> the C ABI), each with its
    compulsory bytes over its launch duration."""
    from mocopci_amd import pointnet2_utils as pu
    xyz = x1.transpose(1, 2).contiguous()                    # (B,16384,3)
    B, N, _ = xyz.shape
    centres = xyz[:, :2048].contiguous()
    out = []
    for r, ns in ((0.5, 16), (1.0, 16), (2.0, 8), (4.0, 8)):
        sec = _event_time(lambda: pu.ball_query(r, ns, xyz, centres), reps)
        byt = B * (12 * 2048 + 12 * N + 4 * 2048 * ns)
        out.append({"kernel": f"ball_query_kernel (mcp_ball_query) r={r} nsample={ns}, M=2048, N={N}", "bound": "hbm", "achieved": byt / sec / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byt / sec / 1e9 / HBM_PEAK_GBS, "traffic": None, "launches": reps,
                    "avg_launch_us": sec * 1e6, "note": "compulsory bytes B*(12M+12N+4*M*nsample) (SURVEY 8d) / launch duration; the kernel scans "
                    "LDS tiles of the cloud per query wave with early exit at nsample hits, so small radii scan most of the cloud: VALU-bound, not HBM-bound"})
    sec = _event_time(lambda: pu.three_nn(xyz, centres), reps)
    byt = B * (12 * N + 12 * 2048 + 24 * N)
    out.append({"kernel": f"three_nn (pointnet2_cuda.three_nn_wrapper: 2 x build_cloud_kernel + knn_pruned_kernel<4>, K=3) n={N} unknown, m=2048 known",
                "bound": "hbm", "achieved": byt / sec / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": byt / sec / 1e9 / HBM_PEAK_GBS, "traffic": None, "launches": reps, "avg_launch_us": sec * 1e6,
                "note": "compulsory bytes B*(12n+12m+24n) (SURVEY 8d) / duration of the whole call (the space-ordered copies of both clouds "
                "are built inside it); the exhaustive three_nn_kernel (n*m = 33.5 M distance evaluations per cloud, the path of smaller "
                "clouds) took 228 us at this shape (profiles/r04_c4_bench.json)"})
    return out


def run_c5(args):
    """BASELINE configs[4]: synthetic N=65536 dense scan, batch 8 -- the FPS + KNN kernels only (the reference formulation
    cannot hold a 16 GiB-per-element distance matrix, so the model is not run at this size).  A step = FPS 65536 -> 2048 of the
    eight clouds, the 32-NN of the 2048 sampled points in their cloud, and the 32-NN self search (Q = N = 65536)."""
    from mocopci_amd import ops
    from tests.golden_inputs import big_cloud      # the seeded cloud of the full-shape parity test (tests/test_ops_gpu.py:test_config5_full_shape_pins)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    be = ops.backend()
    B, N, M, K = 8, 65536, 2048, 32
    x = big_cloud(N, seed=5, batch=B).to(dev)
    state = {}

    main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)

    def step():
        # as inside a forward (schedule.py: the level-0 self search runs on a lane beside the sampling pyramid): the sampling is a serial
        # chain on 8 of the 256 CUs, the self search and the sorted form of the cloud depend on the input alone -- they run on a second
        # stream beside it; the search of the sampled points follows the sampling and reads the sorted cloud the other stream built
        # (cloud_scope orders the streams and the memory).  Everything of the step is complete when the step ends.
        with be.cloud_scope():           # one sorted form of the cloud serves both searches
            side.wait_stream(main)
            with torch.cuda.stream(side):
                state["knn_n"] = be.knn(x, x, K)
            sel, pts = be.fps(x, M, with_points=True)
            state["knn_q"] = be.knn(pts, x, K)
            main.wait_stream(side)
            state["knn_n"].record_stream(main)
        state["sel"] = sel
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-kernel durations, HIP events on the launch stream, same calls
    pts = be.group_rows(x, state["sel"])
    t_fps = _event_time(lambda: be.fps(x, M), 5)
    with be.cloud_scope():
        be.knn(x, x, K)                 # builds the sorted cloud inside the scope: the timed launches below are the searches alone
        t_build = _event_time(lambda: be._build_cloud(x), 5)
        t_kq = _event_time(lambda: be.knn(pts, x, K), 5)
        t_kn = _event_time(lambda: be.knn(x, x, K), 5)

    def entry(kernel, sec, byt, note, **extra):
        e = {"kernel": kernel, "bound": "hbm", "achieved": byt / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byt / sec / 1e9 / HBM_PEAK_GBS,
             "traffic": None, "launches": 5, "avg_launch_us": sec * 1e6, "note": note}
        e.update(extra)
        return e
    fps_stream = B * (M - 1) * 20 * N
    fps = entry("fps_tiled_kernel (mcp_furthest_point_sampling_fresh, caller workspace)", t_fps, fps_stream,
                "achieved = bytes of the REFERENCE'S streaming formulation B*(M-1)*20*N (SURVEY 8d: 21.5 GB) over the launch -- not a utilisation: "
                "the kernel keeps tile boxes in registers and touches only the tiles a new centre can affect",
                compulsory_GBs=B * (12 * N + 4 * M) / t_fps / 1e9, us_per_iteration=1e6 * t_fps / (M - 1))
    kq = entry("knn_walk_kernel K=32, Q=2048 sampled points, N=65536 (mcp_knn_pruned)", t_kq, B * (12 * M + 12 * N + 4 * M * K),
               "compulsory bytes B*(12Q+12N+4QK); reference formulation 24*B*Q*N = %.1f GB" % (24.0 * B * M * N / 1e9),
               reference_formulation_GBs=24.0 * B * M * N / t_kq / 1e9)
    kn = entry("knn_walk_kernel K=32, Q=N=65536 self search (mcp_knn_pruned)", t_kn, B * (12 * N + 12 * N + 4 * N * K),
               "compulsory bytes B*(12Q+12N+4QK); reference formulation 24*B*Q*N = %.0f GB (a 16 GiB distance matrix per cloud)" % (24.0 * B * N * N / 1e9),
               reference_formulation_GBs=24.0 * B * N * N / t_kn / 1e9)
    bc = entry("mcp_morton_codes + sort + mcp_tile_boxes (sorted form of the 65536-point clouds)", t_build, B * (12 * N * 2 + 4 * N),
               "bytes: cloud read, sorted cloud + permutation written; torch.sort of the Morton codes in between (N > 16384: outside the fused builder)")
    result = {"metric": "clouds/sec through FPS 65536->2048 + 32-NN (Q=2048 and Q=N) [BASELINE configs[4]: kernels only]", "value": B * args.steps / elapsed,
              "unit": "clouds/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": "synthetic N=65536 dense scan (x,y in +-40, z in +-3, 5 % duplicated points), batch=8, 1xMI355X: FPS 65536->2048 + "
                                     "32-NN Q=2048 + 32-NN Q=N per step (BASELINE configs[4])", "bench_config": "c5", "npoints": N, "batch_per_gpu": B,
                         "streams": "the self search (with the sorted form of the cloud) runs on a second stream beside the sampling chain, as the "
                                    "level-0 self search of a forward does; the step ends when both streams have"},
              "roofline": kn, "roofline_others": [fps, kq, bc]}
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_c5(x.cpu(), M, K)
    print(json.dumps(result))


def cpu_baseline_c5(x, M, K):
    """The oracle (C, OpenMP) on a bounded sample of the same step: FPS 65536 -> 2048 and the Q=2048 search of FOUR clouds, the Q=N self
    search of TWO clouds in full (about 10-15 s); clouds/s = 1 / (mean seconds per cloud of the three parts)."""
    from oracle import pointset as orc
    cores = min(len(os.sched_getaffinity(0)), 16)
    orc.lib().orc_set_threads(cores)
    four, two = x[:4].contiguous(), x[:2].contiguous()
    t0 = time.perf_counter()
    sel = orc.furthest_point_sample(four, M)
    t_fps = (time.perf_counter() - t0) / 4
    pts = torch.gather(four, 1, sel.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    t0 = time.perf_counter()
    orc.knn(pts, four, K)
    t_q = (time.perf_counter() - t0) / 4
    t0 = time.perf_counter()
    orc.knn(two, two, K)
    t_n = (time.perf_counter() - t0) / 2
    per_cloud = t_fps + t_q + t_n
    return {"value": 1.0 / per_cloud, "unit": "clouds/s", "cores": cores, "kind": "port",
            "sample": f"oracle FPS 65536->2048 of 4 clouds ({t_fps:.2f} s per cloud), exhaustive 32-NN of their 2048 sampled points ({t_q:.2f} s per cloud), "
                      f"exhaustive 32-NN self search Q=N=65536 of 2 clouds in full ({t_n:.2f} s per cloud)"}


if __name__ == "__main__":
    main()
