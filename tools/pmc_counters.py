"""Per-kernel-family hardware counters from rocprofv3 --pmc passes -> one JSON (committed under profiles/).

    python tools/pmc_counters.py <out.json> <counter_collection.csv> [more.csv ...]

Each CSV is one pass (gpurun refuses --pmc together with tracing, and FETCH_SIZE / WRITE_SIZE do not fit one pass:
MI355X_MICROARCH.md, "rocprofv3 PMC slots").  tools/profile_round.sh runs the passes:
    1  FETCH_SIZE                                  2  WRITE_SIZE
    3  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
       SQ_WAIT_ANY SQ_WAIT_INST_ANY  + GRBM_GUI_ACTIVE
Units and corrections (same guide, HBM section and the SQ-units row of its constants table):
  * FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B for wide coalesced reads, so the read
    side is doubled; narrower gathers are uncalibrated, which makes hbm_bytes an upper estimate of the read side;
  * SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
    summed over SIMDs; durations come from the dispatch timestamps of the same rows.
Derived per family (formulas in the code below, spelled out so nobody has to trust a name): cycles = GRBM_GUI_ACTIVE / 8 (the
counter sums the 8 XCDs); valu_busy_frac_of_chip = 4 * SQ_ACTIVE_INST_VALU / (cycles * 1024 SIMDs)  (VALUBusy, gfx94x formula);
mfma_busy_frac_of_chip = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024)  (MfmaUtil); wave_time_* = share of resident-wave time.
"""
import collections
import csv
import json
import sys

FAMILIES = {  # name -> kernel-name substrings
    "fps": ("fps_spatial_kernel", "fps_resident_kernel", "fps_stream_kernel", "fps_tiled_kernel"),
    "knn": ("knn_walk_kernel", "knn_pruned_kernel", "knn_queue_kernel", "knn_small_kernel"),
    "knn_pruned": ("knn_walk_kernel", "knn_pruned_kernel"),
    "build_cloud": ("build_cloud_kernel",),
    "knn_cosine": ("knn_cosine_kernel", "knn_cosine_split_kernel"),
    "fusion": ("fusion_kernel", "fusion_split_kernel"),
    "cross": ("cross_kernel", "cross256_stream_kernel"),
    "pointconv": ("pointconv_agg_kernel", "pointconv_agg_lowlevel_kernel", "pointconv_linear_kernel"),
    "attention": ("attention_small_kernel", "attention_wide_kernel", "attention_kernel"),
    "attention_small": ("attention_small_kernel",),
    "attention_wide": ("attention_wide_kernel",),
    "ptblock": ("ptblock_kernel",),
    "mlp": ("mlp2_kernel",),
    "linear": ("linear_kernel",),
    "group_rows": ("group_rows_kernel",),
    "ball_query": ("ball_query_kernel", "query_and_group_kernel"),
    "three_nn": ("three_nn_kernel",),
    "fps_tiled": ("fps_tiled_kernel",),
    "knn_walk": ("knn_walk_kernel",),
    "interp3": ("interp3_",),
    "dense": ("dense_",),
    "gemm_library": ("Cijk_",),
}
N_SIMD = 256 * 4      # MI355X: 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4       # nominal shader clock; only used to turn dispatch durations into cycles


def family_of(kernel):
    return [f for f, subs in FAMILIES.items() if any(s in kernel for s in subs)]


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    # family -> counter -> [sum, launches];  family -> [duration_ns sum, launches] (from the SQ pass if present)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for path in paths:
        seen = set()
        with open(path) as fh:
            for row in csv.DictReader(fh):
                fams = family_of(row["Kernel_Name"])
                if not fams:
                    continue
                for f in fams:
                    a = acc[f][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    key = (f, row["Dispatch_Id"])
                    if key not in seen:
                        seen.add(key)
                        dur[f][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        dur[f][1] += 1
    out = {"_units": __doc__.split("Units and corrections")[1].strip()}
    for f in FAMILIES:
        if f not in acc:
            continue
        c = {k: v[0] / v[1] for k, v in acc[f].items()}
        e = {"launches_sampled": max(v[1] for v in acc[f].values()), "avg_dispatch_us_under_pmc": dur[f][0] / max(dur[f][1], 1) / 1e3,
             "counters_per_launch": c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            e["hbm_read_bytes_per_launch"], e["hbm_write_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        if "SQ_WAVE_CYCLES" in c:
            wave = c["SQ_WAVE_CYCLES"]  # quad-cycles summed over waves
            for name, key in (("valu", "SQ_ACTIVE_INST_VALU"), ("lds", "SQ_ACTIVE_INST_LDS"), ("any", "SQ_ACTIVE_INST_ANY")):
                if key in c:
                    e[f"wave_time_issuing_{name}_frac"] = c[key] / wave       # share of resident-wave time spent issuing that class
            for name, key in (("parked", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY")):
                if key in c:
                    e[f"wave_time_{name}_frac"] = c[key] / wave
            # kernel duration in shader cycles: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so /8 is the busy-cycle count at the
            # clock the kernel actually ran at (MFMA-heavy kernels run near 2.0 GHz, not the nominal 2.4); nominal otherwise
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in c else e["avg_dispatch_us_under_pmc"] * 1e3 * CLOCK_GHZ
            # GRBM_GUI_ACTIVE / 8 / wall time reads high on short dispatches (MI355X_MICROARCH.md, DVFS give-back: the quotient is
            # only meaningful from about 0.3 ms up -- round 3 published 5.4-5.9 "GHz" for 5 us kernels): not reported below that
            if "GRBM_GUI_ACTIVE" in c and e["avg_dispatch_us_under_pmc"] >= 300.0:
                e["effective_clock_ghz"] = cycles / (e["avg_dispatch_us_under_pmc"] * 1e3)
            e["mean_resident_waves_per_simd"] = 4.0 * wave / (cycles * N_SIMD)
            if "SQ_ACTIVE_INST_VALU" in c:
                e["valu_busy_frac_of_chip"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (cycles * N_SIMD)   # VALUBusy, gfx94x formula
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_frac_of_chip"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * N_SIMD)   # MfmaUtil, gfx94x formula
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in c and "valu_busy_frac_of_chip" in e and "mfma_busy_frac_of_chip" in e:
                # share of the SIMD-cycles in which the matrix pipe OR the VALU works: on this chip the two do not overlap, within or
                # across waves (tools/peak/overlap2_probe.hip), so for a fused MFMA + VALU kernel THIS is the utilisation of its bound
                e["mfma_valu_coexec_frac_of_chip"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / (cycles * N_SIMD)
                e["simd_issue_busy_frac_of_chip"] = e["valu_busy_frac_of_chip"] + e["mfma_busy_frac_of_chip"] - e["mfma_valu_coexec_frac_of_chip"]
        out[f] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for f, e in out.items():
        if f.startswith("_"):
            continue
        print(f"{f:14s} n={e['launches_sampled']:4d} {e['avg_dispatch_us_under_pmc']:9.1f} us"
              + (f"  hbm {e['hbm_bytes_per_launch'] / 1e6:8.2f} MB" if "hbm_bytes_per_launch" in e else "")
              + (f"  valu {e['valu_busy_frac_of_chip']:.3f}" if "valu_busy_frac_of_chip" in e else "")
              + (f"  mfma {e['mfma_busy_frac_of_chip']:.3f}" if "mfma_busy_frac_of_chip" in e else "")
              + (f"  either {e['simd_issue_busy_frac_of_chip']:.3f}" if "simd_issue_busy_frac_of_chip" in e else ""))


if __name__ == "__main__":
    main()
