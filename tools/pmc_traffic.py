"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output).
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters
are in KiB; FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams, so the read side is doubled."""
import collections, csv, json, sys

GROUPS = {  # roofline name -> kernel-name substrings
    "fps": ("fps_spatial_kernel", "fps_resident_kernel", "fps_stream_kernel"),
    "knn": ("knn_pruned_kernel", "knn_queue_kernel", "knn_small_kernel"),
    "fusion": ("fusion_kernel",),
    "knn_pruned_only": ("knn_pruned_kernel",),
    "cross": ("cross_kernel",),
    "attention": ("attention_small_kernel",),
}


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for name, subs in GROUPS.items():
    fk = [(k, v) for k, v in fetch.items() if any(s in k for s in subs)]
    wk = [(k, v) for k, v in write.items() if any(s in k for s in subs)]
    nf, nw = sum(v[1] for _, v in fk), sum(v[1] for _, v in wk)
    if not nf or not nw:
        continue
    f_kb, w_kb = sum(v[0] for _, v in fk) / nf, sum(v[0] for _, v in wk) / nw
    out[name] = {
        "kernels": sorted({k[:80] for k, _ in fk}),
        "launches_sampled": nf,
        "FETCH_SIZE_KB_per_launch": f_kb,
        "WRITE_SIZE_KB_per_launch": w_kb,
        "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> read side doubled "
                      "(MI355X_MICROARCH.md, HBM); narrower gathers are uncalibrated, so this is an upper estimate of the read side",
    }
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:16s} launches {v['launches_sampled']:4d}  fetch {v['FETCH_SIZE_KB_per_launch']:10.1f} KiB  write {v['WRITE_SIZE_KB_per_launch']:10.1f} KiB"
          f"  -> {v['hbm_bytes_per_launch']/1e6:8.2f} MB/launch")
