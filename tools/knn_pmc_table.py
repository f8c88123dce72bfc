"""Per-dispatch SQ counters of the search kernels (rocprofv3 --pmc CSVs of tools/knn_pmc.sh) -> one table: for every launch of
knn_walk_kernel / knn_pruned_kernel, in dispatch order, the counters per wave (quad-cycle counters x 4 = cycles)."""
import collections, csv, sys
rows = collections.OrderedDict()
for path in sys.argv[1:]:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if "knn_walk_kernel" not in k and "knn_pruned_kernel" not in k:
                continue
            key = (path.split("/")[-3] if False else "", int(r["Dispatch_Id"]))
            d = rows.setdefault((k.split("<")[0].split("::")[-1] + "<" + k.split("<")[1].split(">")[0] + ">", int(r["Dispatch_Id"]), int(r["Grid_Size"])), {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
         "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE", "SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_LEVEL_WAVES", "SQ_WAVES", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"]
print("kernel, dispatch, waves | per wave: " + ", ".join(n.replace("SQ_", "") for n in names))
for (k, did, grid), d in rows.items():
    waves = grid / 64
    print(f"{k:32s} {did:5d} {int(waves):6d} | " + "  ".join(f"{d.get(n, float('nan')) / waves:9.1f}" for n in names))
