"""Per-kernel SQ counters from rocprofv3 --pmc CSVs, averaged over a kernel's launches and divided by its wave count:
usage: python tools/pmc_kernel_table.py <name substrings, comma separated> <counter_collection.csv ...>
Quad-cycle counters (WAVE_CYCLES, BUSY_CYCLES, WAIT_*, ACTIVE_INST_*) are in units of 4 cycles."""
import collections, csv, sys
pats = sys.argv[1].split(",")
acc = collections.OrderedDict()
for path in sys.argv[2:]:
    seen = collections.defaultdict(set)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if not any(p in k for p in pats):
                continue
            name = k.split("(anonymous namespace)::")[-1].split("(")[0]
            d = acc.setdefault(name, {"waves": 0.0, "n": collections.defaultdict(int), "v": collections.defaultdict(float)})
            c = r["Counter_Name"]
            d["v"][c] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen[(name, c)]:
                seen[(name, c)].add(r["Dispatch_Id"])
                d["n"][c] += 1
                d["w_" + c] = d.get("w_" + c, 0.0) + float(r["Grid_Size"]) / 64
for name, d in acc.items():
    print(name)
    for c in sorted(d["v"]):
        print(f"    {c:32s} launches {d['n'][c]:4d}   per launch {d['v'][c] / d['n'][c]:14.1f}   per wave {d['v'][c] / d['w_' + c]:12.1f}")
