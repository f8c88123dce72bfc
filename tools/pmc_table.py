"""Per-kernel averages of the counters in rocprofv3 --pmc CSVs: python tools/pmc_table.py <substring of the kernel name> <csv ...>"""
import collections, csv, sys
pat, paths = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in paths:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if pat not in k:
                continue
            name = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            depth = 0
            for i, ch in enumerate(name):   # cut the argument list: the first "(" outside the template brackets
                depth += ch == "<"
                depth -= ch == ">"
                if ch == "(" and depth == 0:
                    name = name[:i]
                    break
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for name, cs in acc.items():
    print(name)
    wave = cs.get("SQ_WAVE_CYCLES", [0, 1]); wv = wave[0] / max(wave[1], 1)
    for c, (tot, n) in sorted(cs.items()):
        v = tot / n
        print(f"   {c:28s} {v:16.0f}" + (f"   = {v / wv:.3f} of SQ_WAVE_CYCLES" if wv and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""))
