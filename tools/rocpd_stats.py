"""Summarise a rocprofv3 rocpd database: per-kernel stats CSV + per-step ranking of the timed region.
usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import collections, csv, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
agg = collections.defaultdict(list)
for n, s, e, q in rows:
    agg[n].append(e - s)
tot = sum(sum(v) for v in agg.values())
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 1), round(100 * sum(v) / tot, 3), min(v), max(v)])
# one timed step = kernels between two consecutive fusion_kernel launches (last one of the run)
fus = [i for i, r in enumerate(rows) if "fusion_kernel" in r[0]]
a, b = fus[-3], fus[-2]
step = rows[a + 1:b + 1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    m = re.match(r"Cijk.*?(MT\d+x\d+x\d+)", n)
    return "GEMM(hipBLASLt)" if m else re.sub(r"[<(].*", "", n)[:44]
per = collections.defaultdict(lambda: [0, 0])
for n, s, e, q in step:
    per[short(n)][0] += e - s
    per[short(n)][1] += 1
wall = (step[-1][2] - step[0][1]) / 1e6
main_q = collections.Counter(r[3] for r in step).most_common(1)[0][0]
busy = sum(e - s for n, s, e, q in step if q == main_q) / 1e6
print(f"one step: wall {wall:.3f} ms, {len(step)} kernels, main-queue busy {busy:.3f} ms")
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:32]:
    print(f"  {t/1e6:7.3f} ms  {c:4d}x  {k}")
