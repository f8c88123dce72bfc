"""Summarise a rocprofv3 rocpd database (rocprofv3 --kernel-trace): per-kernel stats CSV, and for ONE timed step (the kernels
between two consecutive fusion_kernel launches, the step's last kernel) the time per kernel family on every HIP queue, so the
queue that bounds the step -- the main stream -- can be read off.
usage: python tools/rocpd_stats.py <results.db> [out.csv] [step_out.txt]"""
import collections
import csv
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    if n.startswith("Cijk"):
        return "GEMM(hipBLASLt)"
    return re.sub(r"[<(].*", "", n)[:44]


def main():
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
    fus = [i for i, r in enumerate(rows) if "fusion_kernel" in r[0] or "fusion_split_kernel" in r[0]]
    # one steady-state step = the kernels between two consecutive fusion launches; the serving loop's pipeline fill / drain makes
    # some windows short (two tails back to back), so take the window with the median kernel count
    wins = sorted(range(len(fus) - 1), key=lambda i: fus[i + 1] - fus[i])
    i = wins[len(wins) // 2]
    a, b = fus[i], fus[i + 1]
    step = rows[a + 1:b + 1]
    lines = []
    wall = (step[-1][2] - step[0][1]) / 1e6
    lines.append(f"one step: wall {wall:.3f} ms, {len(step)} kernels")
    queues = collections.defaultdict(list)
    for r in step:
        queues[r[3]].append(r)
    for q, rs in sorted(queues.items(), key=lambda kv: -sum(e - s for _, s, e, _ in kv[1])):
        busy = sum(e - s for _, s, e, _ in rs) / 1e6
        lines.append(f"queue {q}: {len(rs)} kernels, busy {busy:.3f} ms")
        per = collections.defaultdict(lambda: [0, 0])
        for n, s, e, _ in rs:
            per[short(n)][0] += e - s
            per[short(n)][1] += 1
        for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:40]:
            lines.append(f"    {t / 1e6:7.3f} ms  {c:4d}x  {k}")
    # where the main queue (the busiest one) sits idle inside the step: gaps above 15 us with the kernels on both sides
    main_q = max(queues, key=lambda q: sum(e - s for _, s, e, _ in queues[q]))
    rs = sorted(queues[main_q], key=lambda r: r[1])
    gaps = [(rs[i + 1][1] - rs[i][2], short(rs[i][0]), short(rs[i + 1][0]), (rs[i][2] - step[0][1]) / 1e6) for i in range(len(rs) - 1)]
    small = sum(g for g, *_ in gaps if 0 < g <= 15000) / 1e6
    lines.append(f"main queue {main_q}: idle {sum(max(g, 0) for g, *_ in gaps) / 1e6:.3f} ms; {small:.3f} ms of it in gaps <= 15 us ({sum(1 for g, *_ in gaps if 0 < g <= 15000)} boundaries)")
    for g, a, b, at in sorted(gaps, key=lambda t: -t[0])[:14]:
        if g > 15000:
            lines.append(f"    gap {g / 1e3:7.1f} us at +{at:6.3f} ms  after {a}  before {b}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")


if __name__ == "__main__":
    main()
