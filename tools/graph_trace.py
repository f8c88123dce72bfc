"""Per-step composition of a kernel trace by hardware queue, for every window between two fusion launches: used to compare eager
steps with HIP-graph replays of the same step (tools/graph_try.py under rocprofv3 --kernel-trace).
usage: python tools/graph_trace.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
fus = [i for i, r in enumerate(rows) if "fusion_split_kernel" in r[0]]
for a, b in zip(fus, fus[1:]):
    step = rows[a + 1:b + 1]
    wall = (step[-1][2] - step[0][1]) / 1e6
    qs = {}
    for n, s, e, q in step:
        qs.setdefault(q, [0, 0.0])
        qs[q][0] += 1
        qs[q][1] += (e - s) / 1e6
    main = max(qs, key=lambda q: qs[q][1])
    mk = sorted([r for r in step if r[3] == main], key=lambda r: r[1])
    gaps = [mk[i + 1][1] - mk[i][2] for i in range(len(mk) - 1)]
    small = sum(g for g in gaps if 0 < g <= 15000) / 1e6
    big = sum(g for g in gaps if g > 15000) / 1e6
    print(f"wall {wall:7.3f} ms  kernels {len(step):4d}  queues " + "  ".join(f"q{q}:{c}/{t:.2f}ms" for q, (c, t) in sorted(qs.items())) +
          f"  | busiest q{main}: gaps<=15us {small:.2f} ms, larger gaps {big:.2f} ms")
