"""The main-stream kernels of one step in launch order with duration, grid and workgroup size (rocprofv3 --kernel-trace database):
long launches with few waves are the ones that leave the chip idle on the critical path.
usage: python tools/step_timeline.py <results.db> [--all]     (--all: every queue, with the queue id in front)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = [c for c in cols if "grid" in c.lower()]
wx = [c for c in cols if "workgroup" in c.lower()]
sel = ",".join(["name", "start", "end", "queue_id"] + gx[:3] + wx[:3])
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
fus = [i for i, r in enumerate(rows) if "fusion_split_kernel" in r[0] or "fusion_kernel" in r[0]]
wins = sorted(range(len(fus) - 1), key=lambda i: fus[i + 1] - fus[i])   # median-sized window: a steady-state step
i0 = wins[len(wins) // 2]
step = rows[fus[i0] + 1:fus[i0 + 1] + 1]
busy = {}
for r in step:
    busy[r[3]] = busy.get(r[3], 0) + r[2] - r[1]
main_q = max(busy, key=busy.get)
t0 = step[0][1]
for r in step:
    if r[3] != main_q and "--all" not in sys.argv:
        continue
    name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r[0])
    name = "GEMM(hipBLASLt)" if name.startswith("Cijk") else re.sub(r"\(.*", "", name)[:60]
    g = [int(v) for v in r[4:4 + len(gx[:3])]]
    w = [int(v) for v in r[4 + len(gx[:3]):]]
    threads = 1
    for v in g:
        threads *= max(v, 1)
    waves = threads // 64
    flag = " <-- under 2 waves/SIMD" if (r[2] - r[1]) > 15000 and waves < 2048 else ""
    print(("" if "--all" not in sys.argv else f"q{r[3]} ") + f"+{(r[1] - t0) / 1e6:7.3f} ms {(r[2] - r[1]) / 1e3:8.1f} us  waves {waves:7d}  wg {w}  {name}{flag}")
