"""Which schedule node each stream wait blocks on, and for how long, in the REAL (untraced, pipelined) step.

Schedule.TRACE (mocopci_amd/schedule.py) makes every node record (start, done) on its lane and every get() an arrival event on the
reading stream; blocked = max(0, arrival -> done).  Printed per node, averaged over steady-state steps, with the node's own lane
time and how long after the step's first marker it started / finished -- the join VERDICT r4 #1(a) asks for.
    python tools/schedule_waits.py [net.FLAG=v ...] [--move=node:lane ...]"""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import schedule, synth
from mocopci_amd.model import MoCoPCI
from mocopci_amd.schedule import Schedule


def node_of(text):
    return eval(text) if text.startswith("(") else text


for a in sys.argv[1:]:
    if a.startswith("net."):
        k, v = a[4:].split("=")
        setattr(MoCoPCI, k, type(getattr(MoCoPCI, k))(int(v)))
    elif a.startswith("--move="):
        node, lane = a[7:].rsplit(":", 1)
        schedule.NODE_LANES[node_of(node)] = None if lane == "none" else int(lane)
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()


def run(n):
    h = net.prefetch(x1, x2, ev)
    pend = out = None
    for i in range(n):
        nxt = None if i == n - 1 else (x1, x2, ev)
        cur = net.begin(x1, x2, prefetched=h, then_prefetch=nxt)
        if pend is not None:
            out = net.finish(pend)
        pend = cur
        h = net.take_prefetched()
    if pend is not None:
        out = net.finish(pend)
    return out


run(6)
torch.cuda.synchronize()
steps = 14
Schedule.TRACE = []
net._marks = []
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
trace, marks = Schedule.TRACE, net._marks
Schedule.TRACE, net._marks = None, None
main_id = torch.cuda.current_stream().stream_id
starts = [m[1] for m in marks if m[0] == "enc start"]
# a get()'s step = the last "enc start" marker recorded before it (host order): walk both lists in record order is not possible
# (two lists), so place each record by its arrival time against the step starts
t_start = [starts[0].elapsed_time(s) for s in starts]


def step_of(t):
    k = 0
    while k + 1 < len(t_start) and t_start[k + 1] <= t:
        k += 1
    return k


blocked = collections.defaultdict(list)
runs = collections.defaultdict(list)
for rec in trace:
    if rec[0] == "get":
        _, node, reader, arrival, done = rec
        t_arr = starts[0].elapsed_time(arrival)
        k = step_of(t_arr)
        if 2 <= k < len(t_start) - 2:
            blocked[(node, "main" if reader == main_id else f"stream {reader}")].append((max(0.0, arrival.elapsed_time(done)), t_arr - t_start[k]))
    else:
        _, node, lane, start, done = rec
        t_s = starts[0].elapsed_time(start)
        k = step_of(t_s)
        if 2 <= k < len(t_start) - 2:
            runs[node].append((lane, start.elapsed_time(done), t_s - t_start[k], starts[0].elapsed_time(done) - t_start[k]))
print(f"{wall:.3f} ms/step with the trace events; steady-state steps used: {len(t_start) - 4}")
print("reader waits (averages per step; 'at' = when the reader arrived, ms after its step's first marker):")
rows = []
for (node, reader), v in blocked.items():
    n = len(v)
    rows.append((sum(b for b, _ in v) / (len(t_start) - 4), node, reader, n / (len(t_start) - 4), sum(a for _, a in v) / n))
tot = 0.0
for b, node, reader, per, at in sorted(rows, key=lambda r: -r[0]):
    if reader == "main":
        tot += b
    print(f"  {b * 1e3:8.1f} us blocked   {str(node):18s} read by {reader:12s} x{per:.1f}  at +{at:6.3f} ms")
print(f"  main stream blocked {tot:.3f} ms per step in all")
print("nodes (lane, time on the lane, start / end after the step's first marker):")
for node, v in sorted(runs.items(), key=lambda kv: kv[1][0][2]):
    n = len(v)
    print(f"  {str(node):18s} lane {v[0][0]}  {sum(r[1] for r in v) / n * 1e3:8.1f} us   +{sum(r[2] for r in v) / n:6.3f} -> +{sum(r[3] for r in v) / n:6.3f} ms")
