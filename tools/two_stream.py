"""Independent batches on several caller streams (one forward per stream at a time, round-robin) with the model's side lanes folded
onto fewer HIP streams, so that the streams in use do not outnumber the runtime's hardware queues (4 by default) -- streams that
share a hardware queue serialise, and a cross-stream wait in one of them holds up the other.
usage: python tools/two_stream.py [--callers N] [--lanes a,b,c,d,e,f | none] [--steps N] [--prefetch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI

arg = lambda k, d: next((a.split("=")[1] for a in sys.argv[1:] if a.startswith(k + "=")), d)
callers, steps = int(arg("--callers", "2")), int(arg("--steps", "40"))
lanes = arg("--lanes", "0,0,0,0,0,0")
MoCoPCI.LANE_MAP = None if lanes == "none" else tuple(int(v) for v in lanes.split(","))
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()
S = [torch.cuda.Stream() for _ in range(callers)]
for s_ in S:
    s_.wait_stream(torch.cuda.current_stream())
two_batch = "--two-batch" in sys.argv
pend = [None] * callers
hand = [None] * callers


def run(n):
    out = None
    for i in range(n):
        c = i % callers
        last = i >= n - callers
        with torch.cuda.stream(S[c]):
            if two_batch:  # each caller stream runs the bench's own two-batches-in-flight loop
                h = hand[c] if hand[c] is not None else net.prefetch(x1, x2, ev)
                cur = net.begin(x1, x2, prefetched=h, then_prefetch=None if last else (x1, x2, ev))
                if pend[c] is not None:
                    out = net.finish(pend[c])
                pend[c] = cur
                if last:
                    out = net.finish(cur)
                    pend[c] = None
                hand[c] = net.take_prefetched()
            else:
                out = net(x1, x2, inputs_ready=ev)
    return out


run(2 * callers + 2); torch.cuda.synchronize()
res = []
for rep in range(3):
    t0 = time.perf_counter(); out = run(steps); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / steps * 1e3)
print(f"callers {callers} lanes {lanes} {'two-batch' if two_batch else 'plain'}: ms/step " + " ".join(f"{r:.3f}" for r in res), " checksum %.6f" % float(out[0].double().sum()), flush=True)
