"""Serving loop with the side lanes on CU-masked HIP streams (hipExtStreamCreateWithCUMask): a lane kernel with long-lived waves (the
EI cross-former's attention, the level-0 self search of the next batch) otherwise fills every CU and the main stream's small
encoder kernels wait for a slot.  Masked streams are created blocking by the runtime, so the main loop runs on a created stream here.
usage: python tools/cu_mask.py <config> ...   config = name:lanes:pattern, lanes e.g. 1234 (digits), pattern one of
   none | skip4 (every 4th CU off) | skip2 | top64 (CUs 192..255 off) | top128"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI

hip = ctypes.CDLL("libamdhip64.so")


def words(pattern):
    bits = [1] * 256
    if pattern == "skip4":
        bits = [0 if i % 4 == 3 else 1 for i in range(256)]
    elif pattern == "skip2":
        bits = [i % 2 for i in range(256)]
    elif pattern == "top64":
        bits = [1 if i < 192 else 0 for i in range(256)]
    elif pattern == "top128":
        bits = [1 if i < 128 else 0 for i in range(256)]
    elif pattern == "xcd6":   # if consecutive bits walk the XCDs: XCDs 6 and 7 off
        bits = [0 if i % 8 >= 6 else 1 for i in range(256)]
    return [sum(b << j for j, b in enumerate(bits[32 * w:32 * w + 32])) for w in range(8)]


def masked_stream(dev, pattern):
    w = words(pattern)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, (ctypes.c_uint32 * 8)(*w))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()
dev = x1.device
keep = []
for cfg in sys.argv[1:]:
    name, lanes, pattern = cfg.split(":")
    net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
    main = torch.cuda.current_stream() if name.startswith("d_") else torch.cuda.Stream()   # names starting with d_: the default stream
    keep += [main, net]
    with torch.cuda.stream(main):
        sides = net.__dict__.setdefault("_sides", {})
        for which in range(6):
            key = (dev.index, main.stream_id, which)
            sides[key] = masked_stream(dev, pattern) if str(which) in lanes and pattern != "none" else torch.cuda.Stream(device=dev)

        def run(n):
            h = net.prefetch(x1, x2, ev)
            pend = out = None
            for i in range(n):
                cur = net.begin(x1, x2, prefetched=h, then_prefetch=None if i == n - 1 else (x1, x2, ev))
                if pend is not None:
                    out = net.finish(pend)
                pend = cur
                h = net.take_prefetched()
            return net.finish(pend)
        out = run(5); torch.cuda.synchronize()
        res = []
        for rep in range(3):
            t0 = time.perf_counter(); out = run(30); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"{name:14s} lanes {lanes:6s} {pattern:7s} ms/step " + " ".join(f"{r:.3f}" for r in res) + "  checksum %.6f" % float(out[0].double().sum()), flush=True)
