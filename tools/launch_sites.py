"""Which line of mocopci_amd/model.py each device kernel of one inference step comes from: kernels per source line with their
summed device time (torch.profiler with Python stacks).  Library calls made through ctypes carry no torch op, so the step is
run with a recording wrapper around ops._call that opens a record_function per library call."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from mocopci_amd import ops, synth
from mocopci_amd.model import MoCoPCI

dev = torch.device("cuda", 0)
net = MoCoPCI()
net.load_state_dict(synth.weights_by_name(net._spec), strict=True)
net = net.to(dev)
x1, x2, _ = synth.make_batch(2, 8, 8192, device=dev)
for _ in range(3):
    net(x1, x2)
torch.cuda.synchronize()

MODEL = os.path.join("mocopci_amd", "model.py")


HELPERS = {"lin", "conv1d_block", "layer_norm", "bn_eval", "<lambda>", "<genexpr>", "cross_attention", "pointconv", "mlp_t", "fps_gather"}


def site():
    """innermost model.py frame, plus the nearest enclosing frame that is not one of the small helpers"""
    inner = None
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.endswith(MODEL):
            if inner is None:
                inner = f"{fr.name}:{fr.lineno}"
            if fr.name not in HELPERS:
                return f"{fr.name}:{fr.lineno} > {inner}"
    return inner or "?"


# every torch op and library call is attributed to the innermost model.py frame on the Python stack at enqueue time
orig_call = ops._call
def traced_call(name, ref, *a):
    with record_function(f"SITE|{site()}|{name}"):
        return orig_call(name, ref, *a)
ops._call = traced_call

class Mode(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        with record_function(f"SITE|{site()}|{func.__name__}"):
            return func(*args, **(kwargs or {}))

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with Mode():
        net(x1, x2)
    torch.cuda.synchronize()
ops._call = orig_call

events = prof.events()
# device kernels carry the correlation of the runtime launch; walk up the CPU parents to the nearest SITE record
by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
sites = [e for e in events if e.name.startswith("SITE|")]
sites.sort(key=lambda e: e.time_range.start)
kernels = 0
for e in events:
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    # innermost SITE record whose CPU interval contains this launch
    best = None
    for s in sites:
        if s.thread == e.thread and s.time_range.start <= e.time_range.start and e.time_range.end <= s.time_range.end:
            if best is None or s.time_range.start >= best.time_range.start:
                best = s
    key = best.name.split("|")[1] if best else "?"
    for k in e.kernels:
        by_site[key][0] += 1
        by_site[key][1] += k.duration
        by_site[key][2][k.name.split("(")[0].split("<")[0][-40:]] += 1
        kernels += 1
print(f"{kernels} kernels in one step")
for key, (n, us, names) in sorted(by_site.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:4d} kernels {us:8.1f} us  {key}   " + ", ".join(f"{c}x {nm}" for nm, c in names.most_common(4)))
