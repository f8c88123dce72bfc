"""Does torch's TunableOp (runtime GEMM solution selection) speed the library GEMMs of one step up?  ms per step before / after."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI
dev = torch.device("cuda", 0)
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec), strict=True); net = net.to(dev)
x1, x2, _ = synth.make_batch(2, 8, 8192, device=dev)
ev = torch.cuda.Event(); ev.record()
def timed(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): net(x1, x2, inputs_ready=ev)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): net(x1, x2, inputs_ready=ev)
print("default: %.3f ms/step" % timed(), flush=True)
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(20); tn.set_max_tuning_iterations(20)
tn.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "tunableop.csv"))
t0 = time.perf_counter()
for _ in range(2): net(x1, x2, inputs_ready=ev)
torch.cuda.synchronize()
print("tuning pass: %.1f s" % (time.perf_counter() - t0), flush=True)
tn.tuning_enable(False)
for _ in range(2): net(x1, x2, inputs_ready=ev)
print("tuned: %.3f ms/step" % timed(), flush=True)
print("tuned again: %.3f ms/step" % timed(), flush=True)
