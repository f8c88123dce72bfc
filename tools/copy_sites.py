"""Which lines of mocopci_amd/model.py / ops.py issue the torch ops that become device-to-device copies (hipMemcpyAsync: aten.copy_ /
clone / contiguous between dense same-dtype tensors) or fills in one inference step; counts per call site."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI

dev = torch.device("cuda", 0)
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec), strict=True); net = net.to(dev)
x1, x2, _ = synth.make_batch(2, 8, 8192, device=dev)
for _ in range(3):
    net(x1, x2)
torch.cuda.synchronize()
WATCH = ("copy_", "clone", "contiguous", "_to_copy", "fill_", "zero_", "zeros", "zeros_like", "empty_like", "cat", "stack", "repeat", "expand_copy", "index_select", "add", "add_", "mul", "sub")
sites = collections.Counter()


def site():
    out = []
    for fr in reversed(traceback.extract_stack()[:-3]):
        if "mocopci_amd" in fr.filename:
            out.append(f"{os.path.basename(fr.filename)}:{fr.lineno}:{fr.name}")
            if len(out) == 2:
                break
    return " < ".join(out) or "?"


class Mode(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            if t is None and args and isinstance(args[0], (list, tuple)):
                t = args[0][0]
            shape = tuple(t.shape) if t is not None else ()
            sites[(name, site(), shape)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    net(x1, x2)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, s, shape), c in sorted(sites.items(), key=lambda kv: (kv[0][0], -kv[1])):
    tot[name] += c
    print(f"{c:3d} x {name:12s} {str(shape):24s} {s}")
print(dict(tot))
