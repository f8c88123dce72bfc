"""Which torch ops one training step issues, per op and per call site in mocopci_amd (TorchDispatchMode over forward + loss + backward +
optimizer; autograd's own accumulation shows up as sites '?').  usage: python tools/train_op_sites.py [eval|train] [rows]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mocopci_amd import synth, training
from mocopci_amd.model import MoCoPCI

MODE = sys.argv[1] if len(sys.argv) > 1 else "eval"
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
net.train(MODE == "train")
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
x1, x2, gt = synth.make_batch(2, 8, 8192, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
for _ in range(2):
    training.train_step(net, opt, x1, x2, gtc)
ops_n, sites = collections.Counter(), collections.Counter()
VIEW = ("view", "reshape", "_unsafe_view", "t", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "as_strided",
        "split", "split_with_sizes", "unbind", "_reshape_alias", "is_same_size", "sym_size", "sym_stride", "sym_numel", "lift_fresh", "empty", "empty_like",
        "empty_strided", "new_empty", "record_stream", "unsafe_split", "chunk", "narrow", "flatten", "unflatten", "stride", "size", "numel", "dim", "is_contiguous")


def site():
    for fr in reversed(traceback.extract_stack()[:-3]):
        if "mocopci_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}:{fr.name}"
    return "?"


class Mode(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEW:
            ops_n[name] += 1
            sites[(name, site())] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    training.train_step(net, opt, x1, x2, gtc)
torch.cuda.synchronize()
print(f"[{MODE}] {sum(ops_n.values())} non-view torch ops in one training step")
print("by op:", ", ".join(f"{k} {v}" for k, v in ops_n.most_common(40)))
for (name, s), c in sites.most_common(ROWS):
    print(f"{c:5d} x {name:28s} {s}")
