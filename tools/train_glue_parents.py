"""Which autograd nodes / torch ops the fill, copy and add launches of one training step come from: for every such device kernel, the
chain of CPU ops around its launch (torch.profiler), counted by the outermost two ops.  usage: python tools/train_glue_parents.py [eval|train]"""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from mocopci_amd import synth, training
from mocopci_amd.model import MoCoPCI

MODE = sys.argv[1] if len(sys.argv) > 1 else "eval"
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda(); net.train(MODE == "train")
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
x1, x2, gt = synth.make_batch(2, 8, 8192, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
for _ in range(2):
    training.train_step(net, opt, x1, x2, gtc)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    training.train_step(net, opt, x1, x2, gtc)
    torch.cuda.synchronize()
WATCH = {"fill": ("FillFunctor", "fillBuffer", "Memset"), "copy": ("copyBuffer", "Memcpy", "direct_copy"), "add": ("CUDAFunctor_add",)}
counts = {k: collections.Counter() for k in WATCH}
for e in prof.events():
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    for k in e.kernels:
        for group, pats in WATCH.items():
            if any(p in k.name for p in pats):
                chain, p = [], e
                while p is not None:
                    chain.append(p.name)
                    p = p.cpu_parent
                chain = [c for c in chain if not c.startswith("hip") and not c.startswith("cuda")]
                counts[group][" < ".join(chain[-3:][::-1]) if chain else "?"] += 1
for group, c in counts.items():
    print(f"== {group}: {sum(c.values())} launches")
    for name, n in c.most_common(14):
        print(f"   {n:4d}  {name[:150]}")
