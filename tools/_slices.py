import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mocopci_amd import synth, training
from mocopci_amd.model import MoCoPCI
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda(); net.train(False)
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
x1, x2, gt = synth.make_batch(2, 8, 8192, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
training.train_step(net, opt, x1, x2, gtc)
sites = collections.Counter()
def site():
    out = []
    for fr in reversed(traceback.extract_stack()[:-3]):
        if "mocopci_amd" in fr.filename:
            out.append(f"{os.path.basename(fr.filename)}:{fr.lineno}:{fr.name}")
            if len(out) == 2: break
    return " < ".join(out) or "?"
class Mode(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in ("slice", "select", "split", "split_with_sizes", "unbind", "narrow", "index", "chunk"):
            t = args[0]
            if isinstance(t, torch.Tensor) and t.requires_grad:
                sites[(name, site(), tuple(t.shape))] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    frames_f, frames_b, gt_frame, out_lst = net(x1, x2, gtc, None, True)
    loss, parts = training.multiscale_loss(frames_f, frames_b, gt_frame, out_lst, gtc)
for (n, s, sh), c in sites.most_common(60):
    print(f"{c:4d} x {n:8s} {str(sh):22s} {s}")
