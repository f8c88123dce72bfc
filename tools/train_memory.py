"""Where a net.train() training step's memory goes (B=8, N=8192): allocated bytes at the section markers of the forward
(MoCoPCI._mark), at the end of the forward, and the peak of the backward.  usage: python tools/train_memory.py [train|eval]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth, training
from mocopci_amd.model import MoCoPCI
MODE = sys.argv[1] if len(sys.argv) > 1 else "train"
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
net.train(MODE == "train")
x1, x2, gt = synth.make_batch(2, 8, 8192, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
G = 2 ** 30
log = []
orig = MoCoPCI._mark
def mark(self, name):
    torch.cuda.synchronize()
    log.append((name, torch.cuda.memory_allocated() / G, torch.cuda.max_memory_allocated() / G))
    torch.cuda.reset_peak_memory_stats()
    return orig(self, name)
MoCoPCI._mark = mark
for it in range(2):
    log.clear(); torch.cuda.reset_peak_memory_stats()
    frames_f, frames_b, gt_frame, out_lst = net(x1, x2, gtc, None, True)
    loss, parts = training.multiscale_loss(frames_f, frames_b, gt_frame, out_lst, gtc)
    torch.cuda.synchronize()
    fwd_live, fwd_peak = torch.cuda.memory_allocated() / G, torch.cuda.max_memory_allocated() / G
    torch.cuda.reset_peak_memory_stats()
    net.zero_grad(); loss.backward(); torch.cuda.synchronize()
    bwd_peak = torch.cuda.max_memory_allocated() / G
print(f"[{MODE}] allocated GiB at each forward marker (live, peak since the previous marker):")
for name, live, peak in log:
    print(f"  {live:7.2f} {peak:7.2f}  {name}")
print(f"  end of forward + loss: live {fwd_live:.2f}, peak since the last marker {fwd_peak:.2f}; backward peak {bwd_peak:.2f} GiB")
