"""How much of a net.train() step is the batch-statistics fusion MLP: the step as is, and with that block replaced by the fused
(running-statistics) kernel pair -- timing only, the second variant is not the reference's training mode."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth, training, ops
from mocopci_amd.model import MoCoPCI
B, N = 8, 8192
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda(); net.train(True)
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
x1, x2, gt = synth.make_batch(2, B, N, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
def run(tag):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        training.train_step(net, opt, x1, x2, gtc)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{tag}: {dt * 1e3:.1f} ms  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
run("net.train() as is")
orig = MoCoPCI.fusion_batch_stats
def fused(self, p1, p2, idx, calls):
    m = "multi_frame_inference.conv."
    wb = [t for ci, bi in ((0, 1), (3, 4), (6, 7)) for t in self.folded_conv_bn(m + str(ci), m + str(bi), 1e-3)]
    return ops.backend().fusion_mlp(p1, p2, idx, *wb)
MoCoPCI.fusion_batch_stats = fused
run("fusion MLP on the fused kernels (timing only)")
