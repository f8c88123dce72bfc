"""A/B inside one process: training step time with the index-only lanes of a training forward (MoCoPCI.TRAIN_LANES) set to the
given tuples, alternating blocks of steps.  usage: python tools/train_lanes_ab.py [eval|train] [lanes,lanes ...]  e.g. 0,6 0,4,5,6"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth, training
from mocopci_amd.model import MoCoPCI
MODE = sys.argv[1] if len(sys.argv) > 1 else "eval"
variants = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(0, 6), (0, 4, 5, 6)]
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
net.train(MODE == "train")
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
x1, x2, gt = synth.make_batch(2, 8, 8192, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
for _ in range(3):
    training.train_step(net, opt, x1, x2, gtc)
res = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        MoCoPCI.TRAIN_LANES = v
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            training.train_step(net, opt, x1, x2, gtc)
            torch.cuda.synchronize()
            if it:
                res[v].append(time.perf_counter() - t0)
for v, ts in res.items():
    ts.sort()
    print(f"[{MODE}] TRAIN_LANES {v}: min {ts[0] * 1e3:.2f} ms  median {ts[len(ts) // 2] * 1e3:.2f} ms  ({len(ts)} steps)")
