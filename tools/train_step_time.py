"""One training step (forward(train=True) + train.py's multi-scale Chamfer objective + backward + clipped Adam step) at the
BASELINE configs[1] shape, timed on the GPU; prints peak memory.  usage: python tools/train_step_time.py [batch] [npoints] [eval|train] [steps]
"train" (default): after net.train(), as train.py:130 -- batch-statistics BatchNorm and dropout; "eval": the inference graph differentiated."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth, training
from mocopci_amd.model import MoCoPCI
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
MODE = sys.argv[3] if len(sys.argv) > 3 else "train"
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
net.train(MODE == "train")
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
x1, x2, gt = synth.make_batch(2, B, N, device="cuda")
gtc = [g.transpose(1, 2).contiguous() for g in gt]
losses = []
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
times = []
for it in range(STEPS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, parts = training.train_step(net, opt, x1, x2, gtc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    losses.append(loss)
    times.append(dt)
    print(f"[{MODE}] step {it}: loss {loss:.4f}  {dt * 1e3:.1f} ms  peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
print(f"[{MODE}] steady steps: min {min(times[1:]) * 1e3:.1f} ms  median {sorted(times[1:])[len(times[1:]) // 2] * 1e3:.1f} ms")
print("loss decreased" if losses[-1] < losses[0] else "loss did not decrease", losses)
