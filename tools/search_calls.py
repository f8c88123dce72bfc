import os, sys, torch
sys.path.insert(0, '/root/repo')
from mocopci_amd import ops, synth
from mocopci_amd.model import MoCoPCI
net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
net(x1, x2)
be = ops.backend()
log = []
orig_knn, orig_cos, orig_fps = be.knn, be.knn_cosine, be.fps
def knn(q, r, k, mode=0, return_dist=False):
    log.append(("knn", tuple(q.shape), tuple(r.shape), k, mode, q.data_ptr(), r.data_ptr())); return orig_knn(q, r, k, mode=mode, return_dist=return_dist)
def cos(a, b, k, return_dist=False):
    log.append(("cos", tuple(a.shape), tuple(b.shape), k, 0, a.data_ptr(), b.data_ptr())); return orig_cos(a, b, k, return_dist=return_dist)
def fps(x, m):
    log.append(("fps", tuple(x.shape), m, 0, 0, x.data_ptr(), 0)); return orig_fps(x, m)
be.knn, be.knn_cosine, be.fps = knn, cos, fps
net(x1, x2)
for l in log: print(l[:5])
