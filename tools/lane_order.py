"""ms per step of the serving loop (as tools/step_time.py) against the ORDER in which the model's six side lanes are created: the
runtime hands HIP streams to its 4 hardware queues in creation order, lanes that share a queue serialise, and a cross-stream wait
queued in one of them holds up the other.  usage: python tools/lane_order.py [--main=default|new] <order> [<order> ...]
   an order is a comma list of lane numbers 0..5, with 'x' = a dummy stream created at that point, e.g. 0,4,1,2,3,5 or x,0,1,2,3,4,5
--move=node:lane[,node:lane...]   also sweep the node -> lane table (mocopci_amd/schedule.py NODE_LANES; the model reads its copy
   MoCoPCI.NODE_LANES): e.g. --move=up43:2,wf:1 ; tuple nodes are written kind.level, e.g. mfa_proj.2:1 ; lane "-" = inline"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import synth
from mocopci_amd.model import MoCoPCI

orders = [a for a in sys.argv[1:] if not a.startswith("--")]
moves = {}
for a in sys.argv[1:]:
    if a.startswith("--move="):
        for item in a[len("--move="):].split(","):
            node, lane = item.split(":")
            node = (node.split(".")[0], int(node.split(".")[1])) if "." in node else node
            moves[node] = None if lane == "-" else int(lane)
main_new = "--main=new" in sys.argv
x1, x2, _ = synth.make_batch(2, 8, 8192, device="cuda")
ev = torch.cuda.Event(); ev.record()
dev = x1.device
keep = []
for order in orders:
    net = MoCoPCI(); net.load_state_dict(synth.weights_by_name(net._spec)); net = net.cuda()
    if moves:
        net.NODE_LANES = {**MoCoPCI.NODE_LANES, **moves}   # an instance attribute: this model's table only
    main = torch.cuda.Stream() if main_new else torch.cuda.current_stream()
    keep.append(main)
    with torch.cuda.stream(main):
        for tok in order.split(","):
            if tok == "x":
                keep.append(torch.cuda.Stream())
            elif tok != "-":
                net.side_stream(dev, int(tok))

        def run(n):
            h = net.prefetch(x1, x2, ev)
            pend = out = None
            for i in range(n):
                cur = net.begin(x1, x2, prefetched=h, then_prefetch=None if i == n - 1 else (x1, x2, ev))
                if pend is not None:
                    out = net.finish(pend)
                pend = cur
                h = net.take_prefetched()
            return net.finish(pend)
        run(5); torch.cuda.synchronize()
        res = []
        for rep in range(3):
            t0 = time.perf_counter(); run(30); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"main {'new' if main_new else 'default'} order {order:24s} ms/step " + " ".join(f"{r:.3f}" for r in res), flush=True)
    keep.append(net)   # its streams stay alive: the next model's lanes are created after them (the rotation continues)
