"""Where the independent chains of one forward run: a table (node -> lane) and the ONE executor that consumes it.

model.py states the math: `sched.run(node, fn)` names a chain of launches that depends only on what exists at that point, and
`sched.get(node)` is where its result is first read.  Which HIP stream a node runs on, the waits in front of it, the event behind
it and the allocator bookkeeping (record_stream) are decided HERE, from NODE_LANES -- data that tools/lane_order.py sweeps -- so a
kernel fusion in model.py no longer means re-tuning stream calls by hand.

Lanes are extra HIP streams beside the caller's (one set per caller stream and device).  The runtime deals streams onto its four
hardware queues in creation order, and what shares a queue serialises, so the assignment matters as much as the overlap itself
(the level-3 chain on lane 3 gives 7.18 ms per step, on lane 1 or 2 -- busy with the encoder's EI branches at
that point -- 7.49)."""
import torch

# node -> lane.  A node that is absent (or mapped to None) runs inline on the caller's stream.
#   0  the encoder's furthest-point-sampling pyramid: four latency-bound launches, one workgroup per cloud
#   1-3  chains that read only encoder outputs of pyramid level 1 / 2 / 3 (EI cross-formers, feature-cosine searches, swapped copies),
#        and, once those are done, the decoder's feature-only chains of the same level
#   4  the level-0 self search (input-only: issued with the pyramid)
#   5  the refinement stage's sampling -- NOT lane 0: the next batch's pyramid may already be queued there
#   6  training forwards only: the ground truth's sampling (train.py:131 -> mocopci.py:1081-1083)
NODE_LANES = {
    "xyz": 0, ("pc", 1): 0, ("pc", 2): 0, ("pc", 3): 0, ("pc", 4): 0, "swap_pc": 0,   # input layout, the four FPS levels, the swapped clouds
    "self_search": 4,
    "gt_down": 6,                               # training forwards: the ground truth's sampling pyramid (no gradient), beside the encoder
    ("swap_f", 1): 1, ("fus", 1): 1, ("cos", 1): 1, "i3_01": 1,
    ("swap_f", 2): 2, ("fus", 2): 2, ("cos", 2): 2,
    "rep0": 1,                                  # level 0's stacked inputs: four copies beside the level-0 interpolation search
    "up43": 3, ("cos", 3): 3, ("t22", 3): 3,     # level 3: upsampled level-4 features, cross3's cosine search, one of its two projections
    ("mfa_proj", 2): 2, ("mfa_proj", 1): 1,      # the feature-only chains in front of Multiframe_Attention, levels 2 / 1
    "wf": 2,                                     # rlevel0 features of the refinement stage, beside the warped clouds' self search
    "refine_fps": 5,
    "i3_refine": 1, "knn_down": 2,               # refinement tail: 3-NN search of the upsampling, the Point-Transformer's 16-NN search
}
# MoCoPCI.SIDE_PROJECTIONS = False (A/B switch) runs these inline
SIDE_PROJECTION_NODES = {"rep0", "up43", ("cos", 3), ("t22", 3), ("mfa_proj", 2), ("mfa_proj", 1), "wf", "knn_down"}


def _tensors(res):
    if isinstance(res, torch.Tensor):
        return (res,)
    if isinstance(res, (tuple, list)):
        return tuple(t for r in res for t in _tensors(r))
    if isinstance(res, dict):
        return tuple(t for r in res.values() for t in _tensors(r))
    return ()


class Schedule:
    """Executor of one forward.  run(node, fn): fn() on the node's lane behind everything enqueued on the current stream so far
    (inline without a lane: CPU backends, training forwards, nodes not in the table); get(node): the reading stream -- whichever is
    current -- waits for that result and is recorded as a user of its memory; provide(node, value, event): a result produced
    elsewhere (the prefetched pyramid)."""

    # Debug mode (tools/schedule_waits.py): a list here makes every node record timing events on its lane (start, done) and every
    # get() an arrival event on the reading stream, appended as ("run", node, lane, start, done) / ("get", node, reader stream id,
    # arrival, done): after a synchronize, max(0, arrival -> done) is how long that reader was blocked by that node, in the real
    # (untraced, pipelined) step.  None (the default) records nothing and creates no timing events.
    TRACE = None

    def __init__(self, model, device):
        self.model, self.device = model, device
        self.items = {}

    def lane(self, node):
        table = self.model.NODE_LANES
        which = table.get(node)
        if which is None or (not self.model.SIDE_PROJECTIONS and node in SIDE_PROJECTION_NODES):
            return None
        return self.model.side_stream(self.device, which)

    def run(self, node, fn, reads=(), after=None):
        """reads: tensors fn reads that were allocated on ANOTHER side lane (results of other nodes): recorded as used on this one.
        after: the events the node depends on, INSTEAD of everything enqueued on the current stream so far (input-only work issued
        ahead of the caller's stream: MoCoPCI.prefetch); () = nothing but the lane's own order."""
        aux = self.lane(node)
        if aux is None:
            self.items[node] = (fn(), None)
            return
        if after is None:
            aux.wait_stream(torch.cuda.current_stream(self.device))  # inputs were produced on (or fetched by) the launching stream
        else:
            for ev in after:
                if ev is not None:
                    aux.wait_event(ev)
        for t in reads:
            t.record_stream(aux)
        trace = Schedule.TRACE
        with torch.cuda.stream(aux):
            if trace is not None:
                start = torch.cuda.Event(enable_timing=True)
                start.record(aux)
            res = fn()
            ev = torch.cuda.Event(enable_timing=trace is not None)
            ev.record(aux)
        if trace is not None:
            trace.append(("run", node, self.model.NODE_LANES.get(node), start, ev))
        self.items[node] = (res, ev)

    def provide(self, node, value, event=None):
        self.items[node] = (value, event)

    def has(self, node):
        return node in self.items

    def peek(self, node):
        """The value without any wait: for a reader on the SAME lane (stream order), or one that passes event(node) as `after`."""
        return self.items[node][0]

    def event(self, node):
        return self.items[node][1]

    def get(self, node):
        res, ev = self.items[node]
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            if Schedule.TRACE is not None:
                arrival = torch.cuda.Event(enable_timing=True)
                arrival.record(cur)
                Schedule.TRACE.append(("get", node, cur.stream_id, arrival, ev))
            cur.wait_event(ev)
            for t in _tensors(res):
                t.record_stream(cur)
        return res
