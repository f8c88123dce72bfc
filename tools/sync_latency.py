"""Cost of one dependent exchange on MI355X: across workgroups through L2 vs inside a workgroup through LDS (tools/peak/sync_latency.hip;
build with tools/peak/build.sh).  Why furthest point sampling stays one workgroup per cloud: its M-1 iterations are dependent, and a
cloud split over several workgroups pays one cross-workgroup exchange per iteration."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "build", "libsync_latency.so"))
lib.sync_latency.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
stream = torch.cuda.current_stream().cuda_stream
names = {0: "two workgroups, ping-pong through L2 (agent-scope atomics), per ROUND TRIP", 1: "two waves of one workgroup, ping-pong through LDS, per round trip",
         2: "dependent global atomic add with return (one L2 atomic round trip)", 3: "1024-thread workgroup: LDS atomic max + barrier + LDS read (FPS's exchange today)"}
for mode in (0, 1, 2, 3):
    iters = 20000
    flags = torch.zeros(8, dtype=torch.int32, device="cuda")
    cyc = torch.zeros(2, dtype=torch.int64, device="cuda")
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    for rep in range(2):
        flags.zero_()
        torch.cuda.synchronize()
        s.record()
        rc = lib.sync_latency(mode, iters, flags.data_ptr(), cyc.data_ptr(), stream)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{names[mode]}: {us:.3f} us", flush=True)
