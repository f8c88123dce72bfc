"""What the bf16 matrix pipe sustains for accumulator chains of v_mfma_f32_32x32x16_bf16 (the stream shape of the split-bf16 layers),
by resident waves per SIMD, independent chains per wave and LDS operand re-reads.  Build first: tools/peak/build.sh."""
import ctypes, os, statistics, sys
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "build", "libmfma_peak.so"))
lib.mfma_peak.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = torch.zeros(4, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ITERS = 2000
F32 = "--f32" in sys.argv        # v_mfma_f32_32x32x2_f32 instead of v_mfma_f32_32x32x16_bf16
RANDOM = "--random" in sys.argv  # random operand bits instead of all-ones: same instruction stream, realistic switching power
for wps in (1, 2, 4):
    for chains in (1, 2):
        for lds_a in (1,):
            ts = []
            for rep in range(4):
                s, e = torch.cuda.Event(True), torch.cuda.Event(True)
                s.record()
                rc = lib.mfma_peak(chains, lds_a, wps, -ITERS if RANDOM else ITERS, out.data_ptr(), stream, int(F32))
                e.record()
                torch.cuda.synchronize()
                assert rc == 0, rc
                ts.append(s.elapsed_time(e))
            ms = statistics.median(ts[1:])
            flop = 256 * wps * 4 * ITERS * 24 * chains * 2 * 32 * 32 * (2 if F32 else 16)
            peak = 157.3 if F32 else 2500.0
            print(f"{'random' if RANDOM else 'ones  '} waves/SIMD {wps} chains {chains} lds_a {lds_a}: {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TFLOP/s {'f32 ' if F32 else 'bf16'} "
                  f"({flop / ms / 1e9 / peak:.2f} of {peak:g} TF)", flush=True)
