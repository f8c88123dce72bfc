import torch, time
dev = "cuda"
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); r = fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e))
    return sorted(v)[len(v) // 2]
for (B, T, N) in ((24, 524288, 8192), (48, 65536, 2048), (16, 262144, 8192), (48, 16384, 512)):
    idx = torch.randint(0, N, (B, T), device=dev, dtype=torch.int32)
    a = t(lambda: torch.sort(idx, dim=1, stable=True))
    def flat():
        pos = torch.arange(T, device=dev, dtype=torch.int64)
        keys = (idx.long() + torch.arange(B, device=dev, dtype=torch.int64).view(B, 1) * N) * T + pos
        s = torch.sort(keys.reshape(-1))[0]
        return (s % T).int().view(B, T), s // T
    b = t(flat)
    k2, o2 = torch.sort(idx, dim=1, stable=True)
    o3, _ = flat()
    print(f"B={B} T={T} N={N}: batched stable int32 sort {a:.3f} ms, flat int64 sort {b:.3f} ms, same order: {torch.equal(o2.int(), o3)}", flush=True)
