"""oracle/make_hashes.py -- full-size regression pins: SHA-256 of index tensors the C ORACLE
produces on seeded clouds at BASELINE sizes, so the -m gpu tests can check bit-exactness at N=8192/16384/65536
without re-running the (slow) oracle on the GPU box.  These pin the kernels to the oracle, not to the reference.

    python -m oracle.make_hashes
"""
import hashlib
import json
import os

import torch

from oracle import pointset as orc
from tests.golden_inputs import big_cloud

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hashes.json")


def h(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def main():
    out = {}
    for n, m in ((8192, 2048), (16384, 2048), (65536, 512)):
        x = big_cloud(n)
        out[f"fps_{n}_{m}"] = h(orc.furthest_point_sample(x, m))
    for n, s in ((8192, 2048), (16384, 2048), (65536, 2048)):
        u, k = big_cloud(n), big_cloud(s, seed=2)
        d, i = orc.three_nn(u, k)
        out[f"three_nn_{n}_{s}"] = h(i)
    x = big_cloud(8192)
    out["knn32_8192"] = h(orc.knn(x, x, 32))
    x = big_cloud(16384)
    out["knn16_direct_16384"] = h(orc.knn(x, x, 16, mode=1))
    out["ball_query_16384_r1_16"] = h(orc.ball_query(1.0, 16, x, x[:, :2048].contiguous()))
    # BASELINE configs[4] at its full shape (synthetic N=65536 dense scan, batch 8): FPS 65536 -> 2048 for all 8 clouds, and the
    # K=32 self search for a 4096-query slice of cloud 0 (the GPU test runs the whole Q=65536 search and compares the slice)
    x = big_cloud(65536, seed=5, batch=8)
    out["fps_c5_8x65536_2048"] = h(orc.furthest_point_sample(x, 2048))
    out["knn32_c5_65536_rows_30000_34096"] = h(orc.knn(x[:1, 30000:34096].contiguous(), x[:1], 32))
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
