#!/usr/bin/env python3
"""GPU box: randomized (B, mel frames, activation format) cases of MusicEncoder + proj (`encode_music`, transformer.py:447-459) against the
oracle - frame counts off every tile edge of the LDS-tiled convolutions and the stride-3 pool, from the 4 frames the reference's reflection padding needs.
Test infrastructure: the oracle is the checker.  usage: python tools/fuzz_encoder.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import O, batch_mel, make_model, oracle_params, rel_l2  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {"split": 1e-4, "f16": 6e-4}
m = make_model("fp16")
p = oracle_params()
bad, worst, t0 = 0, {}, time.perf_counter()
for case in range(N):
    kind = rng.integers(0, 3)
    Tm = int(rng.integers(4, 40)) if kind == 0 else int(rng.integers(40, 700)) if kind == 1 else int(rng.integers(700, 5401))
    B = int(rng.integers(1, 12 if Tm < 700 else 3))
    fmt = str(rng.choice(["split", "f16"]))
    mel = torch.from_numpy(batch_mel(B, Tm, first=int(rng.integers(0, 100))))
    with torch.no_grad():
        rp, rx = O.encode_music(p, mel)
    os.environ["DC_ME_PREC"] = fmt
    try:
        xp, x = m.encode_music(mel.cuda(), "cuda:0")
        torch.cuda.synchronize()
    finally:
        del os.environ["DC_ME_PREC"]
    e = max(rel_l2(xp, rp), rel_l2(x, rx))
    ok = bool(torch.isfinite(x).all()) and tuple(x.shape) == tuple(rx.shape) and e <= TOL[fmt]
    bad += not ok
    worst[fmt] = max(worst.get(fmt, 0.0), e)
    print(f"case {case:3d} B={B:2d} Tm={Tm:4d} -> T={x.shape[1]:4d} {fmt:5s}: {e:.3e}{'' if ok else '   <-- FAIL'}", flush=True)
print(f"{N} cases, {bad} failures, {time.perf_counter() - t0:.0f} s; worst per format: " + ", ".join(f"{k} {v:.3e}" for k, v in sorted(worst.items())))
sys.exit(1 if bad else 0)
