#!/usr/bin/env python3
"""GPU box: randomized (B, T, length, precision, steps) cases of the whole DDIM loop against the oracle - shapes off every grid the
fixed tests name (T from 1 frame up, batches on both sides of every dispatch switch, ragged lengths down to 1).  Test infrastructure:
the oracle is the checker here, never the thing measured.  usage: python tools/fuzz_shapes.py [cases] [seed] [precision: every case in it]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import O, batch_noise, make_diffusion, make_model, oracle_params, rel_l2, xf_pair  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))      # (the oracle's small GEMMs: more threads than that only contend)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {"fp16": 1e-3, "bf16": 1e-3, "mixed": 1e-3, "bf16x3": 1e-4}
models = {}
worst = {}
bad = 0
t0 = time.perf_counter()
for case in range(N):
    kind = rng.integers(0, 4)
    if kind == 0:        # short clips, many of them
        T, B = int(rng.integers(1, 300)), int(rng.integers(1, 40))
    elif kind == 1:      # around the 256-frame switch of the record forms and the 32-frame groups
        T, B = int(rng.choice([255, 256, 257, 288, 319, 320, 321, 511, 512, 513])), int(rng.integers(1, 20))
    elif kind == 2:      # long clips, small batches (16-token waves / narrow units)
        T, B = int(rng.integers(600, 1801)), int(rng.integers(1, 5))
    else:
        T, B = int(rng.integers(300, 1000)), int(rng.integers(1, 12))
    while B * T > 5000 and B > 1:
        B = max(1, B // 2)
    no_eff = rng.random() < 0.2 and T >= 32
    if no_eff:
        T = min(T, 700)
    length = [int(rng.integers(1, T + 1)) if rng.random() < 0.6 else T for _ in range(B)]
    prec = str(rng.choice(["fp16"] if no_eff else ["fp16", "fp16", "bf16", "mixed", "bf16x3"]))
    if len(sys.argv) > 3:
        prec = sys.argv[3]
        no_eff = no_eff and prec == "fp16"
    S = int(rng.choice([21, 25, 50] if B * T <= 2500 else [21, 25]))
    first = int(rng.integers(0, 200))
    mk = (prec, no_eff)
    if mk not in models:
        models[mk] = make_model(prec, no_eff=no_eff)
    xfp, xfo = xf_pair(B, T, first=first)
    noise = torch.from_numpy(batch_noise(B, T, first=first))
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S, no_eff=no_eff)
    gd = make_diffusion(S)
    out = gd.ddim_sample_loop(models[mk], (B, T, 26), noise=noise.cuda(), clip_denoised=False, progress=False,
                              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
    torch.cuda.synchronize()
    errs = [rel_l2(out[c:c + 1], ref[c:c + 1]) for c in range(B)]
    e = max(errs)
    ok = bool(torch.isfinite(out).all()) and e <= TOL[prec]
    bad += not ok
    worst[prec + ("/no_eff" if no_eff else "")] = max(worst.get(prec + ("/no_eff" if no_eff else ""), 0.0), e)
    print(f"case {case:3d} B={B:2d} T={T:4d} S={S:2d} {prec + ('/no_eff' if no_eff else ''):11s} min length {min(length):4d}: worst clip {e:.3e}{'' if ok else '   <-- FAIL'}", flush=True)
print(f"{N} cases, {bad} failures, {time.perf_counter() - t0:.0f} s; worst clip per precision: " + ", ".join(f"{k} {v:.3e}" for k, v in sorted(worst.items())))
sys.exit(1 if bad else 0)
