#!/usr/bin/env python3
"""GPU box: the bf16 precision's precise tail - golden DDIM-50 (B=1, T=1800) rel-L2 of x0 against the reference fixture and the loop time
at bs=32 x 1800 for tails of 0 / 1 / 2 / 4 / 8 evaluations, with the tail's FiLM GEMM on f16 operands (the "mixed" evaluation form,
default since round 6) and on bf16 operands (DC_TAIL_FILM_BF16=1, round 5's form).   usage: python tools/bf16_tail_sweep.py"""
import os
import sys
import time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402
from helpers import batch_noise, golden, make_diffusion, make_model, rel_l2, xf_pair  # noqa: E402

S = 50
gd = make_diffusion(S)
coef = gd.native_coefficients()
g = golden("g5_ddim50_b1.npz")
xfp1, xfo1 = (t.cuda() for t in xf_pair(1, 1800))
nz1 = torch.from_numpy(batch_noise(1, 1800)).cuda()
xfp, xfo = (t.cuda() for t in xf_pair(32, 1800))
nz = torch.from_numpy(batch_noise(32, 1800)).cuda()
m = make_model("bf16")
for film in ("f16", "bf16"):
    if film == "bf16":
        os.environ["DC_TAIL_FILM_BF16"] = "1"
    for tail in (0, 1, 2, 4, 8, 50):
        n1 = m.set_conditioning(xfp1, xfo1, [1800])
        n1.set_precise_tail(tail)
        o1, _ = n1.ddim_loop(nz1, coef)
        err = rel_l2(o1, g["x0"])
        nb = m.set_conditioning(xfp, xfo, [1800] * 32)
        nb.set_precise_tail(tail)
        for _ in range(2):
            nb.ddim_loop(nz, coef)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ob, _ = nb.ddim_loop(nz, coef)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"bf16 precision, tail FiLM GEMM on {film:4s} operands, tail {tail:2d}: golden DDIM-50 rel-L2 {err:.3e}   bs=32 loop {ms:.2f} ms   status {nb.status()}", flush=True)
    os.environ.pop("DC_TAIL_FILM_BF16", None)
