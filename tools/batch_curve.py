#!/usr/bin/env python3
"""GPU box: the whole batch-size curve of the DDIM-50 loop at T = 1800 - ms per loop and frames/s for bs = 1 .. 32 with the launch form
the library picks (k_layer16 while every 64-token unit has a CU: bs <= 8; the 32-token narrow form while every 128-token unit has one:
bs <= 17; 8-wave workgroups above), and for the batch sizes near the two switch points the neighbouring form forced through its
environment switch, same box, alternating - so that the dispatch rule can be checked against what wins.
usage: python tools/batch_curve.py [lo hi]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402
from helpers import batch_noise, make_diffusion, make_model, xf_pair  # noqa: E402

T, S = 1800, 50
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 32)
m = make_model("fp16")
coef = make_diffusion(S).native_coefficients()


def loop_ms(nat, noise, n=5):
    for _ in range(2):
        nat.ddim_loop(noise, coef)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(n):
            nat.ddim_loop(noise, coef)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


print("| bs | form | ms per DDIM-50 loop | frames/s | other form (forced) |")
print("|---|---|---|---|---|")
for B in range(lo, hi + 1):
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T)).cuda()
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
    form = "k_layer16 (16-token waves)" if B <= 8 else ("narrow (4 x 32 tokens)" if B <= 17 else "wide (8 x 32 tokens)")
    t = loop_ms(nat, noise)
    other = ""
    alt = {"DC_NO_LAYER16": "narrow"} if 5 <= B <= 8 else ({"DC_NO_NARROW": "wide"} if 9 <= B <= 17 else None)
    if alt:
        (env, name), = alt.items()
        os.environ[env] = "1"
        try:
            t2 = loop_ms(nat, noise)
        finally:
            del os.environ[env]
        t = min(t, loop_ms(nat, noise))
        other = f"{name}: {t2:.2f} ms ({100 * (t2 / t - 1):+.1f} %)"
    print(f"| {B} | {form} | {t:.2f} | {B * T / t * 1e3:,.0f} | {other} |", flush=True)
