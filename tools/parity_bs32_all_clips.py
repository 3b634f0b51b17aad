#!/usr/bin/env python3
"""GPU box: EVERY clip of the headline batch (bs=32 x 1800, DDIM-50, flat 256-token units in fp16; clip-aligned units in mixed) against the CPU
oracle's run of that clip alone - the per-clip rel-L2 of x0 (tests/test_gpu_parity.py::test_bs32_interior_clips_vs_oracle gates clips 13, 17, 31).
~2 min of CPU for the 32 oracle loops.   python tools/parity_bs32_all_clips.py > gpurun_out/r04_parity_bs32_all_clips.txt"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import O, batch_noise, make_diffusion, make_model, oracle_params, rel_l2, xf_pair
B, T, S = 32, 1800, 50
xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T))
torch.set_num_threads(min(16, os.cpu_count() or 8))
p = oracle_params()
cache = os.environ.get("DC_REFS_CACHE")        # (several libraries compared in one session: the 32 oracle loops run once)
if cache and os.path.exists(cache):
    refs = list(torch.load(cache))
else:
    with torch.no_grad():
        refs = [O.ddim_sample_loop(p, noise[c:c + 1], xfp[c:c + 1], xfo[c:c + 1], [T], S) for c in range(B)]
    if cache:
        torch.save(refs, cache)
gd = make_diffusion(S)
for mode in os.environ.get("DC_PARITY_MODES", "fp16,mixed").split(","):
    m = make_model(mode)
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
    out, _ = nat.ddim_loop(noise.cuda(), gd.native_coefficients())
    torch.cuda.synchronize()
    errs = [rel_l2(out[c:c + 1], refs[c]) for c in range(B)]
    print(f"{mode}: per-clip rel-L2 vs the oracle, clips 0..31: " + " ".join(f"{e:.2e}" for e in errs))
    print(f"{mode}: min {min(errs):.3e}  median {float(np.median(errs)):.3e}  max {max(errs):.3e} (clip {int(np.argmax(errs))});  whole batch {rel_l2(out, torch.cat(refs)):.3e};  status {nat.status()}")
    del nat, m
