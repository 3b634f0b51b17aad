#!/usr/bin/env python3
"""GPU box host: seconds of one oracle DDIM-10 (B=3 and B=1, T=1800) by torch thread count - which setting the GPU tests' oracle calls should use."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from helpers import xf_pair, batch_noise, oracle_params
import oracle.ddim_oracle as O
p = oracle_params()
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
for B in (3, 1):
    xfp, xfo = xf_pair(B, 1800); noise = torch.from_numpy(batch_noise(B, 1800))
    for n in (0, 8, 16, 32, 64):
        if n: torch.set_num_threads(n)
        with torch.no_grad():
            t0 = time.perf_counter(); O.ddim_sample_loop(p, noise, xfp, xfo, [1800] * B, 10); dt = time.perf_counter() - t0
        print(f"B={B} threads={n or 'default'}: {dt:.2f} s for DDIM-10", flush=True)
