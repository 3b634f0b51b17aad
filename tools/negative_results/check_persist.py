#!/usr/bin/env python3
"""GPU box: the persistent layer launch (all 8 layers in one kernel) must reproduce the per-layer launches bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_model, make_diffusion, xf_pair, batch_noise
for B, T, S in ((32, 1800, 50), (20, 1024, 25), (9, 1800, 25), (40, 1600, 25)):
    m = make_model("fp16")
    xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T)).cuda()
    length = [T if b % 3 else max(1, T - 37 * b) for b in range(B)]
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), length)
    gd = make_diffusion(S)
    os.environ["DC_PERSIST"] = "1"
    a, _ = nat.ddim_loop(noise, gd.native_coefficients())
    st_a = nat.status()
    del os.environ["DC_PERSIST"]
    b, _ = nat.ddim_loop(noise, gd.native_coefficients())
    os.environ["DC_PERSIST"] = "1"
    c, _ = nat.ddim_loop(noise, gd.native_coefficients())
    del os.environ["DC_PERSIST"]
    torch.cuda.synchronize()
    print(f"B={B} T={T} S={S}: persistent == per-layer: {torch.equal(a, b)}  re-run identical: {torch.equal(a, c)}  finite {bool(torch.isfinite(a).all())}  status {st_a}",
          flush=True)
