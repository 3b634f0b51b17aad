#!/usr/bin/env python3
"""GPU box: the DC_FILM4=1 experiment kernel (256-accumulator FiLM GEMM tile) must reproduce the default GEMM bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_model, make_diffusion, xf_pair, batch_noise
for B, T in ((3, 300), (32, 1800)):
    m = make_model("fp16")
    xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T)).cuda()
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
    gd = make_diffusion(50)
    os.environ["DC_NO_FUSE_EMBED"] = "1"
    a, _ = nat.ddim_loop(noise, gd.native_coefficients())
    os.environ["DC_FILM4"] = "1"
    b, _ = nat.ddim_loop(noise, gd.native_coefficients())
    del os.environ["DC_FILM4"]
    torch.cuda.synchronize()
    print(f"B={B} T={T}: film4 == film3: {torch.equal(a, b)}  finite {bool(torch.isfinite(b).all())}")
    assert torch.equal(a, b)
