#!/usr/bin/env python3
"""GPU box: DDIM-50 of the golden clip (G5) stepped by hand - the first S - k model evaluations through the fp16 denoiser, the last k
through the split-operand one (`mixed`) - against the reference's x0: how much of the fp16 mode's error the last evaluations carry
(tests/study_operand_rounding.py found the weights' rounding in the final evaluations dominant).  usage: python tools/last_steps_study.py"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402
from helpers import batch_noise, golden, make_diffusion, make_model, rel_l2, xf_pair  # noqa: E402

g = golden("g5_ddim50_b1.npz")
xfp, xfo = xf_pair(1, 1800)
noise = torch.from_numpy(batch_noise(1, 1800)).cuda()
S = 50
gd = make_diffusion(S)
models = {p: make_model(p) for p in ("fp16", "mixed")}
mk = {"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor([1800])}
with torch.no_grad():
    for k in (0, 1, 2, 4, 8, 50):
        img = noise.clone()
        for it, i in enumerate(reversed(range(S))):
            m = models["mixed"] if it >= S - k else models["fp16"]
            t = torch.tensor([i], device="cuda")
            img = gd.ddim_sample(m, img, t, clip_denoised=False, model_kwargs=mk)["sample"]
        torch.cuda.synchronize()
        print(f"last {k:2d} of {S} evaluations split-operand, the others fp16: x0 rel-L2 vs the reference {rel_l2(img, g['x0']):.3e}")
