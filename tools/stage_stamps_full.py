#!/usr/bin/env python3
"""Diagnostic: where one k_layer_full workgroup (layer 3, workgroup 3) spends its time, from s_memrealtime stamps (100 MHz).
Needs a -DDC_FULL_STAMPS build:  tools/ab.sh build ST -DDC_FULL_STAMPS ; on the GPU box:
DC_DDIM_LIB=$PWD/diffusion-conductor_amd/libdc_ddim_ST.alt DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps_full.py"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import make_model, make_diffusion, xf_pair, batch_noise
B, T = 32, 1800
m = make_model("fp16", no_eff=True)
xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T)).cuda()
nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
gd = make_diffusion(50)
for _ in range(1):
    nat.ddim_loop(noise, gd.native_coefficients())
torch.cuda.synchronize()
st = nat.debug_read("stamps", np.uint64, 8 * 32).reshape(8, 32).astype(np.int64)
rows = [("load h, LN, query projection (SA)", 0, 8), ("barrier + key loop (SA)", 8, 9), ("normalise, LN stats", 9, 1),
        ("load h + stylize (SA)", 1, 2), ("store h, LN, query projection (CA)", 2, 10), ("barrier + key loop (CA)", 10, 11),
        ("normalise, LN stats", 11, 3), ("load h + stylize (CA)", 3, 4), ("FFN", 4, 5), ("stylize (FFN)", 5, 6),
        ("store h, LN, K / V projection of the next layer", 6, 7)]
print("wave:".ljust(50) + "".join(f"{w:8d}" for w in range(8)))
for name, a, b in rows:
    print(name.ljust(50) + "".join(f"{(st[w, b] - st[w, a]) / 100.0:8.2f}" for w in range(8)))
print("total (us)".ljust(50) + "".join(f"{(st[w, 7] - st[w, 0]) / 100.0:8.2f}" for w in range(8)))
