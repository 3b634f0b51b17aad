#!/usr/bin/env python3
"""Diagnostic (-DDC_L16_STAMPS build): per-stage time (us) of the 4 waves of one k_layer16 workgroup (layer 3), from s_memrealtime stamps.
GPU box:  DC_DDIM_LIB=.../libdc_ddim_L.alt DC_L16_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps16.py [bs]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import make_model, make_diffusion, xf_pair, batch_noise
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 1800
m = make_model("fp16")
xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T)).cuda()
nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
gd = make_diffusion(50)
for _ in range(2):
    nat.ddim_loop(noise, gd.native_coefficients())
torch.cuda.synchronize()
st = nat.debug_read("stamps", np.uint64, 8 * 32).reshape(8, 32).astype(np.int64)[:4]
names = ["load h, E, records; combine", "image landed + barrier", "LN + Q + softmax + attend (SA)", "closer", "stylize (SA)", "closer", "LN + Q + softmax + attend (CA)",
         "closer", "stylize (CA)", "closer", "FFN", "closer", "stylize (FFN)", "closer", "LN + K + keys", "value image + barrier", "factors + V + K^T V", "barrier",
         "sum + record"]
print(f"k_layer16, bs={B}, T={T}, workgroup 3 of layer 3 (us):   wave " + "".join(f"{w:8d}" for w in range(4)))
for k in range(1, 20):
    print(f"{k:2d} {names[k - 1]:32s}" + "".join(f"{(st[w, k] - st[w, k - 1]) / 100.0:8.2f}" for w in range(4)))
print(f"   {'total':32s}" + "".join(f"{(st[w, 19] - st[w, 0]) / 100.0:8.2f}" for w in range(4)))
if st[0, 20] > st[0, 0]:      # shared combine (round 5): stamp 20 sits between the slice's publication and the gather
    print(f"   {'prologue: loads + slice published':32s}" + "".join(f"{(st[w, 20] - st[w, 0]) / 100.0:8.2f}" for w in range(4)))
    print(f"   {'prologue: FiLM loads issued':32s}" + "".join(f"{(st[w, 1] - st[w, 20]) / 100.0:8.2f}" for w in range(4)))
    if st[0, 21] > st[0, 0]:
        print(f"   {'stage 1: LN + Q + softmax':32s}" + "".join(f"{(st[w, 21] - st[w, 2]) / 100.0:8.2f}" for w in range(4)))
        print(f"   {'stage 1: gather + barrier':32s}" + "".join(f"{(st[w, 22] - st[w, 21]) / 100.0:8.2f}" for w in range(4)))
        print(f"   {'stage 1: attend':32s}" + "".join(f"{(st[w, 3] - st[w, 22]) / 100.0:8.2f}" for w in range(4)))
