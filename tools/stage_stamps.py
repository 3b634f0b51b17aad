#!/usr/bin/env python3
"""Diagnostic: per-stage time (us) of the 8 waves of one k_layer workgroup (layer 3), from s_memrealtime stamps.
Run on the GPU box:  DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_model, make_diffusion, xf_pair, batch_noise
B, T = 32, 1800
m = make_model("fp16")
xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T)).cuda()
nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
gd = make_diffusion(50)
for _ in range(2):
    nat.ddim_loop(noise, gd.native_coefficients())
torch.cuda.synchronize()
st = nat.debug_read("stamps", np.uint64, 8 * 32).reshape(8, 32).astype(np.int64)
names = ["load h + image 0", "Q + attend (SA)", "barrier", "stylize (SA)", "barrier", "Q + attend (CA)", "barrier + stylize (CA)", "barrier", "FFN", "barrier + stylize (FFN)", "barrier", "LN + K proj + barrier", "store h + V proj + partial records"]
t0 = st[:, 0].min()
print("wave:      " + "".join(f"{w:8d}" for w in range(8)))
for k in range(1, 14):
    print(f"{k:2d} {names[k-1][:16]:16s}" + "".join(f"{(st[w, k] - st[w, k-1]) / 100.0:8.2f}" for w in range(8)))
print("total (us) " + "".join(f"{(st[w, 13] - st[w, 0]) / 100.0:8.2f}" for w in range(8)))
print("prologue split (us): issue->combine done " + "".join(f"{(st[w, 14] - st[w, 0]) / 100.0:7.2f}" for w in range(8)))
print("                      ->own loads landed  " + "".join(f"{(st[w, 15] - st[w, 14]) / 100.0:7.2f}" for w in range(8)))
print("                      ->barrier           " + "".join(f"{(st[w, 1] - st[w, 15]) / 100.0:7.2f}" for w in range(8)))
def seg(a, b): return "".join(f"{(st[w, b] - st[w, a]) / 100.0:7.2f}" for w in range(8))
print("tail split (us): 11->LN done      " + seg(11, 19))
print("                 ->K pass 1 done   " + seg(19, 20))
print("                 ->image landed    " + seg(20, 21))
print("                 ->barrier         " + seg(21, 12))
print("                 ->pass 2 done     " + seg(12, 16))
print("                 ->barrier         " + seg(16, 17))
print("                 ->record written  " + seg(17, 13))
print("combine split (us): 0->loads issued   " + seg(0, 22))
print("                    ->phase A done    " + seg(22, 23))
print("                    ->barrier         " + seg(23, 24))
print("                    ->FMAs done       " + seg(24, 25))
print("                    ->frags written   " + seg(25, 14))
ck = nat.debug_read("stamps", np.uint64, 8 * 32).astype(np.int64)[252:256]
if ck[3] > ck[1]:
    print(f"k_film_gemm workgroup 5: {(ck[3] - ck[1]) / 100.0:.1f} us, core clock {(ck[2] - ck[0]) / (ck[3] - ck[1]) * 100.0:.0f} MHz")
