#!/usr/bin/env python3
"""Diagnostic: per-stage time (us) of the 8 waves of one k_layer workgroup (layer 3), from s_memrealtime stamps.
Run on the GPU box:  DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_model, make_diffusion, xf_pair, batch_noise
B, T = int(os.environ.get("DC_STAMP_BS", "32")), 1800
m = make_model(os.environ.get("DC_STAMP_PREC", "fp16"))
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
print("core clock (MHz) " + "".join(f"{(st[w, 27] - st[w, 26]) / max(st[w, 13] - st[w, 0], 1) * 100.0:8.0f}" for w in range(8)))
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
ef = nat.debug_read("stamps", np.uint64, 8 * 32 + 8).astype(np.int64)[256:264]
if ef[7] > ef[0]:
    names = ["x loads + joint_embed", "seq_emb, store h, LN", "images landed + barrier", "K proj + maxima + barrier", "V proj + partial tiles",
             "barrier", "record written"]
    print("k_embed_front workgroup 100, wave 0 (us): " + "   ".join(f"{n} {(ef[i + 1] - ef[i]) / 100.0:.2f}" for i, n in enumerate(names))
          + f"   total {(ef[7] - ef[0]) / 100.0:.2f}")
Tp = nat.clip_stride()
nwg = (B * Tp + 255) // 256
wg = nat.debug_read("stamps", np.uint64, 8 * 32 + 8 + 1024).astype(np.int64)[264:].reshape(2, 256, 2)[:, :nwg]
if wg[0, :, 1].max() > 0:
    t0 = wg[0, :, 0].min()
    q = lambda a: " ".join(f"{v / 100.0:7.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))
    print("per-workgroup stamps, us relative to the first workgroup of layer 3 (min p10 p50 p90 max):")
    for i, name in enumerate(("layer 3", "layer 4")):
        print(f"  {name} begin {q(wg[i, :, 0] - t0)}   end {q(wg[i, :, 1] - t0)}   lifetime {q(wg[i, :, 1] - wg[i, :, 0])}")
    print(f"  last end of layer 3 -> first begin of layer 4: {(wg[1, :, 0].min() - wg[0, :, 1].max()) / 100.0:.2f} us;"
          f"  first begin to first begin: {(wg[1, :, 0].min() - wg[0, :, 0].min()) / 100.0:.2f} us")
    try:
        first = nat.debug_read("stamps", np.uint64, 8 * 32 + 8 + 1024 + 1024 + 256 + 8 + 512).astype(np.int64)[2576:].reshape(2, 256)[:, :nwg]
        if first[1].max() > 0:
            su = (wg[:, :, 0] - first) / 100.0
            print(f"  first instruction -> begin stamp (kernel arguments, model record, workgroup map, residual-stream loads issued): "
                  f"layer 3 {q(wg[0, :, 0] - first[0])}   layer 4 {q(wg[1, :, 0] - first[1])}")
            print(f"  last end of layer 3 -> FIRST INSTRUCTION of layer 4: {(first[1].min() - wg[0, :, 1].max()) / 100.0:.2f} us;  "
                  f"first instructions of layer 4 spread over {(first[1].max() - first[1].min()) / 100.0:.2f} us")
    except Exception as e:
        print("  (no first-instruction stamps:", e, ")")
if wg[0, :, 1].max() > 0:
    # which workgroups are the slow ones?  logical index (wg_index: contiguous runs per XCD) and whether the 256-token unit spans two clips
    q_, r_ = nwg >> 3, nwg & 7
    logical = np.array([(b & 7) * q_ + min(b & 7, r_) + (b >> 3) for b in range(nwg)])
    strad = np.array([(w * 256) // Tp != min(w * 256 + 255, B * Tp - 1) // Tp for w in logical])
    life = (wg[0, :, 1] - wg[0, :, 0]) / 100.0
    if strad.any() and (~strad).any():
        print(f"  layer 3 lifetime: straddling units ({strad.sum()}) mean {life[strad].mean():.2f} max {life[strad].max():.2f};"
              f"  others mean {life[~strad].mean():.2f} max {life[~strad].max():.2f}")
    for x in range(min(8, nwg)):
        sel = (np.arange(nwg) & 7) == x
        print(f"  XCD {x}: {sel.sum()} workgroups, lifetime mean {life[sel].mean():.2f} max {life[sel].max():.2f}, end max {(wg[0, sel, 1].max() - t0) / 100.0:.2f}")
raw = nat.debug_read("stamps", np.uint64, 8 * 32 + 8 + 1024 + 1024 + 256 + 8).astype(np.int64)
fk = raw[1288:2312].reshape(256, 4)
units = raw[2312:2568]
if fk[:, 3].max() > 0:
    live = fk[:, 3] > 0
    nlive = int(live.sum())
    t0f = fk[live, 1].min()
    print(f"k_film_gemm: {nlive} workgroups; first start -> last start {(fk[live, 1].max() - t0f) / 100.0:.1f} us; first start -> last finish {(fk[live, 3].max() - t0f) / 100.0:.1f} us")
    print("k_film_gemm per XCD (workgroup b runs on XCD b & 7): core clock MHz (mean), sweep us (mean / max), finish us after the first start (max)")
    for x in range(8):
        sel = ((np.arange(256) & 7) == x) & live
        if not sel.any(): continue
        mhz = (fk[sel, 2] - fk[sel, 0]) / (fk[sel, 3] - fk[sel, 1]) * 100.0
        us = (fk[sel, 3] - fk[sel, 1]) / 100.0
        print(f"  XCD {x}: {mhz.mean():6.0f} MHz   {us.mean():6.1f} / {us.max():6.1f}   {(fk[sel, 3].max() - t0f) / 100.0:6.1f}")
    us_all = np.where(live, (fk[:, 3] - fk[:, 1]) / 100.0, 0.0)
    print("  all workgroups: sweep us min/p10/p50/p90/max " + " ".join(f"{v:.1f}" for v in np.percentile(us_all[live], [0, 10, 50, 90, 100]))
          + f";  units per workgroup min/mean/max {units[live].min()} / {units[live].mean():.2f} / {units[live].max()}")
    order = np.argsort(np.where(live, us_all, np.inf))[:nlive]
    print("  slowest 8 workgroups (id, units, us, us per unit): " + "  ".join(f"{i}:{units[i]}:{us_all[i]:.0f}:{us_all[i] / max(units[i], 1):.2f}" for i in order[-8:]))
    print("  fastest 8 workgroups (id, units, us, us per unit): " + "  ".join(f"{i}:{units[i]}:{us_all[i]:.0f}:{us_all[i] / max(units[i], 1):.2f}" for i in order[:8]))
    if nlive == 256:
        nfill = np.array([len(set(range(int(units[:i].sum()) // 12, (int(units[:i + 1].sum()) - 1) // 12 + 1))) for i in range(256)])
        for f in sorted(set(nfill)):
            sel = nfill == f
            print(f"  workgroups with {f} slab fills: {sel.sum()}, us per unit mean {np.mean(us_all[sel] / np.maximum(units[sel], 1)):.2f}")
    ft = raw[2568:2572]
    if ft[3] > 0:
        print(f"  workgroup 5: {ft[3]} slab fills, per fill (us): wait for the previous slab's slowest wave {ft[0] / ft[3] / 100.0:.2f},"
              f" SiLU + writes + second half {ft[1] / ft[3] / 100.0:.2f}, last loads + barrier {ft[2] / ft[3] / 100.0:.2f}")
