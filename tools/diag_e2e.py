#!/usr/bin/env python3
"""GPU box, diagnostic: per-call end-to-end times of generate_music_motion with host timestamps of its stages (encode_music returned,
loop returned), to find what an occasional +6 ms call spends its time on."""
import os, sys, time, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import make_model, batch_noise
from diffusion_conductor_amd import DDPMTrainer
from diffusion_conductor_amd.synthetic import batch_mel
B, T, S = 32, 1800, 50
dev = torch.device("cuda", 0)
m = make_model("fp16")
tr = DDPMTrainer(types.SimpleNamespace(device=dev, diffusion_steps=S, is_train=False), m); tr.eval_mode()
mel_h = torch.from_numpy(batch_mel(B, 3 * T)).pin_memory()
noise = torch.from_numpy(batch_noise(B, T)).cuda()
out_h = torch.empty((B, T, 26), dtype=torch.float32).pin_memory()
marks = []
enc0, loop0 = m.encode_music, tr.diffusion.ddim_sample_loop
def enc(*a, **k):
    marks.append(("enc_in", time.perf_counter())); r = enc0(*a, **k); marks.append(("enc_out", time.perf_counter())); return r
def loop(*a, **k):
    marks.append(("loop_in", time.perf_counter())); r = loop0(*a, **k); marks.append(("loop_out", time.perf_counter())); return r
m.encode_music = enc; tr.diffusion.ddim_sample_loop = loop
if os.environ.get("DC_DIAG_ALONE"):          # as bench.py does before its end-to-end calls
    for rep in range(2):
        mel = mel_h.to(dev, non_blocking=True); torch.cuda.synchronize(); enc0(mel, dev); torch.cuda.synchronize()
for i in range(int(os.environ.get("DC_DIAG_N", "30"))):
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = tr.generate_music_motion(mel_h, 26, noise=noise)
    t1 = time.perf_counter()
    out_h.copy_(o, non_blocking=True); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"call {i:2d}: total {1e3 * (t2 - t0):6.2f} ms; " + " ".join(f"{k} {1e3 * (t - t0):6.2f}" for k, t in marks) + f" returned {1e3 * (t1 - t0):6.2f}"
          f"; reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
