#!/usr/bin/env python3
"""GPU box: end-to-end ms of DDPMTrainer.generate_music_motion on a pinned host batch (bs=32, 60 s clips, DDIM-50) -> poses in a
pinned host buffer, for several host-to-device chunk schedules of encode_music (DC_H2D_CHUNKS), alternating on ONE box.
Usage: python tools/time_e2e.py ["" "4,28" "2,6,24" ...]"""
import os, sys, time, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import make_model, batch_noise
from diffusion_conductor_amd import DDPMTrainer
from diffusion_conductor_amd.synthetic import batch_mel
B, T, S = int(os.environ.get("DC_BS", "32")), 1800, 50
dev = torch.device("cuda", 0)
m = make_model("fp16")
tr = DDPMTrainer(types.SimpleNamespace(device=dev, diffusion_steps=S, is_train=False), m); tr.eval_mode()
mel_h = torch.from_numpy(batch_mel(B, 3 * T)).pin_memory()
noise = torch.from_numpy(batch_noise(B, T)).cuda()
out_h = torch.empty((B, T, 26), dtype=torch.float32).pin_memory()
scheds = sys.argv[1:] or ["", "4,28", "2,6,24", "16,16"]
res = {s: [] for s in scheds}
ref = None
for rep in range(4):
    for s in scheds:
        if s: os.environ["DC_H2D_CHUNKS"] = s
        else: os.environ.pop("DC_H2D_CHUNKS", None)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o = tr.generate_music_motion(mel_h, 26, noise=noise)
            out_h.copy_(o, non_blocking=True); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        if rep: res[s].append(float(np.median(ts)))
        if ref is None: ref = out_h.clone()
        assert torch.equal(ref, out_h), "schedules must not change the result"
for s in scheds:
    print(f"DC_H2D_CHUNKS={s or '(default)':10s}: median of 5, three rounds: " + " ".join(f"{v:.2f}" for v in res[s]) + " ms", flush=True)
