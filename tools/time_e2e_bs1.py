#!/usr/bin/env python3
"""GPU box: the reference's own call - ONE clip per DDPMTrainer.generate_music_motion call (trainers/ddpm_trainer.py:183-201) - end to end:
numpy mel [5400,128] in, poses on the host out; stage split with a synchronisation between the stages.  usage: python tools/time_e2e_bs1.py"""
import os
import sys
import time
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from diffusion_conductor_amd import DDPMTrainer  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_mel  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model("fp16", False, dev)
tr = DDPMTrainer(types.SimpleNamespace(device=dev, diffusion_steps=50, is_train=False), model)
tr.eval_mode()
mel = batch_mel(1, 5400)[0]                      # numpy [5400,128], as tools/visualization.py hands it over
noise = torch.randn(1, 1800, 26)
ts = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = tr.generate_music_motion(mel, 26, noise=noise).cpu()
    ts.append(time.perf_counter() - t0)
print(f"generate_music_motion(one clip), numpy mel in -> poses on the host: calls {[round(1e3 * t, 2) for t in ts]} ms; median of the last 5: {1e3 * sorted(ts[3:])[2]:.2f} ms")
mel_d = torch.from_numpy(mel)[None].to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t = [time.perf_counter()]
    xp, x = model.encode_music(mel_d, dev); torch.cuda.synchronize(); t.append(time.perf_counter())
    nat = model.set_conditioning(xp, x, [1800]); torch.cuda.synchronize(); t.append(time.perf_counter())
    o, _ = nat.ddim_loop(noise.to(dev), tr.diffusion.native_coefficients()); torch.cuda.synchronize(); t.append(time.perf_counter())
print("stages (ms): encode_music %.2f, set_conditioning %.2f, loop %.2f" % tuple(1e3 * (t[i + 1] - t[i]) for i in range(3)))
