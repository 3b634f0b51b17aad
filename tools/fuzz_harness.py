#!/usr/bin/env python3
"""GPU box: randomized calls of DDPMTrainer.generate_music_motion (trainers/ddpm_trainer.py:183-201: encode_music + the DDIM loop) against
the oracle's generate_music_motion - batches of mel spectrograms of random length as pinned host memory (the pipelined H2D + encode
path), pageable host memory, numpy and device tensors; both MusicEncoder formats, optional snapshots.  Test infrastructure: the oracle is
the checker.  usage: python tools/fuzz_harness.py [cases] [seed]"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import O, batch_mel, batch_noise, make_model, oracle_params, rel_l2  # noqa: E402
from diffusion_conductor_amd import DDPMTrainer  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
trainers, bad, worst, t0 = {}, 0, 0.0, time.perf_counter()
for case in range(N):
    S = int(rng.choice([25, 50]))
    Tm = int(rng.choice([int(rng.integers(4, 120)), int(rng.integers(120, 1500)), 270, 2700]))
    B = int(rng.integers(1, 40))
    T = (Tm - 1) // 3 + 1
    while B * T * S > 90000 and B > 1:
        B = max(1, B // 2)
    if B * T * S > 90000:
        S = 25
    kind = str(rng.choice(["pinned", "pageable", "numpy", "device"]))
    fmt = str(rng.choice(["f16", "split"]))
    prec = str(rng.choice(["fp16", "fp16", "mixed"]))
    first = int(rng.integers(0, 50))
    key = (prec, S)
    if key not in trainers:
        trainers[key] = DDPMTrainer(types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=S, is_train=False), make_model(prec))
        trainers[key].eval_mode()
    mel_np = batch_mel(B, Tm, first=first)
    noise = torch.from_numpy(batch_noise(B, T, first=first))
    with torch.no_grad():
        ref = O.generate_music_motion(oracle_params(), torch.from_numpy(mel_np), 26, S, noise)
    mel = {"pinned": lambda: torch.from_numpy(mel_np).pin_memory(), "pageable": lambda: torch.from_numpy(mel_np), "numpy": lambda: mel_np,
           "device": lambda: torch.from_numpy(mel_np).cuda()}[kind]()
    if B == 1 and kind == "numpy" and rng.random() < 0.5:
        mel = mel_np[0]              # the reference's own call: one [Tm, 128] array
    os.environ["DC_ME_PREC"] = fmt
    try:
        out = trainers[key].generate_music_motion(mel, 26, noise=noise)
        torch.cuda.synchronize()
    finally:
        del os.environ["DC_ME_PREC"]
    e = max(rel_l2(out[c:c + 1], ref[c:c + 1]) for c in range(B))
    ok = tuple(out.shape) == (B, T, 26) and bool(torch.isfinite(out).all()) and e <= 1e-3
    bad += not ok
    worst = max(worst, e)
    print(f"case {case:3d} B={B:2d} Tm={Tm:4d} T={T:4d} S={S} {kind:8s} encoder {fmt:5s} {prec:5s}: worst clip {e:.3e}{'' if ok else '   <-- FAIL'}", flush=True)
print(f"{N} cases, {bad} failures, {time.perf_counter() - t0:.0f} s; worst clip {worst:.3e}")
sys.exit(1 if bad else 0)
