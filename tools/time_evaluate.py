#!/usr/bin/env python3
"""GPU box: throughput of the batched evaluation driver (evaluate.evaluate_dataset; replaces the serial loops of
Diffusion_Stage/tools/eval_new.py:104-134) on a synthetic dataset in the reference's on-disk format - `--clips` directories of
mel.npy [5400,128] + motion.npy [1800,13,2] on tmpfs - beside the bench's `end_to_end` definition measured in the same process
(one pinned batch through DDPMTrainer.generate_music_motion + D2H into a pinned buffer, median of 5 calls).
usage: python tools/time_evaluate.py [--clips 294] [--ddim 50] [--batch_size 32]   (DC_EVAL_SERIAL=1: round 4's behaviour - a stream
synchronisation and the MSEs between two batches - for the A/B)"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from diffusion_conductor_amd import DDPMTrainer, evaluate as ev  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_mel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=294)       # the reference's test split (README.md:74-81)
ap.add_argument("--ddim", type=int, default=50)
ap.add_argument("--batch_size", type=int, default=32)
ap.add_argument("--repeat", type=int, default=2)
args = ap.parse_args()

dev = torch.device("cuda", 0)
base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3e9 else None
root = tempfile.mkdtemp(prefix="dc_eval_", dir=base)
try:
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    for lo in range(0, args.clips, 32):
        n = min(32, args.clips - lo)
        mels = batch_mel(n, 5400, first=lo)
        for i in range(n):
            d = os.path.join(root, f"{lo + i:04d}")
            os.mkdir(d)
            np.save(os.path.join(d, "mel.npy"), mels[i])
            np.save(os.path.join(d, "motion.npy"), rng.standard_normal((1800, 13, 2)).astype(np.float32))
    print(f"dataset: {args.clips} clips under {root} ({time.perf_counter() - t0:.1f} s to write)", file=sys.stderr)
    model = bench.build_model("fp16", False, dev)
    tr = DDPMTrainer(types.SimpleNamespace(device=dev, diffusion_steps=args.ddim, is_train=False), model)
    tr.eval_mode()
    # the bench's end_to_end: one pinned batch per call, poses into a pinned buffer
    B = args.batch_size
    mel_h = torch.from_numpy(batch_mel(B, 5400)).pin_memory()
    noise = torch.randn(B, 1800, 26, device=dev)
    out_h = torch.empty(B, 1800, 26).pin_memory()
    te = []
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = tr.generate_music_motion(mel_h, 26, noise=noise)
        out_h.copy_(o, non_blocking=True)
        torch.cuda.synchronize()
        te.append(time.perf_counter() - t0)
    e2e = sorted(te[2:])[2]
    line = {"serial": bool(os.environ.get("DC_EVAL_SERIAL")), "clips": args.clips, "ddim": args.ddim, "batch_size": B, "end_to_end_ms_per_batch": round(1e3 * e2e, 2),
            "end_to_end_frames_per_s": round(B * 1800 / e2e, 1), "runs": []}
    for rep in range(args.repeat):
        r = ev.evaluate_dataset(tr, root, 26, batch_size=B, seed=1, verbose=False)
        line["runs"].append({"seconds": round(r["seconds"], 3), "frames_per_s": round(r["frames_per_s"], 1),
                             "steady_frames_per_s": round(r.get("steady_frames_per_s", 0.0), 1), "final_mse": r["final_mse"],
                             "main_thread_s": r["main_thread_s"]})
    best = max(x["frames_per_s"] for x in line["runs"])
    line["evaluate_over_end_to_end"] = round(best / line["end_to_end_frames_per_s"], 4)
    line["steady_over_end_to_end"] = round(max(x["steady_frames_per_s"] for x in line["runs"]) / line["end_to_end_frames_per_s"], 4)
    print(json.dumps(line))
finally:
    shutil.rmtree(root, ignore_errors=True)
