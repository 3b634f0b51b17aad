#!/usr/bin/env python3
"""GPU box: the batched evaluation driver (evaluate.py, replaces tools/eval_new.py:104-134) on datasets of random size with random batch
sizes: its pipelined form (loader / scorer threads, two staging slots, graph park for the last short batch) must give every clip the MSE
the serial form (DC_EVAL_SERIAL=1) gives - bit for bit where both run the same kernels, within 1e-4 where the pipelined form's unchecked
loops take the small-batch layer kernel without the in-launch combine exchange (evaluate.py: DC_L16_OWN_COMBINE; another summation order) -,
the same again when repeated, and within 2e-3 for another batch size.
usage: python tools/fuzz_evaluate.py [cases] [seed]"""
import os
import shutil
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import batch_mel, make_model  # noqa: E402
from diffusion_conductor_amd import DDPMTrainer  # noqa: E402
from diffusion_conductor_amd import evaluate as ev  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tr = DDPMTrainer(types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=25, is_train=False), make_model("fp16"))
tr.eval_mode()
bad, t0 = 0, time.perf_counter()
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
for case in range(N):
    n = int(rng.integers(1, 60))
    Tm = int(rng.choice([270, 271, 540, 811]))
    T = (Tm - 1) // 3 + 1
    bs = int(rng.integers(1, 34))
    bs2 = int(rng.integers(1, 34))
    seed = int(rng.integers(0, 1000))
    root = tempfile.mkdtemp(prefix="dc_fuzz_eval_", dir=base)
    try:
        mels = batch_mel(n, Tm, first=int(rng.integers(0, 50)))
        for i in range(n):
            d = os.path.join(root, f"{i:03d}")
            os.mkdir(d)
            np.save(os.path.join(d, "mel.npy"), mels[i])
            np.save(os.path.join(d, "motion.npy"), rng.standard_normal((T, 13, 2)).astype(np.float32))
        a = ev.evaluate_dataset(tr, root, 26, batch_size=bs, seed=seed, verbose=False)
        a2 = ev.evaluate_dataset(tr, root, 26, batch_size=bs, seed=seed, verbose=False)
        os.environ["DC_EVAL_SERIAL"] = "1"
        try:
            s = ev.evaluate_dataset(tr, root, 26, batch_size=bs, seed=seed, verbose=False)
        finally:
            del os.environ["DC_EVAL_SERIAL"]
        b = ev.evaluate_dataset(tr, root, 26, batch_size=bs2, seed=seed, verbose=False)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    ids = [f"{i:03d}" for i in range(n)]
    rd = lambda x, y: max(abs(x["per_clip"][k] - y["per_clip"][k]) / abs(x["per_clip"][k]) for k in ids)     # noqa: E731
    same = list(a["per_clip"]) == ids and a["per_clip"] == a2["per_clip"]
    ds = rd(a, s)
    close = all(abs(b["per_clip"][k] - a["per_clip"][k]) <= 2e-3 * abs(a["per_clip"][k]) for k in ids)
    ok = same and ds <= 1e-4 and close and np.isfinite(a["final_mse"])
    bad += not ok
    print(f"case {case:3d} clips={n:2d} Tm={Tm} batch {bs:2d} / {bs2:2d}: pipelined == repeated {same}, vs serial {ds:.1e}, other batch size within 2e-3 {close}"
          f"{'' if ok else '   <-- FAIL'}", flush=True)
print(f"{N} cases, {bad} failures, {time.perf_counter() - t0:.0f} s")
sys.exit(1 if bad else 0)
