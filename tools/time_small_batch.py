#!/usr/bin/env python3
"""GPU box: ms per DDIM-50 loop at small batches (T = 1800), 16-token layer kernel (default) vs the 32-token narrow form
(DC_NO_LAYER16=1), alternating on ONE box.  Usage: python tools/time_small_batch.py [bs ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from helpers import make_model, make_diffusion, xf_pair, batch_noise, rel_l2
T, S = int(os.environ.get("DC_T", "1800")), 50
m = make_model("fp16")
coef = make_diffusion(S).native_coefficients()
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T)).cuda()
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
    res = {}
    for rep in range(3):
        for name, env in (("16-token", None), ("32-token", "DC_NO_LAYER16")):
            if env: os.environ[env] = "1"
            try:
                for _ in range(2): out, _ = nat.ddim_loop(noise, coef)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): out, _ = nat.ddim_loop(noise, coef)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            finally:
                if env: del os.environ[env]
            res.setdefault(name, []).append(dt * 1e3); res[name + "_out"] = out
    a, b = min(res["16-token"]), min(res["32-token"])
    print(f"bs={B} T={T}: 16-token {a:.3f} ms per loop ({['%.3f' % v for v in res['16-token']]}), 32-token {b:.3f} ({['%.3f' % v for v in res['32-token']]}): "
          f"{100 * (a / b - 1):+.1f} %; 16 vs 32 rel-L2 {rel_l2(res['16-token_out'], res['32-token_out'].cpu().numpy()):.2e}; status {nat.status()}", flush=True)
