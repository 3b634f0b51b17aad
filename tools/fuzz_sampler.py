#!/usr/bin/env python3
"""GPU box: randomized sampler branches of the captured loop (gaussian_diffusion.py:783-831, 871-965) against the oracle - eta > 0 with
the caller's per-iteration draws, clip_denoised, ModelMeanType.EPSILON, snapshot iterations (`idxs`), step counts on both sides of the
64-steps-per-graph split (several replays per loop, the precise tail in the last one), short ragged clips.  Test infrastructure: the
oracle is the checker.  usage: python tools/fuzz_sampler.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import O, batch_noise, make_model, oracle_params, rel_l2, xf_pair  # noqa: E402
from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_step_noise  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
models, bad, worst, t0 = {}, 0, {}, time.perf_counter()
for case in range(N):
    T = int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(200, 600)), 256, 320]))
    B = int(rng.integers(1, 9))
    while B * T > 1500 and B > 1:
        B -= 1
    S = int(rng.choice([int(rng.integers(21, 65)), 50, 65, 100, 128, 130]))
    if B * T * S > 60000:
        S = int(rng.integers(21, 51))
    eta = float(rng.choice([0.0, 0.0, 0.2, 0.5, 1.0]))
    clip = bool(rng.random() < 0.5)
    eps_model = bool(rng.random() < 0.3)
    if eps_model:
        clip = True        # (an unclipped epsilon model with random weights grows past fp16's range within a few steps: reported as overflow)
    idxs = sorted(set(int(v) for v in rng.integers(0, S, size=int(rng.integers(0, 4)))))
    length = [int(rng.integers(1, T + 1)) if rng.random() < 0.5 else T for _ in range(B)]
    prec = str(rng.choice(["fp16", "fp16", "mixed", "bf16"]))
    no_eff = bool(prec == "fp16" and T >= 32 and rng.random() < 0.35)          # (full attention is offered in fp16, from T = 32)
    first = int(rng.integers(0, 100))
    if (prec, no_eff) not in models:
        models[(prec, no_eff)] = make_model(prec, no_eff=no_eff)
    xfp, xfo = xf_pair(B, T, first=first)
    noise = torch.from_numpy(batch_noise(B, T, first=first))
    z = torch.from_numpy(batch_step_noise(S, B, T, first=first)) if eta else None
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S, eta=eta, idxs=tuple(idxs), clip_denoised=clip,
                                 eps_model=eps_model, step_noise=z, no_eff=no_eff)
    gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.EPSILON if eps_model else ModelMeanType.START_X,
                           model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
    kw = {"step_noise": z.cuda()} if eta else {}
    if eps_model and no_eff and eta == 0.0:      # refused since round 6 (outside the bound on some loops: 1.26e-3 at seed 802, case 3): must raise
        try:
            gd.ddim_sample_loop(models[(prec, no_eff)], (B, T, 26), noise=noise.cuda(), clip_denoised=clip, progress=False, eta=eta, idxs=idxs,
                                model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
            refused = False
        except ValueError:
            refused = True
        bad += not refused
        print(f"case {case:3d} B={B} T={T:3d} S={S:3d} eta={eta} clip={int(clip)} eps=1 fp16+no_eff: {'refused (EPSILON x no_eff x eta = 0)' if refused else 'NOT REFUSED   <-- FAIL'}", flush=True)
        continue
    out = gd.ddim_sample_loop(models[(prec, no_eff)], (B, T, 26), noise=noise.cuda(), clip_denoised=clip, progress=False, eta=eta, idxs=idxs,
                              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)}, **kw)
    torch.cuda.synchronize()
    if not idxs:
        out, ref = {S: out}, {S: ref}
    ok = set(out) == set(ref)
    # (an epsilon model divides by sqrt(1 / abar - 1) -> the snapshots of early iterations are compared as they are; the bound is the
    # parity bound of the final sample, snapshots may sit a little above it in the 16-bit modes: 2e-3; bf16's snapshots in front of its
    # precise tail are plain bf16: 2e-2)
    e_fin = rel_l2(out[S], ref[S]) if ok else float("inf")
    e_snap = max([rel_l2(out[k], ref[k]) for k in ref if k != S] + [0.0]) if ok else float("inf")
    ok = ok and all(bool(torch.isfinite(v).all()) for v in out.values()) and e_fin <= 1e-3 and e_snap <= (2e-2 if prec == "bf16" and not eps_model else 2e-3)
    bad += not ok
    tagp = prec + ("+no_eff" if no_eff else "")
    worst[tagp] = max(worst.get(tagp, 0.0), e_fin)
    print(f"case {case:3d} B={B} T={T:3d} S={S:3d} eta={eta:.1f} clip={int(clip)} eps={int(eps_model)} idxs={idxs} {tagp:11s}: final {e_fin:.3e} "
          f"snapshots {e_snap:.3e}{'' if ok else '   <-- FAIL'}", flush=True)
print(f"{N} cases, {bad} failures, {time.perf_counter() - t0:.0f} s; worst final sample per precision: " + ", ".join(f"{k} {v:.3e}" for k, v in sorted(worst.items())))
sys.exit(1 if bad else 0)
