#!/usr/bin/env python3
"""bench.py - DDIM sampling throughput of the MI355X-native sampler (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch: a full DDIM-50 sampling loop (x_T -> x_0) for
bs=32 clips of 60 s (T=1800 frames) per GPU - BASELINE.json configs[1].  Conditioning (xf_proj',
cross-attention matrices) and x_T are resident in HBM when the timed region starts (the loop-only
number SURVEY.md section 8d defines as the roofline number).  value = frames completed by all ranks
per second = n_gpus * 32 * 1800 * K / max-over-ranks(time).

Multi-GPU: one process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE); clips are
sharded (weak scaling: 32 per GPU); the only collective is one RCCL all-gather of the final poses
per step, inside the timed region.

Extra objects on the JSON line:
  roofline      dominant kernel (k_film_gemm) algorithmic FLOPs per launch / its mean duration,
                measured with HIP events on the library's own stream in a separate eager pass;
                peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)
  cpu_baseline  the oracle (a port of the reference's eager path, PyTorch CPU ops) timed on this
                box's host cores on a bounded sample of config 1 (bs=1, T=1800, a few DDIM steps)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_FLOPS = 2.5e15            # dense bf16 MFMA, MI355X
FLOP_PER_TOKEN_STEP = 2 * 4250112   # algorithmic, hoisted (SURVEY.md section 8d / BASELINE.md section 4)
FILM_FLOP_PER_TOKEN = 2 * 512 * 6144
LAYER_BYTES_PER_TOKEN = 512 + 512 + 24 * 64 + 72 + 36    # k_layer, per token and layer
PEAK_HBM_BYTES = 8.0e12                                   # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def cpu_baseline(steps_sample=6):
    """Oracle on the host cores: bs=1, T=1800, `steps_sample` denoiser steps of the 50-step schedule
    (every step costs the same), extrapolated to frames/s of a full DDIM-50 run.  PyTorch CPU ops do not
    scale to hundreds of threads at this size, so a short sweep picks the fastest thread count."""
    import torch
    from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise, synthetic_state_dict
    from oracle import ddim_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))      # cores this process may actually run on
    except AttributeError:
        avail = os.cpu_count() or 1
    p = O.to_torch_params(synthetic_state_dict())
    xf = torch.from_numpy(batch_music_features(1, 1800))
    xfp = torch.nn.functional.linear(xf, p["proj.weight"], p["proj.bias"])
    x = torch.from_numpy(batch_noise(1, 1800))

    def run(n):
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(n):
                O.denoiser_forward(p, x, torch.tensor([49 - i]), [1800], xfp, xf)
        return (time.perf_counter() - t0) / n

    best_n, best = 1, float("inf")
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        run(1)
        dt = run(2)
        if dt < best:
            best_n, best = n, dt
    torch.set_num_threads(best_n)
    dt = run(steps_sample)
    return {"value": round(1800 / (50 * dt), 1), "unit": "frames/s", "cores": best_n, "kind": "port",
            "sample": f"bs=1 T=1800 fp32: {steps_sample} denoiser steps timed ({dt*1e3:.0f} ms/step) x50 = one DDIM-50 loop; "
                      f"fastest of 8/16/32/64 threads on {avail} available cores"}


def log(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bs", type=int, default=32, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=1800)
    ap.add_argument("--ddim", type=int, default=50)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "mixed", "bf16", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eff", action="store_true", help="full T x T attention variant (reference --no_eff)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from diffusion_conductor_amd import MotionTransformer
    from diffusion_conductor_amd.sampler import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,
                                                 get_named_beta_schedule)
    from diffusion_conductor_amd.sharding import gather_poses
    from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise, synthetic_state_dict

    B, T, S = args.bs, args.frames, args.ddim
    sd = synthetic_state_dict()
    model = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device=dev,
                              no_clip=True, precision=args.precision, no_eff=args.no_eff)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.START_X,
                           model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
    # this rank's shard of the global batch (clip ids rank*B .. rank*B+B-1), synthetic, resident in HBM
    xf = torch.from_numpy(batch_music_features(B, T, first=rank * B)).to(dev)
    xfp = torch.nn.functional.linear(xf, model.proj.weight, model.proj.bias).contiguous()
    noise = torch.from_numpy(batch_noise(B, T, first=rank * B)).to(dev)
    log("inputs resident; building native sampler + conditioning")
    nat = model.set_conditioning(xfp, xf, [T] * B)
    torch.cuda.synchronize()
    coef = gd.native_coefficients()
    log(f"conditioning done, workspace {nat.workspace_bytes() / 2**20:.0f} MiB")

    def step():
        out, _ = nat.ddim_loop(noise, coef)
        return gather_poses(out, world * B) if world > 1 else out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup {i} done")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(out).all()
    log(f"timed region: {dt:.3f} s for {args.steps} steps")

    frames = world * B * T * args.steps
    value = frames / dt
    line = {
        "metric": "motion frames/sec (DDIM-50, 60s clip, bs=32 per GPU)", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp16": "f16", "mixed": "bf16x3+f16", "bf16": "bf16", "bf16x3": "bf16x3"}[args.precision], "data": "synthetic",
        "config": {"workload": f"{'configs[1]: ' if (B, T, S) == (32, 1800, 50) and not args.no_eff else ''}DDIM-{S} sampling loop, bs={B} clips/GPU x {T} frames "
                               f"({T // 30} s), {'full T x T attention (no_eff)' if args.no_eff else 'linear attention'}, "
                               f"precision={args.precision}, conditioning + x_T resident in HBM (loop-only)",
                   "clips_per_gpu": B, "frames_per_clip": T, "ddim_steps": S, "parallelism": f"clip-dp{world}"},
        # no_eff: 7.56 G + 13.27 G (T/1800) MAC per clip-step of 1800 tokens
        "mfma_roofline_frac_whole_loop": round(value / world * S * (2 * (7.56e9 + 13.27e9 * T / 1800) / 1800 if args.no_eff
                                                                      else FLOP_PER_TOKEN_STEP) / PEAK_BF16_FLOPS, 4),
    }
    if rank == 0:
        # end to end for the same batch (reported beside `value`, never as it): pinned host mel -> H2D -> encode_music
        # (HIP conv stack) -> set_conditioning (step-invariant pre-pass) -> DDIM loop -> poses back on the host
        from diffusion_conductor_amd.synthetic import batch_mel
        mel_h = torch.from_numpy(batch_mel(B, 3 * T)).pin_memory()
        e2e = {}
        for rep in range(2):                      # first repetition warms the encoder's activation planes
            torch.cuda.synchronize()
            t = [time.perf_counter()]
            mel = mel_h.to(dev, non_blocking=True)
            torch.cuda.synchronize(); t.append(time.perf_counter())
            exp, ex = model.encode_music(mel, dev)
            torch.cuda.synchronize(); t.append(time.perf_counter())
            nat2 = model.set_conditioning(exp, ex, [T] * B)
            torch.cuda.synchronize(); t.append(time.perf_counter())
            o2, _ = nat2.ddim_loop(noise, coef)
            o2h = o2.cpu(); t.append(time.perf_counter())
            e2e = {"ms": round(1e3 * (t[-1] - t[0]), 2), "frames_per_s": round(B * T / (t[-1] - t[0]), 1),
                   "h2d_mel_ms": round(1e3 * (t[1] - t[0]), 2), "encode_music_ms": round(1e3 * (t[2] - t[1]), 2),
                   "set_conditioning_ms": round(1e3 * (t[3] - t[2]), 2), "loop_and_d2h_ms": round(1e3 * (t[4] - t[3]), 2)}
        line["end_to_end"] = e2e
        log(f"end to end: {e2e}")
        nat = model.set_conditioning(xfp, xf, [T] * B)      # back to the benchmark's conditioning for the profile pass
        # roofline of the dominant kernel: separate eager pass with per-launch HIP events
        prof, _ = nat.profile_loop(noise, coef)
        log("profile pass done: " + ", ".join(f"{k} {v[0]:.2f}ms/{v[1]}" for k, v in prof.items()))
        tot = sum(ms for ms, _ in prof.values())
        tj = {}
        tf = os.path.join(ROOT, "profiles", "r01_traffic.json")   # HBM bytes per launch from the committed PMC passes of this command
        if os.path.exists(tf) and (B, T, S, args.precision, args.no_eff) == (32, 1800, 50, "fp16", False):
            tj = json.load(open(tf))

        def film_roofline():
            ms, cnt = prof["k_film_gemm"]
            per = ms / cnt * 1e-3
            ach = FILM_FLOP_PER_TOKEN * B * T / per / 1e12
            return {"bound": "mfma", "kernel": "k_film_gemm", "achieved": round(ach, 1), "peak": PEAK_BF16_FLOPS / 1e12,
                    "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK_BF16_FLOPS, 4),
                    "traffic": tj.get("k_film_gemm", {}).get("traffic_bytes"), "avg_launch_us": round(per * 1e6, 1), "launches": cnt}

        def layer_roofline():
            # k_layer (one decoder layer for all tokens) moves per token: residual stream 512 B in + 512 B out, FiLM tiles
            # 24 x 64 B, workgroup records 72 B out + 36 B in and runs ~0.28 MFLOP: HBM is its nearer roof
            ms, cnt = prof["k_layer"]
            per = ms / cnt * 1e-3
            ach = LAYER_BYTES_PER_TOKEN * B * T / per / 1e9
            return {"bound": "hbm", "kernel": "k_layer", "achieved": round(ach, 1), "peak": PEAK_HBM_BYTES / 1e9, "unit": "GB/s",
                    "frac": round(ach * 1e9 / PEAK_HBM_BYTES, 4), "traffic": tj.get("k_layer", {}).get("traffic_bytes"),
                    "avg_launch_us": round(per * 1e6, 1), "launches": cnt}

        # `roofline` = the kernel with the largest share of the loop; the other of the two big kernels rides along
        if args.no_eff or prof["k_film_gemm"][0] >= prof["k_layer"][0]:
            line["roofline"], other = film_roofline(), (None if args.no_eff else layer_roofline())
        else:
            line["roofline"], other = layer_roofline(), film_roofline()
        line["roofline"]["time_share_by_kernel"] = {k: round(v[0] / tot, 3) for k, v in prof.items()}
        if other:
            line["roofline_second_kernel"] = other
        if not args.no_cpu_baseline and world == 1:      # contract: rank 0 at N=1 only
            log("cpu baseline (oracle on host cores) ...")
            line["cpu_baseline"] = cpu_baseline()
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
