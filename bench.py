#!/usr/bin/env python3
"""bench.py - DDIM sampling throughput of the MI355X-native sampler (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch: a full DDIM-50 sampling loop (x_T -> x_0) for
bs=32 clips of 60 s (T=1800 frames) per GPU - BASELINE.json configs[1].  Conditioning (xf_proj',
cross-attention matrices) and x_T are resident in HBM when the timed region starts (the loop-only
number SURVEY.md section 8d defines as the roofline number).  value = frames completed by all ranks
per second = n_gpus * 32 * 1800 * K / max-over-ranks(time).

Multi-GPU: one process per GPU.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the
environment is the PARENT: it never touches the GPU, starts N child processes of this file (one per
rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), forwards rank 0's
JSON line and exits with the children's worst return code.  Launched by `python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks run directly.  Clips are
sharded (weak scaling: 32 per GPU); the only collective is one RCCL all-gather of the final poses
per step, inside the timed region.  With N > 1 the JSON line also carries what proves the run was N ranks on N devices:
`multi_gpu` = {rccl_ranks (dist.get_world_size()), devices (one "pci bus id / uuid" per rank, all-gathered, asserted
distinct), ms_per_step_by_rank (min, max), gather_block_equals_local (every rank found its own block of the all-gather bit-equal
to its local result), sharded_equals_single_gpu (rank 0 re-sampled the LAST rank's shard alone and found it bit-equal to that
rank's block - SURVEY.md section 8d, config 3)}.

Extra objects on the JSON line:
  roofline        the kernel with the largest share of the loop (k_layer, HBM roof; or k_film_gemm, MFMA
                  roof): algorithmic bytes / FLOPs per launch over its mean duration, measured with HIP
                  events on the library's own stream in a separate eager pass; `traffic` = HBM bytes per
                  launch from the committed PMC passes of this very command, `traffic_source` names the file
  cpu_baseline    the oracle (a port of the reference's eager path, PyTorch CPU ops) timed on this box's
                  host cores on config 1 (bs=1, T=1800): 1 warm-up + 3 full DDIM-50 loops, median;
                  loop-only (`value`) and end to end (`end_to_end`: encode_music + loop)
  bf16_mode       the same loop in the bf16-MFMA mode that meets the parity bound ("mixed")
  bs1             the reference's actual call pattern: one clip per call (ms per DDIM-50 loop)
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_FLOPS = 2.5e15            # dense bf16 MFMA, MI355X
FLOP_PER_TOKEN_STEP = 2 * 4250112   # algorithmic, hoisted (SURVEY.md section 8d / BASELINE.md section 4)
FILM_FLOP_PER_TOKEN = 2 * 512 * 6144
PEAK_HBM_BYTES = 8.0e12             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# HBM bytes per launch from the committed PMC passes of this command (tools/collect_profiles.sh + tools/make_traffic.py): newest round
TRAFFIC_FILE = max((os.path.join("profiles", f) for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_traffic\.json", f)),
                   default=os.path.join("profiles", "r02_traffic.json"))


def cpu_baseline(runs=3):
    """the oracle on the host cores, config 1 (one 60 s clip, DDIM-50, fp32):
    1 warm-up + `runs` full loops, median; loop-only and end to end (encode_music + loop).  PyTorch CPU ops do
    not scale to hundreds of threads at this size, so a short sweep over denoiser forwards picks the thread
    count first.  Returns (record, x0 of the last loop-only run) - the x0 doubles as the parity checker."""
    import numpy as np
    import torch
    from diffusion_conductor_amd.synthetic import batch_mel, batch_music_features, batch_noise, synthetic_state_dict
    from oracle import ddim_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))      # cores this process may actually run on
    except AttributeError:
        avail = os.cpu_count() or 1
    p = O.to_torch_params(synthetic_state_dict())
    xf = torch.from_numpy(batch_music_features(1, 1800))
    xfp = torch.nn.functional.linear(xf, p["proj.weight"], p["proj.bias"])
    x = torch.from_numpy(batch_noise(1, 1800))
    mel = torch.from_numpy(batch_mel(1, 5400))

    def fwd(n):
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(n):
                O.denoiser_forward(p, x, torch.tensor([49 - i]), [1800], xfp, xf)
        return (time.perf_counter() - t0) / n

    best_n, best = 1, float("inf")
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        fwd(1)
        dt = fwd(2)
        if dt < best:
            best_n, best = n, dt
    torch.set_num_threads(best_n)
    loop, e2e, x0 = [], [], None
    with torch.no_grad():
        for r in range(runs + 1):                 # run 0 = warm-up
            t0 = time.perf_counter()
            x0 = O.ddim_sample_loop(p, x, xfp, xf, [1800], 50)
            t1 = time.perf_counter()
            O.generate_music_motion(p, mel, 26, 50, x)
            t2 = time.perf_counter()
            if r:
                loop.append(t1 - t0)
                e2e.append(t2 - t1)
    tl, te = float(np.median(loop)), float(np.median(e2e))
    rec = {"value": round(1800 / tl, 1), "unit": "frames/s", "cores": best_n, "kind": "port",
           "sample": f"config 1 (bs=1, T=1800, DDIM-50, fp32): 1 warm-up + {runs} full loops, median {tl:.2f} s loop-only, "
                     f"{te:.2f} s end to end; fastest of 8/16/32/64 threads on {avail} available cores",
           "end_to_end": {"value": round(1800 / te, 1), "unit": "frames/s", "s": round(te, 3)}, "loop_s": round(tl, 3)}
    return rec, x0


def log(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bs", type=int, default=32, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=1800)
    ap.add_argument("--ddim", type=int, default=50)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "mixed", "bf16", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bf16_mode / bs1 / end_to_end side measurements")
    ap.add_argument("--no-eff", action="store_true", help="full T x T attention variant (reference --no_eff)")
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Parent of an N-rank run: spawns the ranks BEFORE anything here touches the GPU (this process never does),
    streams rank 0's stdout through and returns the worst child return code."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # DC_BENCH_WORKER: the rank program (tests/test_host_logic.py runs a stub through the real launcher); default: this file
        procs.append(subprocess.Popen([sys.executable, os.environ.get("DC_BENCH_WORKER") or os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    deadline = None
    while procs:
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0:
                worst = worst or rc
                deadline = deadline or time.time() + 30     # a failed rank leaves the others in a collective: bounded wait
        if deadline and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    return worst


def device_identity(local):
    """One string per physical device: PCI bus / device id + uuid + name (what the ranks all-gather to prove they are N devices)."""
    import torch
    pr = torch.cuda.get_device_properties(local)
    return f"{getattr(pr, 'pci_bus_id', '?'):02x}:{getattr(pr, 'pci_device_id', '?'):02x} {getattr(pr, 'uuid', '')} {pr.name}" \
        if isinstance(getattr(pr, "pci_bus_id", None), int) else f"{getattr(pr, 'uuid', '')} {pr.name} #{local}"


def multi_gpu_evidence(dist, dev, ident, rank, world, B, s_per_step, gathered, resample_own, resample_shard):
    """What proves an N-rank run (module docstring).  Collectives outside the timed region; every rank takes part.
    `ident`: this rank's device identity; `gathered`: the all-gathered poses [world * B, T, P] of the last timed step;
    `resample_own()`: this rank's shard sampled again; `resample_shard(r)`: rank r's shard sampled alone on THIS device (inputs
    regenerated from their seeds).  Host logic only - tests/test_sharding_gloo.py runs it on two gloo ranks with stub samplers."""
    import torch
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    assert len(set(idents)) == world, f"ranks share a device: {idents}"
    t = torch.tensor([s_per_step * 1e3], device=dev, dtype=torch.float64)
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    ms = [float(x.item()) for x in ts]
    # the gathered tensor is [world * B, T, P] in rank order: my block must be my own result, bit for bit
    mine = resample_own()
    ok = torch.tensor([int(torch.equal(gathered[rank * B:(rank + 1) * B], mine))], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    same_single = None
    if rank == 0:        # rank 0 samples the LAST rank's shard alone and compares
        r = world - 1
        same_single = bool(torch.equal(resample_shard(r), gathered[r * B:(r + 1) * B]))
    return {"rccl_ranks": int(dist.get_world_size()), "backend": dist.get_backend(), "devices": idents,
            "ms_per_step_by_rank": {"min": round(min(ms), 3), "max": round(max(ms), 3)},
            "gather_block_equals_local": bool(ok.item()), "sharded_equals_single_gpu": same_single,
            "all_gather_bytes_per_rank": int(mine.numel() * 4)}


def build_model(precision, no_eff, dev):
    import numpy as np
    import torch
    from diffusion_conductor_amd import MotionTransformer
    from diffusion_conductor_amd.synthetic import synthetic_state_dict
    model = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device=dev,
                              no_clip=True, precision=precision, no_eff=no_eff)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_state_dict().items()}, strict=True)
    return model.to(dev).eval()


def time_loops(nat, noise, coef, n, warm=1):
    import torch
    for _ in range(warm):
        nat.ddim_loop(noise, coef)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out, _ = nat.ddim_loop(noise, coef)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def rel_l2(a, b):
    import torch
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path for the sampler)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    selftest_multi = world == 1 and bool(os.environ.get("DC_BENCH_SELFTEST_MULTI"))     # exercise the N > 1 evidence code on one GPU
    if selftest_multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    result_out = sys.stdout
    if world > 1 or selftest_multi:
        # RCCL prints a version banner through C stdio on fd 1 (it lands BEHIND the result line when stdout is a pipe): keep fd 1 for
        # the ONE JSON line only - everything else that writes to "stdout" from here on goes to stderr
        sys.stdout.flush()
        result_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from diffusion_conductor_amd.sampler import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,
                                                 get_named_beta_schedule)
    from diffusion_conductor_amd.native import precise_tail_default
    from diffusion_conductor_amd.sharding import gather_poses
    from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise

    B, T, S = args.bs, args.frames, args.ddim
    model = build_model(args.precision, args.no_eff, dev)
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
        st = nat.status()                          # the product path's numeric health check (one stream synchronisation per loop)
        assert st == 0 or os.environ.get("DC_BENCH_ANY_STATUS"), f"sampler status {st}"      # (diagnostic builds with invalid results set the variable)
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
    dt = dt_local = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(out).all() or os.environ.get("DC_BENCH_ANY_STATUS")
    log(f"timed region: {dt:.3f} s for {args.steps} steps")
    multi = None
    if world > 1 or selftest_multi:
        def resample_shard(r):
            xf_r = torch.from_numpy(batch_music_features(B, T, first=r * B)).to(dev)
            xfp_r = torch.nn.functional.linear(xf_r, model.proj.weight, model.proj.bias).contiguous()
            nz_r = torch.from_numpy(batch_noise(B, T, first=r * B)).to(dev)
            return model.set_conditioning(xfp_r, xf_r, [T] * B).ddim_loop(nz_r, coef)[0]
        multi = multi_gpu_evidence(dist, dev, device_identity(local), rank, world, B, dt_local / args.steps, out,
                                   lambda: nat.ddim_loop(noise, coef)[0], resample_shard)

    frames = world * B * T * args.steps
    value = frames / dt
    headline = (B, T, S) == (32, 1800, 50) and not args.no_eff
    line = {
        "metric": "motion frames/sec (DDIM-50, 60s clip, bs=32 per GPU)", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp16": "f16", "mixed": "bf16x3+f16", "bf16": "bf16", "bf16x3": "bf16x3"}[args.precision], "data": "synthetic",
        "config": {"workload": f"{'configs[1]: ' if headline else ''}DDIM-{S} sampling loop, bs={B} clips/GPU x {T} frames "
                               f"({T // 30} s), {'full T x T attention (no_eff)' if args.no_eff else 'linear attention'}, "
                               f"precision={args.precision}, conditioning + x_T resident in HBM (loop-only)",
                   "clips_per_gpu": B, "frames_per_clip": T, "ddim_steps": S, "parallelism": f"clip-dp{world}",
                   # fp16 precision: the loop's last evaluation(s) run on split fp16 operands (dc_sampler_set_precise_tail, default 1;
                   # DC_PRECISE_TAIL overrides) - inside the timed loop, like everything else the product path does
                   "precise_tail_steps": int(os.environ.get("DC_PRECISE_TAIL", precise_tail_default(args.precision)))},
        # no_eff: 7.56 G + 13.27 G (T/1800) MAC per clip-step of 1800 tokens
        "mfma_roofline_frac_whole_loop": round(value / world * S * (2 * (7.56e9 + 13.27e9 * T / 1800) / 1800 if args.no_eff
                                                                      else FLOP_PER_TOKEN_STEP) / PEAK_BF16_FLOPS, 4),
    }
    if multi is not None:
        line["multi_gpu"] = multi
    if rank == 0:
        from diffusion_conductor_amd import native
        extras = not args.no_extras
        if extras:
            # end to end for the same batch (reported beside `value`, never as it): pinned host mel -> H2D -> encode_music
            # (HIP conv stack) -> set_conditioning (step-invariant pre-pass) -> DDIM loop -> poses back on the host
            from diffusion_conductor_amd.synthetic import batch_mel
            mel_h = torch.from_numpy(batch_mel(B, 3 * T)).pin_memory()
            e2e = {}
            # the two overlapped stages on their own first (separate, synchronised; second of two repetitions: the first sizes
            # the encoder's activation planes for 32 clips): what the pipelining below hides
            alone = {}
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                mel = mel_h.to(dev, non_blocking=True)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                model.encode_music(mel, dev)
                torch.cuda.synchronize(); t2 = time.perf_counter()
                alone = {"h2d_mel_alone_ms": round(1e3 * (t1 - t0), 2), "encode_music_alone_ms": round(1e3 * (t2 - t1), 2)}
            out_h = torch.empty(tuple(noise.shape), dtype=torch.float32).pin_memory()     # the poses land in a pinned buffer (as in evaluate.py)
            # ... through the drop-in's own entry point (trainers/ddpm_trainer.py:183-201): DDPMTrainer.generate_music_motion on the
            # pinned host batch - encode_music copies it in chunks beside the encoder (denoiser.py) -, poses into a pinned buffer
            import types
            from diffusion_conductor_amd import DDPMTrainer
            tr = DDPMTrainer(types.SimpleNamespace(device=dev, diffusion_steps=S, is_train=False), model)
            tr.eval_mode()
            te = []
            for rep in range(7):      # (2 warm-up calls - the first sizes buffers - then the median of 5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o2 = tr.generate_music_motion(mel_h, noise.shape[2], noise=noise)
                out_h.copy_(o2, non_blocking=True)
                torch.cuda.synchronize()
                te.append(time.perf_counter() - t0)
            tm = sorted(te[2:])[2]
            e2e = {"ms": round(1e3 * tm, 2), "frames_per_s": round(B * T / tm, 1), "calls_ms": [round(1e3 * v, 2) for v in te],
                   "path": "DDPMTrainer.generate_music_motion(pinned mel [B,5400,128]) -> poses in a pinned host buffer; median of calls 3-7"}
            for rep in range(2):      # the same calls with a synchronisation between the stages: where the time goes
                torch.cuda.synchronize()
                t = [time.perf_counter()]
                exp, ex = model.encode_music(mel_h, dev)
                torch.cuda.synchronize(); t.append(time.perf_counter())
                nat2 = model.set_conditioning(exp, ex, [T] * B)
                torch.cuda.synchronize(); t.append(time.perf_counter())
                o2, _ = nat2.ddim_loop(noise, coef)
                out_h.copy_(o2, non_blocking=True); torch.cuda.synchronize(); t.append(time.perf_counter())
                e2e.update({"h2d_mel_and_encode_music_ms": round(1e3 * (t[1] - t[0]), 2),
                            "set_conditioning_ms": round(1e3 * (t[2] - t[1]), 2), "loop_and_d2h_ms": round(1e3 * (t[3] - t[2]), 2)})
            e2e.update(alone)
            line["end_to_end"] = e2e
            log(f"end to end: {e2e}")
            nat = model.set_conditioning(xfp, xf, [T] * B)      # back to the benchmark's conditioning for the profile pass
        # roofline of the dominant kernel: separate eager pass with per-launch HIP events
        prof, _ = nat.profile_loop(noise, coef)
        log("profile pass done: " + ", ".join(f"{k} {v[0]:.2f}ms/{v[1]}" for k, v in prof.items() if v[1]))
        tot = sum(ms for ms, _ in prof.values())
        tj = {}
        if os.path.exists(os.path.join(ROOT, TRAFFIC_FILE)) and headline and args.precision == "fp16":
            tj = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
        alg = native.algorithmic_work(B * T)       # per-launch algorithmic bytes / FLOPs of the build's kernels

        def roof(name, prof=prof, tj=tj):
            ms, cnt = prof[name]
            per = ms / cnt * 1e-3
            w = dict(alg[name])
            ach, peak, unit = w["flops"] / per / 1e12, PEAK_BF16_FLOPS / 1e12, "TFLOP/s"      # the MFMA roof: the path's primary bound
            r = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": peak, "unit": unit,
                 "frac": round(ach / peak, 4), "traffic": tj.get(name, {}).get("traffic_bytes"),
                 "traffic_source": TRAFFIC_FILE if name in tj else None,
                 "avg_launch_us": round(per * 1e6, 1), "launches": cnt, "layers_per_launch": w.get("layers_per_launch", 1)}
            r["mfma_frac"] = r["frac"]
            if "design_bytes" in w:
                # beside it: the HBM bytes THIS decomposition moves per launch (fp32 residual stream in / out + the FiLM tiles the GEMM
                # wrote one launch earlier + unit records) over the launch time - design bytes, not section 8d's compulsory bytes
                r["hbm_design_bytes"] = int(w["design_bytes"])
                r["hbm_design_frac"] = round(w["design_bytes"] / per / PEAK_HBM_BYTES, 4)
            if name == "k_layer":
                r["note"] = ("the eighth layer launch of a step also does the NEXT step's front work (embedding of x_{t-1} + layer 0's self-attention "
                             "front half: there is no front kernel beside the first step's); flops per launch = the step's non-FiLM algorithmic "
                             "FLOPs / 8, front half included - as in the rounds in which a separate launch did that part")
            if name == "k_film_gemm":
                r["graph_kernel"] = "k_film_embed: the captured loop runs this GEMM and k_embed_front as ONE launch; this eager pass times them apart"
            return r

        big = sorted((k for k in prof if prof[k][1] and k in alg), key=lambda k: -prof[k][0])
        if not args.no_eff and big:
            line["roofline"] = roof(big[0])
            line["roofline"]["time_share_by_kernel"] = {k: round(v[0] / tot, 3) for k, v in prof.items() if v[1]}
            if "k_layer" in tj and ("k_film_gemm" in tj or "k_film_embed" in tj):
                # memory-side bytes of ONE step from the committed PMC passes (the FiLM launch + the layer launches, whose eighth carries the
                # next step's front work since round 6) and the rate the timed loop sustains with them; SURVEY section 8d's compulsory bytes
                # per step beside it (x_t in/out, pp, A_ca)
                sb = tj.get("k_film_gemm", tj.get("k_film_embed"))["traffic_bytes"] + model.num_layers * tj["k_layer"]["traffic_bytes"]
                line["roofline"]["step_bytes_pmc"] = int(sb)
                line["roofline"]["loop_avg_TBps"] = round(sb * S / (dt / args.steps) / 1e12, 3)
                line["roofline"]["step_bytes_compulsory"] = int(B * (2 * T * 26 * 4 + T * 512 * 4 + 8 * 16 * 1024))
            if len(big) > 1:
                line["roofline_second_kernel"] = roof(big[1])
        cpu_x0 = None
        if not args.no_cpu_baseline and world == 1:      # contract: rank 0 at N=1 only
            log("cpu baseline (oracle on host cores: 1 warm-up + 3 DDIM-50 loops at bs=1) ...")
            line["cpu_baseline"], cpu_x0 = cpu_baseline()
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        if extras and headline and world == 1:
            # the reference's own call pattern: one clip per call
            nat1 = model.set_conditioning(xfp[:1].contiguous(), xf[:1].contiguous(), [T])
            t1, o1 = time_loops(nat1, noise[:1].contiguous(), coef, 5)
            line["bs1"] = {"ms_per_loop": round(1e3 * t1, 3), "frames_per_s": round(T / t1, 1)}
            if cpu_x0 is not None:
                line["bs1"]["rel_l2_vs_oracle"] = float(f"{rel_l2(o1, cpu_x0):.3e}")
            log(f"bs=1: {line['bs1']}")
            # the throughput option: flat 256-token units (228 workgroups instead of the 256 clip-aligned ones; a clip then depends on its
            # neighbours at the rounding level) - dc_sampler_set_clip_aligned(0)
            nat = model.set_conditioning(xfp, xf, [T] * B)      # (the one-clip run above re-conditioned the same sampler object)
            nat.set_clip_aligned(False)
            tf, of = time_loops(nat, noise, coef, 3)
            nat.set_clip_aligned(None)
            oa, _ = nat.ddim_loop(noise, coef)
            line["flat_units"] = {"ms_per_step": round(1e3 * tf, 3), "frames_per_s": round(B * T / tf, 1),
                                  "worst_clip_rel_l2_vs_clip_aligned": float(f"{max(rel_l2(of[c:c + 1], oa[c:c + 1]) for c in range(B)):.3e}"),
                                  "note": "value / ms_per_step above are the batch-invariant default (clip-aligned units)"}
            del of, oa
            if args.precision == "fp16":
                # the bf16-operand mode: plain bf16 MFMA operands in every GEMM (FiLM GEMM included), the loop's last 8 of 50 model
                # evaluations on split bf16 (three MFMAs per product; dc_sampler_set_precise_tail) - what carries its precision
                del nat1, nat
                m2 = build_model("bf16", False, dev)
                n2 = m2.set_conditioning(xfp, xf, [T] * B)
                tail2 = native.precise_tail_default("bf16")
                t2, o2 = time_loops(n2, noise, coef, 3)
                line["bf16_mode"] = {"precision": "bf16", "ms_per_step": round(1e3 * t2, 3), "frames_per_s": round(B * T / t2, 1),
                                     "rel_l2": float(f"{rel_l2(o2[:1], cpu_x0):.3e}") if cpu_x0 is not None else None,
                                     "precise_tail_steps": tail2,
                                     "operands": f"every GEMM on plain bf16 MFMA operands; the last {tail2} of the loop's {S} model evaluations in the "
                                                 "'mixed' form: 128-wide GEMMs on split bf16 (hi*hi + lo*hi + hi*lo), FiLM GEMM on f16 operands.  "
                                                 "plain_bf16_rel_l2: the same mode without that tail"}
                # the mode's own kernel-level evidence: eager passes with per-launch events over the loop as it runs (tail included), with the
                # tail off (the plain launches) and with every evaluation split (the split launches, the tail's f16 FiLM GEMM)
                prof_t, _ = n2.profile_loop(noise, coef)
                n2.set_precise_tail(0)
                prof_p, _ = n2.profile_loop(noise, coef)
                n2.set_precise_tail(S)                      # every evaluation split: the split instantiation's launch time directly
                prof_s, _ = n2.profile_loop(noise, coef)
                n2.set_precise_tail(-1)
                n_split = tail2 * model.num_layers
                split_us = prof_s["k_layer"][0] / max(prof_s["k_layer"][1], 1) * 1e3
                tot_t, tot_p = sum(v[0] for v in prof_t.values()), sum(v[0] for v in prof_p.values())
                line["bf16_mode"]["roofline"] = {
                    "k_film_gemm": roof("k_film_gemm", prof_p, {}), "k_layer_plain": roof("k_layer", prof_p, {}),
                    "k_layer_split_launch_us": round(split_us, 1), "k_layer_split_launches": n_split,
                    "k_film_gemm_tail_f16_launch_us": round(prof_s["k_film_gemm"][0] / max(prof_s["k_film_gemm"][1], 1) * 1e3, 1),
                    "tail_share_of_loop": round(1.0 - tot_p / tot_t * (S - tail2) / S, 4) if tot_t > 0 else None,
                    "eager_kernel_ms": {"with_tail": {k: round(v[0], 3) for k, v in prof_t.items() if v[1]},
                                        "tail_off": {k: round(v[0], 3) for k, v in prof_p.items() if v[1]}},
                    "note": "eager passes (per-launch HIP events on the library's stream); the captured loop fuses k_embed_front into the FiLM launch"}
                if cpu_x0 is not None:      # the plain-bf16 error (one clip, one loop)
                    n1b = m2.set_conditioning(xfp[:1].contiguous(), xf[:1].contiguous(), [T])
                    n1b.set_precise_tail(0)
                    o3, _ = n1b.ddim_loop(noise[:1].contiguous(), coef)
                    n1b.set_precise_tail(-1)
                    line["bf16_mode"]["plain_bf16_rel_l2"] = float(f"{rel_l2(o3, cpu_x0):.3e}")
                del n2, m2
                # ... and the split-operand mode (`mixed`: split bf16 in every 128-wide GEMM of every evaluation, f16 FiLM GEMM)
                m4 = build_model("mixed", False, dev)
                n4 = m4.set_conditioning(xfp, xf, [T] * B)
                t4, o4 = time_loops(n4, noise, coef, 3)
                line["split_mode"] = {"precision": "mixed", "ms_per_step": round(1e3 * t4, 3), "frames_per_s": round(B * T / t4, 1),
                                      "rel_l2": float(f"{rel_l2(o4[:1], cpu_x0):.3e}") if cpu_x0 is not None else None}
                log(f"bf16 mode: {line['bf16_mode']}; split mode: {line['split_mode']}")
        print(json.dumps(line), file=result_out, flush=True)
    if world > 1 or selftest_multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
