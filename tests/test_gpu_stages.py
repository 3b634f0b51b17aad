"""Per-block and per-stage parity of the HIP kernels (through the C-ABI test hooks), the full-size property tests of
BASELINE configs 4 and 5, RCCL on one GPU, and the visualization-shaped entry point.  Needs an MI355X.

Tolerances (rel-L2): split-bf16 everywhere ("bf16x3") <= 1e-4 - catches any layout / indexing error; the default
"fp16" mode and "mixed" <= 1e-3 (north_star's bound).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import (O, ROOT, batch_mel, batch_noise, golden, make_diffusion, make_model, oracle_params, rel_l2,
                     state_dict_np, xf_pair)

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": 1e-4, "fp16": 1e-3, "mixed": 1e-3}


@pytest.fixture(scope="module")
def models():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return {p: make_model(p) for p in TOL}


# ---- G3: the reference's own block outputs (oracle/make_golden.py:144-160), decoder layer 2 on a random residual stream
@pytest.mark.parametrize("prec", list(TOL))
@pytest.mark.parametrize("name,first,last", [("sa", 1, 1), ("ca", 2, 2), ("ffn", 3, 3), ("layer", 1, 3)])
def test_block_golden(models, prec, name, first, last):
    g = golden("g3_blocks.npz")
    m = models[prec]
    nat = m.set_conditioning(torch.from_numpy(g["xf_proj"]).cuda(), torch.from_numpy(g["xf_out"]).cuda(), g["length"])
    out = nat.debug_layer(g["h"], g["t"], 2, first, last)
    err = rel_l2(out, g[name])
    d_err = rel_l2(out - g["h"], g[name] - g["h"])          # on what the block adds to the residual stream
    print(f"G3 {name}[{prec}] rel-L2 {err:.3e} (block delta {d_err:.3e}; |delta| gpu {np.linalg.norm(out - g['h']):.3e} "
          f"ref {np.linalg.norm(g[name] - g['h']):.3e})")
    assert np.isfinite(out).all() and err <= TOL[prec] and d_err <= 30 * TOL[prec]


def test_block_chain_vs_oracle(models):
    """Blocks 1..2 and 2..3 of layer 2 on the G3 residual stream against the oracle's composition of the same blocks."""
    g = golden("g3_blocks.npz")
    p = oracle_params()
    h, xfp, xfo = (torch.from_numpy(g[k]) for k in ("h", "xf_proj", "xf_out"))
    with torch.no_grad():
        F = torch.nn.functional
        te = O.timestep_embedding(torch.from_numpy(g["t"]), 128)
        emb = F.linear(F.silu(F.linear(te, p["time_embed.0.weight"], p["time_embed.0.bias"])), p["time_embed.2.weight"],
                       p["time_embed.2.bias"]).unsqueeze(1) + F.linear(xfp, p["linear.weight"], p["linear.bias"])
        xo = F.linear(xfo, p["linear.weight"], p["linear.bias"])
        mask = O.generate_src_mask(64, g["length"]).unsqueeze(-1)
        pre = "temporal_decoder_blocks.2"
        sa = O.linear_self_attention(p, pre + ".sa_block", h, emb, mask, 8)
        sa_ca = O.linear_cross_attention(p, pre + ".ca_block", sa, xo, emb, 8)
        ca = O.linear_cross_attention(p, pre + ".ca_block", h, xo, emb, 8)
        ca_ffn = O.ffn(p, pre + ".ffn", ca, emb)
    nat = models["bf16x3"].set_conditioning(xfp.cuda(), xfo.cuda(), g["length"])
    e12 = rel_l2(nat.debug_layer(g["h"], g["t"], 2, 1, 2), sa_ca)
    e23 = rel_l2(nat.debug_layer(g["h"], g["t"], 2, 2, 3), ca_ffn)
    print(f"blocks 1..2 rel-L2 {e12:.3e}; blocks 2..3 rel-L2 {e23:.3e}")
    assert e12 <= 1e-4 and e23 <= 1e-4


# ---- every stage of every layer of the real pipeline against the oracle's taps -----------------------------------------
@pytest.mark.parametrize("prec", ["bf16x3", "fp16"])
@pytest.mark.parametrize("B,T", [(3, 100), (2, 900)])      # per-group records + combine launches | workgroup records
def test_stage_taps(models, prec, B, T):
    p = oracle_params()
    xfp, xfo = xf_pair(B, T)
    x = torch.from_numpy(batch_noise(B, T))
    t = torch.tensor([(7 * b + 3) % 50 for b in range(B)])
    length = [T if b % 2 == 0 else max(1, T - 17 - b) for b in range(B)]
    taps = {}
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo, taps=taps)
    nat = models[prec].set_conditioning(xfp.cuda(), xfo.cuda(), length)
    xd = x.cuda()
    M = B * T
    nat.debug_denoise(xd, t.numpy(), 0, 0)
    torch.cuda.synchronize()
    worst = rel_l2(nat.read_h()[:M].reshape(B, T, 128), taps["h0"])
    assert worst <= TOL[prec], ("h0", worst)
    for i in range(8):
        for stage, tap in ((1, f"sa{i}"), (2, f"ca{i}"), (3, f"ffn{i}")):
            nat.debug_denoise(xd, t.numpy(), i + 1, stage)
            torch.cuda.synchronize()
            e = rel_l2(nat.read_h()[:M].reshape(B, T, 128), taps[tap])
            worst = max(worst, e)
            assert e <= TOL[prec], (tap, e)
    out = nat.denoise(xd, t.numpy())
    torch.cuda.synchronize()
    e = rel_l2(out, ref)
    print(f"stage taps B={B} T={T} [{prec}]: worst stage {worst:.3e}, forward {e:.3e}")
    assert e <= TOL[prec]


# ---- BASELINE configs 4 and 5 at full size ----------------------------------------------------------------------------
def _ddim(model, S, noise, xfp, xfo, length):
    gd = make_diffusion(S)
    B, T, P = noise.shape
    out = gd.ddim_sample_loop(model, (B, T, P), noise=noise.cuda(), clip_denoised=False, progress=False,
                              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(list(length))})
    torch.cuda.synchronize()
    return out


def test_config4_ddim1000_bs32_full_size(models):
    """bs=32, T=1800, the full 1000-step schedule (20 replays of a 50-step hipGraph): finite, bit-identical when re-run,
    clip 0 within the parity bound of the reference's DDIM-1000 output (G6)."""
    B, T = 32, 1800
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T))
    a = _ddim(models["fp16"], 1000, noise, xfp, xfo, [T] * B)
    b = _ddim(models["fp16"], 1000, noise, xfp, xfo, [T] * B)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    err = rel_l2(a[:1], golden("g6_variants.npz")["ddim1000_x0"])
    print(f"config 4 clip 0 rel-L2 {err:.3e}")
    assert err <= 1e-3
    # ... and a clip from inside the batch (both ends in units shared with a neighbour) against the oracle's DDIM-1000 of that clip
    # (one clip: the oracle's 1000 CPU steps per clip are the suite's longest single cost)
    idx = [17]
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise[idx], xfp[idx], xfo[idx], [T] * len(idx), 1000)
    errs = [rel_l2(a[i:i + 1], ref[k:k + 1]) for k, i in enumerate(idx)]
    print("config 4 interior clips vs oracle: " + "  ".join(f"clip {i}: {e:.3e}" for i, e in zip(idx, errs)))
    assert max(errs) <= 1e-3


def test_config5_t900_bs128_full_size(models):
    """bs=128 clips of 30 s (T=900), fp16, DDIM-50: finite, bit-identical when re-run, and clips 0-1 (lengths 900 / 700)
    within the parity bound of the reference's output for those two clips (G6)."""
    B, T = 128, 900
    xfp, xfo = xf_pair(B, T, first=10)
    noise = torch.from_numpy(batch_noise(B, T, first=10))
    length = [900, 700] + [900 - 7 * (b % 13) for b in range(2, B)]
    a = _ddim(models["fp16"], 50, noise, xfp, xfo, length)
    b = _ddim(models["fp16"], 50, noise, xfp, xfo, length)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    err = rel_l2(a[:2], golden("g6_variants.npz")["t900_x0"])
    print(f"config 5 clips 0-1 rel-L2 {err:.3e}")
    assert err <= 1e-3
    # ... and eight clips from inside the batch (ragged lengths, flat units shared with their neighbours) against the oracle's run of those clips
    idx = [5, 31, 32, 63, 64, 77, 126, 127]
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise[idx], xfp[idx], xfo[idx], [length[i] for i in idx], 50)
    errs = [rel_l2(a[i:i + 1], ref[k:k + 1]) for k, i in enumerate(idx)]
    print("config 5 interior clips vs oracle: " + "  ".join(f"clip {i}: {e:.3e}" for i, e in zip(idx, errs)))
    assert max(errs) <= 1e-3


# ---- harness: a pinned host batch at production length (chunked H2D beside the encoder) against the oracle end to end ---------------
def test_harness_pinned_batch_production_length_vs_oracle(models):
    """DDPMTrainer.generate_music_motion on a pinned host batch of 17 x 60 s of mel (the path bench.py's end_to_end times: from 16
    clips on the copy runs in two chunks, 8 + 9 here, beside the MusicEncoder, denoiser.py) against the oracle's encode_music + DDIM-25 of the same batch
    (transformer.py:289-340,447-459; gaussian_diffusion.py:871-915), per clip."""
    import types
    from diffusion_conductor_amd import DDPMTrainer
    B, S = 17, 25
    opt = types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=S, is_train=False)
    tr = DDPMTrainer(opt, models["fp16"])
    tr.eval_mode()
    assert B >= 2 * models["fp16"].h2d_chunk          # (the chunked path)
    mel = batch_mel(B, 5400)
    noise = torch.from_numpy(batch_noise(B, 1800, first=90))
    a = tr.generate_music_motion(torch.from_numpy(mel).pin_memory(), 26, noise=noise)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.generate_music_motion(oracle_params(), torch.from_numpy(mel), 26, S, noise)
    errs = [rel_l2(a[c:c + 1], ref[c:c + 1]) for c in range(B)]
    print(f"harness, pinned batch of {B}: per-clip rel-L2 " + " ".join(f"{e:.2e}" for e in errs))
    assert tuple(a.shape) == (B, 1800, 26) and max(errs) <= 1e-3


# ---- harness: seed= with a mel length that is not a multiple of 3 ---------------------------------------------------------
def test_harness_seed_odd_mel_length(models):
    import types
    from diffusion_conductor_amd import DDPMTrainer
    opt = types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=25, is_train=False)
    tr = DDPMTrainer(opt, models["fp16"])
    tr.eval_mode()
    mel = batch_mel(2, 271)                                  # (271 - 1) // 3 + 1 = 91 frames
    a = tr.generate_music_motion(mel, 26, seed=5)
    assert tuple(a.shape) == (2, 91, 26) and torch.isfinite(a).all()
    noise = torch.randn(2, 91, 26, generator=torch.Generator().manual_seed(5))
    b = tr.generate_music_motion(mel, 26, noise=noise)
    assert torch.equal(a, b)
    with torch.no_grad():
        ref = O.generate_music_motion(oracle_params(), torch.from_numpy(mel), 26, 25, noise)
    assert rel_l2(a, ref) <= 1e-3


# ---- RCCL on one GPU: init + the all-gather of the final poses -----------------------------------------------------------
_NCCL_SCRIPT = r"""
import os, sys, types
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from helpers import batch_mel, make_model
from diffusion_conductor_amd import DDPMTrainer
from diffusion_conductor_amd.sharding import gather_poses, dist_info
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist_info() == (0, 1) and dist.get_backend() == "nccl"
opt = types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=25, is_train=False)
tr = DDPMTrainer(opt, make_model("fp16"))
tr.eval_mode()
mel = batch_mel(3, 810)
a = tr.generate_music_motion(mel, 26, seed=3)            # sharded path: world-size-1 group, RCCL all_gather_into_tensor
g = gather_poses(a, 3)                                   # the collective on HIP tensors, directly
mel_h = torch.from_numpy(batch_mel(17, 810)).pin_memory()   # a pinned host batch stays on the host: the rank copies only its own clips,
p_ = tr.generate_music_motion(mel_h, 26, seed=4)            # in chunks beside the encoder (17 >= 2 * h2d_chunk: the pipelined path)
d_ = tr.generate_music_motion(mel_h.cuda(), 26, seed=4)     # the same batch already on the device
assert tuple(p_.shape) == (17, 270, 26) and torch.equal(p_, d_)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
b = tr.generate_music_motion(mel, 26, seed=3)            # no group: local path
assert tuple(a.shape) == (3, 270, 26) and torch.isfinite(a).all()
assert torch.equal(a, b) and torch.equal(g, a)
print("NCCL_OK")
"""


def test_rccl_world_size_one_gather():
    import socket
    with socket.socket() as sk:                 # a free port: concurrent runs on one box must not collide
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_SCRIPT.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "NCCL_OK" in r.stdout


def test_bench_multi_gpu_evidence_fields_on_one_rank():
    """bench.py's N > 1 evidence code (rccl_ranks, device identities, per-rank times, gather block == local == single-GPU re-run)
    runs on a one-rank RCCL group (DC_BENCH_SELFTEST_MULTI=1): no 8-GPU box is needed to know it does not fall over there."""
    import json
    env = dict(os.environ, DC_BENCH_SELFTEST_MULTI="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bs", "2", "--frames", "300", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    print(r.stderr[-1500:])
    assert r.returncode == 0
    out = r.stdout.strip().splitlines()
    assert len(out) == 1, out                     # exactly ONE line on stdout (RCCL's version banner goes to stderr)
    line = json.loads(out[0])
    m = line["multi_gpu"]
    assert m["rccl_ranks"] == 1 and m["backend"] == "nccl" and len(m["devices"]) == 1 and len(m["devices"][0]) > 4
    assert m["gather_block_equals_local"] is True and m["sharded_equals_single_gpu"] is True
    assert m["ms_per_step_by_rank"]["min"] > 0 and m["all_gather_bytes_per_rank"] == 2 * 300 * 26 * 4


# ---- tools/visualization.py-shaped entry point ---------------------------------------------------------------------------
def test_visualize_entry_point_end_to_end(tmp_path):
    """opt.txt -> get_opt -> build_models -> DDPMTrainer.load(latest.tar) -> generate_music_motion -> smooth -> np.save, against
    the oracle's generate_music_motion for the same mel / x_T and scipy's savgol_filter (tools/visualization.py:126,180-223)."""
    from scipy.signal import savgol_filter
    from diffusion_conductor_amd import visualize
    root = tmp_path / "checkpoints" / "ConductorMotion100" / "train"
    (root / "model").mkdir(parents=True)
    torch.save({"encoder": {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np().items()}, "ep": 3, "total_it": 77},
               root / "model" / "latest.tar")
    (root / "opt.txt").write_text(
        "------------ Options -------------\n"
        f"checkpoints_dir: {tmp_path / 'checkpoints'}\ndataset_name: ConductorMotion100\nname: train\nunit_length: 4\n"
        "latent_dim: 128\nnum_layers: 8\ndiffusion_steps: 50\nno_clip: True\nno_eff: False\nlr: 0.0002\nis_train: True\n"
        "-------------- End ----------------\n")
    mel = batch_mel(1, 810)[0]
    np.save(tmp_path / "mel.npy", mel)
    for smooth in (False, True):
        out_path = tmp_path / f"kp_{int(smooth)}.npy"
        argv = ["--opt_path", str(root / "opt.txt"), "--music_path", str(tmp_path / "mel.npy"), "--npy_path", str(out_path),
                "--gpu_id", "0", "--seed", "7"] + (["--smooth"] if smooth else [])
        got = visualize.main(argv)
        saved = np.load(out_path)
        assert saved.shape == (270, 13, 2) and np.array_equal(saved, got)
        noise = torch.randn(1, 270, 26, generator=torch.Generator().manual_seed(7))
        with torch.no_grad():
            ref = O.generate_music_motion(oracle_params(), torch.from_numpy(mel), 26, 50, noise)[0].numpy().reshape(270, 13, 2)
        if smooth:
            ref = savgol_filter(ref.astype(np.float64), 19, 5, axis=0)
        err = rel_l2(saved, ref)
        print(f"visualize (smooth={smooth}) rel-L2 {err:.3e}")
        assert err <= 1e-3
    # a directory of mels is sampled as one batch
    d = tmp_path / "mels"
    d.mkdir()
    for i, m in enumerate(batch_mel(2, 810)):
        np.save(d / f"{i}.npy", m)
    got = visualize.main(["--opt_path", str(root / "opt.txt"), "--music_path", str(d), "--gpu_id", "0", "--seed", "7"])
    assert got.shape == (2, 270, 13, 2) and np.isfinite(got).all()
