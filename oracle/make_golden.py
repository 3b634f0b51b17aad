#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

Run:  python oracle/make_golden.py [g1 .. g10]   (needs /root/reference; everything: ~7 min on 8 cores;
                                                 with section names only those fixtures are regenerated)

What it does
  1. stubs the imports the reference needs but this image lacks (`cv2`, `mmcv`), adds
     /root/reference/Diffusion_Stage to sys.path and imports the reference's
     MotionTransformer / GaussianDiffusion / DDPMTrainer;
  2. loads the seeded synthetic checkpoint (diffusion-conductor_amd/synthetic.py) with
     load_state_dict(strict=True) and switches to eval();
  3. runs the reference on seeded inputs and stores inputs (when not regenerable from a
     seed) and outputs as small fixtures;
  4. runs oracle/ddim_oracle.py on the same inputs and asserts agreement - this is what
     pins the oracle ("pinned against the reference run here").

Only data (inputs / expected outputs) is written; no reference source text is copied.
The GPU box never runs this file (it has no /root/reference).
"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/Diffusion_Stage"
OUT = os.path.join(ROOT, "tests", "golden")


def _stub_modules():
    cv2 = types.ModuleType("cv2")
    cv2.norm = lambda *a, **k: None
    sys.modules["cv2"] = cv2
    mmcv = types.ModuleType("mmcv")
    runner = types.ModuleType("mmcv.runner")
    runner.get_dist_info = lambda: (0, 1)
    runner.init_dist = lambda *a, **k: None
    utils = types.ModuleType("mmcv.utils")

    class Registry:
        def __init__(self, *a, **k):
            pass

        def register_module(self, *a, **k):
            return lambda c: c

    utils.Registry = Registry
    utils.build_from_cfg = lambda *a, **k: None
    parallel = types.ModuleType("mmcv.parallel")
    parallel.MMDataParallel = object
    parallel.MMDistributedDataParallel = object
    parallel.collate = lambda *a, **k: None
    mmcv.runner, mmcv.utils, mmcv.parallel = runner, utils, parallel
    for n, m in (("mmcv", mmcv), ("mmcv.runner", runner), ("mmcv.utils", utils), ("mmcv.parallel", parallel)):
        sys.modules[n] = m


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    assert os.path.isdir(REF), "the reference is only present in the build container"
    only = {a.lower() for a in sys.argv[1:]}            # e.g. `make_golden.py g8 g9`: regenerate those fixtures only

    def want(name):
        return not only or name in only
    _stub_modules()
    sys.path.insert(0, REF)
    from models.transformer import MotionTransformer  # noqa: E402  (the reference)
    from models import gaussian_diffusion as rgd       # noqa: E402
    from diffusion_conductor_amd.synthetic import (synthetic_state_dict, batch_mel, batch_noise,
                                                   batch_music_features)
    from diffusion_conductor_amd.param_spec import DenoiserConfig
    from oracle import ddim_oracle as O

    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    sd_np = synthetic_state_dict(DenoiserConfig(), seed=0)
    sd_t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    p = O.to_torch_params(sd_np)

    def ref_model(no_eff=False):
        m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128,
                              device="cpu", music_model_path=None, no_clip=True, no_eff=no_eff)
        m.load_state_dict(sd_t, strict=True)
        return m.eval()

    model = ref_model()
    log = {}

    from models.transformer import timestep_embedding as ref_te   # noqa: E402  (the reference)
    lazy = {}

    def model_no_eff():
        if "ne" not in lazy:
            lazy["ne"] = ref_model(no_eff=True)
        return lazy["ne"]

    def ref_diffusion(S, mean_type="START_X"):
        return rgd.GaussianDiffusion(betas=rgd.get_named_beta_schedule("linear", S),
                                     model_mean_type=getattr(rgd.ModelMeanType, mean_type),
                                     model_var_type=rgd.ModelVarType.FIXED_SMALL, loss_type=rgd.LossType.MSE)

    def ref_ddim(m, S, noise, xfp, xfo, length, idxs=(), clip_denoised=False, eta=0.0, mean_type="START_X"):
        return ref_diffusion(S, mean_type).ddim_sample_loop(
            m, tuple(noise.shape), noise=noise, clip_denoised=clip_denoised, progress=False, eta=eta,
            model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.as_tensor(length)}, idxs=list(idxs))

    def features(B, T, first=0, params=None):
        xf_ = torch.from_numpy(batch_music_features(B, T, first=first))
        q = params or p
        with torch.no_grad():
            return torch.nn.functional.linear(xf_, q["proj.weight"], q["proj.bias"]), xf_

    def make_g1():
        # ---- G1: schedule tables straight from the reference's GaussianDiffusion ----------
        g1 = {}
        for S in (50, 1000):
            betas = rgd.get_named_beta_schedule("linear", S)
            gd = rgd.GaussianDiffusion(betas=betas, model_mean_type=rgd.ModelMeanType.START_X,
                                       model_var_type=rgd.ModelVarType.FIXED_SMALL, loss_type=rgd.LossType.MSE)
            tab = O.ddim_tables(O.linear_beta_schedule(S))
            for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                      "sqrt_recipm1_alphas_cumprod"):
                ref = getattr(gd, k)
                assert np.array_equal(ref, tab[k]), (S, k)
                g1[f"S{S}_{k}"] = ref
            # the fp32 per-step scalars as ddim_sample forms them (gaussian_diffusion.py:812-830)
            t = torch.arange(S)
            shp = (S, 1)
            ex = rgd._extract_into_tensor
            ab, abp = ex(gd.alphas_cumprod, t, shp), ex(gd.alphas_cumprod_prev, t, shp)
            sigma = 0.0 * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
            co = torch.cat([ex(gd.sqrt_recip_alphas_cumprod, t, shp), ex(gd.sqrt_recipm1_alphas_cumprod, t, shp),
                            torch.sqrt(abp), torch.sqrt(1 - abp - sigma ** 2), sigma], dim=1).numpy()
            mine = O.ddim_step_coefficients(tab)
            assert np.allclose(co, mine, rtol=2e-7, atol=0), S   # torch sqrt: stride-0 vs contiguous paths differ by <=1 ulp
            g1[f"S{S}_step_coeff"] = co
        np.savez_compressed(os.path.join(OUT, "g1_schedule.npz"), **g1)


    def make_g2():
        # ---- G2: timestep embedding + time_embed MLP table ---------------------------------
        with torch.no_grad():
            t = torch.arange(1000)
            te = ref_te(t, 128)
            assert torch.equal(te, O.timestep_embedding(t, 128))
            table = model.time_embed(te).numpy()
        mine = torch.nn.functional.linear(torch.nn.functional.silu(torch.nn.functional.linear(
            te, p["time_embed.0.weight"], p["time_embed.0.bias"])), p["time_embed.2.weight"], p["time_embed.2.bias"]).numpy()
        assert np.array_equal(table, mine)
        rows = np.array([0, 1, 7, 24, 49, 500, 998, 999])
        np.savez_compressed(os.path.join(OUT, "g2_time_embed.npz"), rows=rows, table_rows=table[rows],
                            table_sum=np.float64(table.astype(np.float64).sum()),
                            table_abs_sum=np.float64(np.abs(table.astype(np.float64)).sum()))


    def make_g3():
        # ---- G3: block-level known answers at B=2, T=64, ragged length ---------------------
        B, T = 2, 64
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, T, 26, generator=g)
        h = torch.randn(B, T, 128, generator=g)
        xf_proj = torch.randn(B, T, 64, generator=g)
        xf_out = torch.randn(B, T, 64, generator=g)
        tt = torch.tensor([37, 5])
        length = torch.tensor([64, 40])
        g3 = dict(x=x.numpy(), h=h.numpy(), xf_proj=xf_proj.numpy(), xf_out=xf_out.numpy(), t=tt.numpy(),
                  length=length.numpy())
        with torch.no_grad():
            emb = model.time_embed(ref_te(tt, 128)).unsqueeze(1) + model.linear(xf_proj)
            xo = model.linear(xf_out)
            mask = model.generate_src_mask(T, length).unsqueeze(-1)
            blk = model.temporal_decoder_blocks[2]
            g3["emb"] = emb.numpy()
            g3["styl"] = blk.sa_block.proj_out(h, emb).numpy()
            g3["sa"] = blk.sa_block(h, emb, mask).numpy()
            g3["ca"] = blk.ca_block(h, xo, emb).numpy()
            g3["ffn"] = blk.ffn(h, emb).numpy()
            g3["layer"] = blk(h, xo, emb, mask).numpy()
            g3["forward"] = model(x, tt, length=length, xf_proj=xf_proj, xf_out=xf_out).numpy()
            m_ne = model_no_eff()
            blk_ne = m_ne.temporal_decoder_blocks[2]
            g3["full_sa"] = blk_ne.sa_block(h, emb, mask).numpy()
            g3["full_ca"] = blk_ne.ca_block(h, xo, emb).numpy()
            g3["forward_no_eff"] = m_ne(x, tt, length=length, xf_proj=xf_proj, xf_out=xf_out).numpy()
        pre = "temporal_decoder_blocks.2"
        with torch.no_grad():
            chk = {
                "styl": O.stylization(p, pre + ".sa_block.proj_out", h, emb),
                "sa": O.linear_self_attention(p, pre + ".sa_block", h, emb, mask, 8),
                "ca": O.linear_cross_attention(p, pre + ".ca_block", h, xo, emb, 8),
                "ffn": O.ffn(p, pre + ".ffn", h, emb),
                "full_sa": O.full_self_attention(p, pre + ".sa_block", h, emb, mask, 8),
                "full_ca": O.full_cross_attention(p, pre + ".ca_block", h, xo, emb, 8),
                "forward": O.denoiser_forward(p, x, tt, length, xf_proj, xf_out),
                "forward_no_eff": O.denoiser_forward(p, x, tt, length, xf_proj, xf_out, no_eff=True),
            }
        for k, v in chk.items():
            log[f"g3_{k}"] = rel_l2(v.numpy(), g3[k])
            assert log[f"g3_{k}"] < 5e-6, (k, log[f"g3_{k}"])  # fp32 summation-order noise only
        np.savez_compressed(os.path.join(OUT, "g3_blocks.npz"), **g3)


    def make_g4():
        # ---- G4: encode_music -----------------------------------------------------------------
        mel_small = torch.from_numpy(batch_mel(1, 270)[..., :])
        with torch.no_grad():
            rp, rx = model.encode_music(mel_small, "cpu")
            op_, ox = O.encode_music(p, mel_small)
        assert torch.equal(rp, op_) and torch.equal(rx, ox)
        mel_full = torch.from_numpy(batch_mel(1, 5400))
        with torch.no_grad():
            rpf, rxf = model.encode_music(mel_full, "cpu")
            opf, oxf = O.encode_music(p, mel_full)
        log["g4_full_xproj"] = rel_l2(opf.numpy(), rpf.numpy())
        assert log["g4_full_xproj"] < 1e-6
        np.savez_compressed(os.path.join(OUT, "g4_encode_music.npz"), small_x_proj=rp.numpy(), small_x=rx.numpy(),
                            full_x_proj_sub=rpf.numpy()[:, ::25], full_x_sub=rxf.numpy()[:, ::25])


    def make_g5():
        # ---- G5: end-to-end DDIM-50, config 1 (B=1, T=1800) -------------------------------
        xfp, xf = features(1, 1800)
        noise = torch.from_numpy(batch_noise(1, 1800))
        t0 = time.time()
        ref = ref_ddim(model, 50, noise, xfp, xf, [1800], idxs=(0, 24))
        log["g5_ref_seconds"] = time.time() - t0
        t0 = time.time()
        with torch.no_grad():
            mine = O.ddim_sample_loop(p, noise, xfp, xf, [1800], 50, idxs=(0, 24))
        log["g5_oracle_seconds"] = time.time() - t0
        for k in ref:
            log[f"g5_idx{k}"] = rel_l2(mine[k].numpy(), ref[k].numpy())
            assert log[f"g5_idx{k}"] < 1e-6, (k, log[f"g5_idx{k}"])
        np.savez_compressed(os.path.join(OUT, "g5_ddim50_b1.npz"), x0=ref[50].numpy(),
                            idx0_sub=ref[0].numpy()[:, ::20], idx24_sub=ref[24].numpy()[:, ::20])


    def make_g6():
        # ---- G6: 30 s clips (T=900) B=2 ragged, DDIM-50; DDIM-1000 B=1 --------------------
        xf2 = torch.from_numpy(batch_music_features(2, 900, first=10))
        with torch.no_grad():
            xfp2 = torch.nn.functional.linear(xf2, p["proj.weight"], p["proj.bias"])
        noise2 = torch.from_numpy(batch_noise(2, 900, first=10))
        ref2 = ref_ddim(model, 50, noise2, xfp2, xf2, [900, 700])
        with torch.no_grad():
            mine2 = O.ddim_sample_loop(p, noise2, xfp2, xf2, [900, 700], 50)
        log["g6_t900"] = rel_l2(mine2.numpy(), ref2.numpy())
        assert log["g6_t900"] < 1e-6
        t0 = time.time()
        xfp, xf = features(1, 1800)
        noise = torch.from_numpy(batch_noise(1, 1800))
        ref1000 = ref_ddim(model, 1000, noise, xfp, xf, [1800])
        log["g6_ref1000_seconds"] = time.time() - t0
        np.savez_compressed(os.path.join(OUT, "g6_variants.npz"), t900_x0=ref2.numpy(), ddim1000_x0=ref1000.numpy())


    def make_g6b():
        # ---- G6b: no_eff DDIM at small T (full T x T attention) ---------------------------
        xf3 = torch.from_numpy(batch_music_features(2, 96, first=20))
        with torch.no_grad():
            xfp3 = torch.nn.functional.linear(xf3, p["proj.weight"], p["proj.bias"])
        noise3 = torch.from_numpy(batch_noise(2, 96, first=20))
        ref3 = ref_ddim(model_no_eff(), 50, noise3, xfp3, xf3, [96, 70])
        with torch.no_grad():
            mine3 = O.ddim_sample_loop(p, noise3, xfp3, xf3, [96, 70], 50, no_eff=True)
        log["g6_no_eff"] = rel_l2(mine3.numpy(), ref3.numpy())
        assert log["g6_no_eff"] < 1e-5, log["g6_no_eff"]
        np.savez_compressed(os.path.join(OUT, "g6b_no_eff.npz"), x0=ref3.numpy())


    def make_g7():
        # ---- G7: the harness, DDPMTrainer.generate_music_motion ---------------------------
        import trainers.ddpm_trainer as rt  # noqa: E402

        class _NoPretrain:
            def __init__(self):
                self.motion_encoder = torch.nn.Identity()

        rt.MotionPretrain = _NoPretrain           # hard-coded /home/... checkpoint path
        opt = types.SimpleNamespace(device=torch.device("cpu"), diffusion_steps=50, is_train=False)
        rt.DDPMTrainer.to = lambda self, device: None
        trainer = rt.DDPMTrainer(opt, model)
        mel = batch_mel(1, 5400)[0]
        torch.manual_seed(1234)
        out = trainer.generate_music_motion(mel, 26)
        torch.manual_seed(1234)
        nz = torch.randn(1, 1800, 26)
        with torch.no_grad():
            mine = O.generate_music_motion(p, mel, 26, 50, nz)
        log["g7_harness"] = rel_l2(mine.numpy(), out.numpy())
        assert log["g7_harness"] < 1e-6
        np.savez_compressed(os.path.join(OUT, "g7_harness.npz"), x0=out.numpy(), torch_seed=np.int64(1234))



    def make_g8():
        # ---- G8: DDIM-50, B=1, T=1800 on OTHER checkpoints / inputs than every fixture above uses (they all share
        # synthetic_state_dict(seed=0) and white-noise inputs): two more initialisation-scale draws, the "trained-like"
        # stress checkpoint (synthetic.stress_state_dict) and a smooth mel through the reference's own encode_music.
        from diffusion_conductor_amd.synthetic import smooth_mel, stress_state_dict
        g8 = {}

        def one(tag, sd, first, mel=None):
            q = O.to_torch_params(sd)
            m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cpu",
                                  music_model_path=None, no_clip=True)
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            m.eval()
            if mel is None:
                xfp_, xf_ = features(1, 1800, first=first, params=q)
            else:
                with torch.no_grad():
                    xfp_, xf_ = m.encode_music(torch.from_numpy(mel[None]), "cpu")
                    oxp, ox = O.encode_music(q, torch.from_numpy(mel[None]))
                assert torch.equal(xfp_, oxp) and torch.equal(xf_, ox)
            nz = torch.from_numpy(batch_noise(1, 1800, first=first))
            ref = ref_ddim(m, 50, nz, xfp_, xf_, [1800])
            with torch.no_grad():
                mine = O.ddim_sample_loop(q, nz, xfp_, xf_, [1800], 50)
            log[f"g8_{tag}"] = rel_l2(mine.numpy(), ref.numpy())
            assert log[f"g8_{tag}"] < 1e-6, (tag, log[f"g8_{tag}"])
            g8[f"{tag}_x0"] = ref.numpy()
            log[f"g8_{tag}_absmax_x0"] = float(ref.abs().max())

        one("seed1", synthetic_state_dict(DenoiserConfig(), seed=1), 30)
        one("seed2", synthetic_state_dict(DenoiserConfig(), seed=2), 31)
        one("stress", stress_state_dict(DenoiserConfig(), seed=0), 32)
        one("smooth", sd_np, 33, mel=smooth_mel(33))
        np.savez_compressed(os.path.join(OUT, "g8_robust.npz"), **g8)

    def make_g9():
        # ---- G9: the sampler's other branches (gaussian_diffusion.py:503-521, 812-830, 876) at B=2, T=96, ragged, DDIM-50,
        # on the seed-0 checkpoint (40 % of its x0 predictions leave [-1, 1], so the clamp is active): clip_denoised=True
        # (the default of ddim_sample_loop), eta=0.5, and ModelMeanType.EPSILON with clip + eta=0.3.  th.randn_like inside
        # ddim_sample is patched to hand out the seeded per-iteration noise, so the oracle / the HIP path can be fed the
        # same draws.  (Not the stress checkpoint: its predictions are ~300x outside [-1, 1], and a clamp at 1 turns a 3e-4
        # relative error of such a prediction into a 3e-2 absolute one on every element that crosses zero - any 11-bit
        # arithmetic, the oracle's own fp16 emulation included, reads 3e-3 there.)
        from diffusion_conductor_amd.synthetic import batch_step_noise
        sd = sd_np
        q = p
        m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cpu",
                              music_model_path=None, no_clip=True)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m.eval()
        B, T, S = 2, 96, 50
        length = [96, 70]
        xfp_, xf_ = features(B, T, first=40, params=q)
        nz = torch.from_numpy(batch_noise(B, T, first=40))
        z = torch.from_numpy(batch_step_noise(S, B, T, first=40))
        g9 = {}
        real_randn_like = torch.randn_like

        def run(tag, clip, eta, mean_type):
            calls = [0]

            def fake(x, *a, **k):
                assert tuple(x.shape) == (B, T, 26)
                calls[0] += 1
                return z[calls[0] - 1].clone()

            torch.randn_like = fake
            try:
                gd = ref_diffusion(S, mean_type)
                outs = list(gd.ddim_sample_loop_progressive(
                    m, (B, T, 26), noise=nz, clip_denoised=clip, progress=False, eta=eta,
                    model_kwargs={"xf_proj": xfp_, "xf_out": xf_, "length": torch.as_tensor(length)}))
                calls[0] = 0
                ref = ref_ddim(m, S, nz, xfp_, xf_, length, idxs=(0, 24), clip_denoised=clip, eta=eta, mean_type=mean_type)
            finally:
                torch.randn_like = real_randn_like
            assert calls[0] == S and torch.equal(outs[-1]["sample"], ref[S]) and torch.equal(outs[24]["sample"], ref[24])
            with torch.no_grad():
                mine, preds = O.ddim_sample_loop(q, nz, xfp_, xf_, length, S, idxs=(0, 24), clip_denoised=clip, eta=eta,
                                                 eps_model=mean_type == "EPSILON", step_noise=z, return_pred=True)
            for k in ref:
                log[f"g9_{tag}_idx{k}"] = rel_l2(mine[k].numpy(), ref[k].numpy())
                assert log[f"g9_{tag}_idx{k}"] < 1e-6, (tag, k, log[f"g9_{tag}_idx{k}"])
                g9[f"{tag}_idx{k}"] = ref[k].numpy()
            for it in (0, 24, 49):
                e = rel_l2(preds[it].numpy(), outs[it]["pred_xstart"].numpy())
                assert e < 1e-6, (tag, it, e)
                g9[f"{tag}_pred{it}"] = outs[it]["pred_xstart"].numpy()
            if clip:
                fr = float(np.mean([float((o["pred_xstart"].abs() >= 1).float().mean()) for o in outs]))
                log[f"g9_{tag}_clamped_fraction"] = fr
                assert fr > 0.01, "the clamp must be active for this fixture to test anything"

        run("clip", True, 0.0, "START_X")
        run("eta", False, 0.5, "START_X")
        run("eps", True, 0.3, "EPSILON")
        np.savez_compressed(os.path.join(OUT, "g9_sampler_branches.npz"), **g9)

    def make_g10():
        # ---- G10: no_eff (full T x T attention, transformer.py:198-287) DDIM-50 x0 at PRODUCTION length - G6b only has 3 key
        # tiles per clip; at T=1800 a query sees 57 key tiles, at T=900 29: the lengths at which an online softmax's running
        # reference point actually moves.  (a) seed-0 checkpoint, B=1, T=1800; (b) the "trained-like" stress checkpoint,
        # B=2, T=900, ragged lengths.
        from diffusion_conductor_amd.synthetic import stress_state_dict
        g10 = {}
        xfp_, xf_ = features(1, 1800, first=50)
        nz = torch.from_numpy(batch_noise(1, 1800, first=50))
        t0 = time.time()
        ref = ref_ddim(model_no_eff(), 50, nz, xfp_, xf_, [1800])
        log["g10_ref_t1800_seconds"] = time.time() - t0
        with torch.no_grad():
            mine = O.ddim_sample_loop(p, nz, xfp_, xf_, [1800], 50, no_eff=True)
        log["g10_no_eff_t1800"] = rel_l2(mine.numpy(), ref.numpy())
        assert log["g10_no_eff_t1800"] < 1e-5, log["g10_no_eff_t1800"]
        g10["t1800_x0"] = ref.numpy()

        sd = stress_state_dict(DenoiserConfig(), seed=0)
        q = O.to_torch_params(sd)
        m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cpu",
                              music_model_path=None, no_clip=True, no_eff=True)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m.eval()
        length = [900, 613]
        xfp2, xf2 = features(2, 900, first=52, params=q)
        nz2 = torch.from_numpy(batch_noise(2, 900, first=52))
        ref2 = ref_ddim(m, 50, nz2, xfp2, xf2, length)
        with torch.no_grad():
            mine2 = O.ddim_sample_loop(q, nz2, xfp2, xf2, length, 50, no_eff=True)
        log["g10_no_eff_stress_t900"] = rel_l2(mine2.numpy(), ref2.numpy())
        assert log["g10_no_eff_stress_t900"] < 1e-5, log["g10_no_eff_stress_t900"]
        g10["stress_t900_x0"] = ref2.numpy()
        g10["stress_t900_length"] = np.asarray(length, np.int64)
        np.savez_compressed(os.path.join(OUT, "g10_no_eff_long.npz"), **g10)

    def make_g11():
        # ---- G11: the sampler's off-path branches - ModelMeanType.PREVIOUS_X (gaussian_diffusion.py:510-514, 545), the learned-variance
        # output split (:472-486) and cond_fn / condition_score (:581-603, 806-808).  (a) host arithmetic on weight-free callables
        # (oracle/toy_models.py), every (mean type x variance type x cond_fn x clip x eta) branch, plus p_mean_variance's whole
        # dict; (b) the same three branches around the reference MotionTransformer at B=2, T=96 (ragged), DDIM-50, for the GPU
        # test whose model call runs in the HIP library.  Here the fixture IS the reference's output (the oracle module has no
        # counterpart of these branches: the package's sampler.py is compared with the fixture directly).
        from oracle import toy_models as TM
        g11 = {}
        real_randn_like = torch.randn_like
        S = 50
        gen = torch.Generator().manual_seed(1234)
        x_T = torch.randn(TM.SHAPE, generator=gen)
        z = torch.randn((S,) + TM.SHAPE, generator=gen)
        g11["toy_x_T"], g11["toy_z"] = x_T.numpy(), z.numpy()

        def gd_of(mean, var):
            return rgd.GaussianDiffusion(betas=rgd.get_named_beta_schedule("linear", S), model_mean_type=getattr(rgd.ModelMeanType, mean),
                                         model_var_type=getattr(rgd.ModelVarType, var), loss_type=rgd.LossType.MSE)

        def patched(draws, fn):
            calls = [0]

            def fake(x, *a, **k):
                calls[0] += 1
                return draws[calls[0] - 1].clone()
            torch.randn_like = fake
            try:
                return fn()
            finally:
                torch.randn_like = real_randn_like

        cases = [("prevx_small_clip", "PREVIOUS_X", "FIXED_SMALL", False, True, 0.0),
                 ("prevx_large_eta", "PREVIOUS_X", "FIXED_LARGE", False, True, 0.4),
                 ("startx_learned_clip", "START_X", "LEARNED", False, True, 0.0),
                 ("eps_range_eta", "EPSILON", "LEARNED_RANGE", False, False, 0.3),
                 ("startx_cond", "START_X", "FIXED_SMALL", True, False, 0.0),
                 ("eps_cond_clip_eta", "EPSILON", "FIXED_SMALL", True, True, 0.2),
                 ("prevx_range_cond", "PREVIOUS_X", "LEARNED_RANGE", True, True, 0.1)]
        for tag, mean, var, cond, clip, eta in cases:
            gd = gd_of(mean, var)
            mdl = TM.toy_model_learned if var.startswith("LEARNED") else TM.toy_model
            outs = patched(z, lambda: list(gd.ddim_sample_loop_progressive(
                mdl, TM.SHAPE, noise=x_T, clip_denoised=clip, cond_fn=TM.toy_cond_fn if cond else None, progress=False, eta=eta,
                device="cpu", model_kwargs={"scale": 0.8})))
            assert len(outs) == S and all(torch.isfinite(o["sample"]).all() for o in outs), tag
            g11[f"toy_{tag}_final"] = outs[-1]["sample"].numpy()
            for it in (0, 24, 49):
                g11[f"toy_{tag}_sample{it}"] = outs[it]["sample"].numpy()
                g11[f"toy_{tag}_pred{it}"] = outs[it]["pred_xstart"].numpy()
            tt = torch.tensor([49, 7, 0])
            with torch.no_grad():
                pmv = gd.p_mean_variance(mdl, x_T, tt, clip_denoised=clip, model_kwargs={"scale": 0.8})
            for k in ("mean", "variance", "log_variance", "pred_xstart"):
                g11[f"toy_{tag}_pmv_{k}"] = pmv[k].expand(TM.SHAPE).numpy().copy()
        # (b) around the reference denoiser
        B, T = 2, 96
        length = [96, 70]
        xfp_, xf_ = features(B, T, first=60)
        nz = torch.from_numpy(batch_noise(B, T, first=60))
        mk = {"xf_proj": xfp_, "xf_out": xf_, "length": torch.as_tensor(length)}

        def learned_wrap(x, t, **kw):          # a 2C-channel model whose first half is the denoiser (the variance half is not read by DDIM)
            y = model(x, t, **kw)
            return torch.cat([y, torch.zeros_like(y)], dim=1)

        for tag, mean, var, cond, clip in (("mt_cond", "START_X", "FIXED_SMALL", True, False),
                                           ("mt_prevx", "PREVIOUS_X", "FIXED_SMALL", False, True),
                                           ("mt_learned", "START_X", "LEARNED_RANGE", False, False)):
            gd = gd_of(mean, var)
            mdl = learned_wrap if var.startswith("LEARNED") else model
            outs = list(gd.ddim_sample_loop_progressive(mdl, (B, T, 26), noise=nz, clip_denoised=clip, progress=False, eta=0.0, device="cpu",
                                                        cond_fn=TM.pose_cond_fn if cond else None, model_kwargs=mk))
            g11[f"{tag}_final"] = outs[-1]["sample"].numpy()
            g11[f"{tag}_pred24"] = outs[24]["pred_xstart"].numpy()
            log[f"g11_{tag}_final_rms"] = float(outs[-1]["sample"].pow(2).mean().sqrt())
        plain = ref_ddim(model, S, nz, xfp_, xf_, length)
        assert torch.equal(plain, torch.from_numpy(g11["mt_learned_final"]))          # DDIM never reads the variance half
        log["g11_mt_cond_vs_plain"] = rel_l2(g11["mt_cond_final"], plain.numpy())     # (the guidance term must matter for the fixture to test it)
        assert log["g11_mt_cond_vs_plain"] > 1e-2
        g11["mt_length"] = np.asarray(length, np.int64)
        np.savez_compressed(os.path.join(OUT, "g11_offpath_branches.npz"), **g11)

    for name, fn in (("g1", make_g1), ("g2", make_g2), ("g3", make_g3), ("g4", make_g4), ("g5", make_g5), ("g6", make_g6),
                     ("g6b", make_g6b), ("g7", make_g7), ("g8", make_g8), ("g9", make_g9), ("g10", make_g10), ("g11", make_g11)):
        if want(name):
            t0 = time.time()
            fn()
            print(f"{name}: done in {time.time() - t0:.0f} s", flush=True)

    # PINNING.txt: one line per comparison; a partial run updates its own lines and keeps the others
    pin = os.path.join(OUT, "PINNING.txt")
    old = {}
    if only and os.path.exists(pin):
        for ln in open(pin).read().splitlines()[2:]:
            k, _, v = ln.partition(": ")
            old[k] = v
    old.update({k: f"{v:.3e}" for k, v in log.items()})
    with open(pin, "w") as f:
        f.write("oracle/ddim_oracle.py vs the imported reference (rel-L2; 0.0 = bit-identical)\n")
        f.write(f"torch {torch.__version__}, numpy {np.__version__}, threads {torch.get_num_threads()}\n")
        for k, v in old.items():
            f.write(f"{k}: {v}\n")
    for k, v in log.items():
        print(f"{k}: {v:.3e}")


if __name__ == "__main__":
    main()
