"""Round-3 parity gates on MORE than the one initialisation-scale checkpoint the other fixtures share, and on the sampler
branches the harness does not use (clip_denoised / eta > 0 / EPSILON).  Fixtures: tests/golden/g8_robust.npz and
g9_sampler_branches.npz, produced by the imported reference (oracle/make_golden.py g8 g9).  Needs an MI355X.

Tolerance: rel-L2 on x0 <= 1e-3 (BASELINE.json north_star) for the default "fp16" mode and for "mixed" (the bf16-MFMA mode).
"""
import numpy as np
import pytest
import torch

from helpers import (DenoiserConfig, O, batch_noise, golden, make_diffusion, rel_l2, synthetic_state_dict, xf_pair)
from diffusion_conductor_amd.synthetic import batch_music_features, batch_step_noise, smooth_mel, stress_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _model(sd, precision):
    from diffusion_conductor_amd import MotionTransformer
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True,
                          precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to("cuda").eval()


def _features(sd, B, T, first):
    xf = torch.from_numpy(batch_music_features(B, T, first=first))
    w, b = torch.from_numpy(sd["proj.weight"]), torch.from_numpy(sd["proj.bias"])
    return torch.nn.functional.linear(xf, w, b), xf


_CASES = {"seed1": (lambda: synthetic_state_dict(DenoiserConfig(), seed=1), 30),
          "seed2": (lambda: synthetic_state_dict(DenoiserConfig(), seed=2), 31),
          "stress": (lambda: stress_state_dict(DenoiserConfig(), seed=0), 32),
          "smooth": (lambda: synthetic_state_dict(DenoiserConfig(), seed=0), 33)}


def _internal_maxima(nat):
    """max |E| (FiLM tiles, fp16 storage) and max |h| (residual stream) as the last step left them."""
    G = (nat.B * nat.clip_stride() + 31) // 32
    E = nat.debug_read("E", np.float16, G * 192 * 64 * 16).astype(np.float32)
    h = nat.debug_read("h", np.float32, G * 4 * 64 * 16)
    return float(np.abs(E).max()), float(np.abs(h).max())


@pytest.mark.parametrize("prec", ["fp16", "mixed"])
@pytest.mark.parametrize("case", ["seed1", "seed2", "stress", "smooth"])
def test_ddim50_other_checkpoints_g8(case, prec):
    """DDIM-50, B=1, T=1800 against the reference's own x0 on: two more init-scale draws, the trained-like stress
    checkpoint (x3 modulation / output weights, log-normal LayerNorm gains, 10x outlier channels) and a smooth mel through
    encode_music (the HIP MusicEncoder feeds the loop here, as in generate_music_motion)."""
    g = golden("g8_robust.npz")
    make_sd, first = _CASES[case]
    sd = make_sd()
    m = _model(sd, prec)
    noise = torch.from_numpy(batch_noise(1, 1800, first=first)).cuda()
    if case == "smooth":
        xfp, xfo = m.encode_music(torch.from_numpy(smooth_mel(first)[None]).cuda(), "cuda")
    else:
        xfp, xfo = (t.cuda() for t in _features(sd, 1, 1800, first))
    gd = make_diffusion(50)
    out = gd.ddim_sample_loop(m, (1, 1800, 26), noise=noise, clip_denoised=False, progress=False,
                              model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor([1800])})
    torch.cuda.synchronize()
    err = rel_l2(out, g[f"{case}_x0"])
    emax, hmax = _internal_maxima(m._native)
    print(f"g8[{case}][{prec}] x0 rel-L2 {err:.3e}  max|x0| {float(out.abs().max()):.3g}  max|E| {emax:.4g}  max|h| {hmax:.4g}  "
          f"status {m._native.status()}")
    assert torch.isfinite(out).all() and err <= TOL


def _g9_setup():
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    B, T, S = 2, 96, 50
    xfp, xfo = _features(sd, B, T, 40)
    return sd, B, T, S, [96, 70], xfp.cuda(), xfo.cuda(), torch.from_numpy(batch_noise(B, T, first=40)).cuda(), \
        torch.from_numpy(batch_step_noise(S, B, T, first=40)).cuda()


_BRANCH = {"clip": (True, 0.0, "START_X"), "eta": (False, 0.5, "START_X"), "eps": (True, 0.3, "EPSILON")}


def _diffusion(S, mean_type):
    from diffusion_conductor_amd.sampler import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,
                                                 get_named_beta_schedule)
    return GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=getattr(ModelMeanType, mean_type),
                             model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)


@pytest.mark.parametrize("prec", ["fp16", "mixed"])
@pytest.mark.parametrize("tag", ["clip", "eta", "eps"])
def test_sampler_branches_graph_path_g9(tag, prec):
    """clip_denoised=True (ddim_sample_loop's own default), eta > 0 and ModelMeanType.EPSILON inside the captured loop
    (dc_sampler_ddim_loop_ex) against the reference's outputs, with the idxs=[0,24] intermediates."""
    g = golden("g9_sampler_branches.npz")
    sd, B, T, S, length, xfp, xfo, noise, z = _g9_setup()
    clip, eta, mt = _BRANCH[tag]
    m = _model(sd, prec)
    gd = _diffusion(S, mt)
    res = gd.ddim_sample_loop(m, (B, T, 26), noise=noise, clip_denoised=clip, progress=False, eta=eta, idxs=[0, 24],
                              model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)}, step_noise=z)
    torch.cuda.synchronize()
    assert set(res) == {0, 24, S}
    errs = {k: rel_l2(res[k], g[f"{tag}_idx{k}"]) for k in res}
    print(f"g9[{tag}][{prec}] graph path rel-L2 " + "  ".join(f"idx{k} {v:.3e}" for k, v in errs.items()))
    assert all(torch.isfinite(v).all() for v in res.values()) and max(errs.values()) <= TOL
    # the captured loop is what ran (no per-step host loop): same call with the graph disabled is bit-identical
    import os
    os.environ["DC_DISABLE_GRAPH"] = "1"
    try:
        res2 = gd.ddim_sample_loop(m, (B, T, 26), noise=noise, clip_denoised=clip, progress=False, eta=eta, idxs=[0, 24],
                                   model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)}, step_noise=z)
    finally:
        del os.environ["DC_DISABLE_GRAPH"]
    assert all(torch.equal(res[k], res2[k]) for k in res)


@pytest.mark.parametrize("S", [50, 100])
def test_epsilon_model_runs_every_evaluation_on_split_operands(S):
    """ModelMeanType.EPSILON at eta = 0: the final sample is sqrt(1/abar) x_t - sqrt(1/abar - 1) eps, so the plain 16-bit evaluations' error
    stays in x_t whatever the precise tail (fp16 1.4 - 1.8e-3, found by tools/fuzz_sampler.py).  By default such a loop runs every evaluation
    on split operands (S = 100: two graph replays, both split) - in the bf16 precision with the FiLM GEMM's operands in fp16, i.e. as
    "mixed"-precision evaluations; an explicit tail is honoured.  No warning: the results are inside the bound."""
    import os
    import warnings
    from helpers import O
    sd, B, T, _, length, xfp, xfo, noise, _ = _g9_setup()
    gd = _diffusion(S, "EPSILON")
    with torch.no_grad():
        ref = O.ddim_sample_loop(O.to_torch_params(sd, torch.float32), noise.cpu(), xfp.cpu(), xfo.cpu(), length, S, clip_denoised=True, eps_model=True)
    errs = {}
    for prec in ("fp16", "bf16"):
        m = _model(sd, prec)

        def run():
            out = gd.ddim_sample_loop(m, (B, T, 26), noise=noise, clip_denoised=True, progress=False,
                                      model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)})
            torch.cuda.synchronize()
            return rel_l2(out, ref)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            errs[prec] = run()
        assert not w, [str(x.message) for x in w]
        os.environ["DC_PRECISE_TAIL"] = "1"
        try:
            errs[prec + ", tail 1"] = run()
        finally:
            del os.environ["DC_PRECISE_TAIL"]
    print(f"EPSILON, eta = 0, S = {S}: " + "  ".join(f"{k} {v:.3e}" for k, v in errs.items()))
    assert errs["fp16"] <= 0.5 * TOL and errs["fp16, tail 1"] > 2 * errs["fp16"]
    assert errs["bf16"] <= 0.5 * TOL and errs["bf16, tail 1"] > 3 * errs["bf16"]


def test_epsilon_model_full_attention_eta0_is_refused_eta_above_zero_is_inside_the_bound(monkeypatch):
    """EPSILON model x full attention (`no_eff`): k_layer_full's split-operand instantiation (query / key / value projections, stylization
    out-projections and FFN on split operands; scores, weights and values plain fp16) runs every evaluation of such a loop.  At eta = 0 that
    is NOT robustly inside the bound - 2.3e-4 ... 1.26e-3 over 14 randomized loops of tools/fuzz_sampler.py (this fixed case: 4.4e-4; round 5
    returned 1.4e-3 here with a warning) - so the combination is refused by the sampler and by the library's loop; with eta > 0 the fresh
    noise damps what the evaluations leave in x_t (<= 2e-4 on the same tool) and the loop runs.  DC_ALLOW_EPSILON_NO_EFF_ETA0=1 lets the
    refused loop run for the record."""
    from helpers import O
    from diffusion_conductor_amd import MotionTransformer, native
    sd, B, T, S, length, xfp, xfo, noise, z = _g9_setup()
    gd = _diffusion(S, "EPSILON")
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True, no_eff=True, precision="fp16")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    kw = dict(noise=noise, clip_denoised=True, progress=False, model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)})
    with pytest.raises(ValueError, match="no_eff"):
        gd.ddim_sample_loop(m, (B, T, 26), **kw)
    with pytest.raises(ValueError, match="no_eff"):
        next(gd.ddim_sample_loop_progressive(m, (B, T, 26), **kw))
    nat = m.set_conditioning(xfp, xfo, length)                       # ... and the library's own loop, under the Python layer
    with pytest.raises(native.DcError):
        nat.ddim_loop(noise, gd.native_coefficients(0.0), [], native.UPDATE_CLIP_DENOISED | native.UPDATE_EPSILON, None, None)
    out = gd.ddim_sample_loop(m, (B, T, 26), eta=0.3, step_noise=z, **kw)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.ddim_sample_loop(O.to_torch_params(sd, torch.float32), noise.cpu(), xfp.cpu(), xfo.cpu(), length, S, clip_denoised=True,
                                 eps_model=True, no_eff=True, eta=0.3, step_noise=z.cpu())
    e_eta = rel_l2(out, ref)
    monkeypatch.setenv("DC_ALLOW_EPSILON_NO_EFF_ETA0", "1")
    out0 = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref0 = O.ddim_sample_loop(O.to_torch_params(sd, torch.float32), noise.cpu(), xfp.cpu(), xfo.cpu(), length, S, clip_denoised=True,
                                  eps_model=True, no_eff=True)
    print(f"EPSILON, no_eff, fp16: eta = 0.3: {e_eta:.3e}   eta = 0 (refused by default; forced here): {rel_l2(out0, ref0):.3e}")
    assert torch.isfinite(out).all() and e_eta <= TOL


@pytest.mark.parametrize("prec,no_eff", [("fp16", False), ("bf16", False)])
def test_epsilon_model_stepped_through_single_evaluations(prec, no_eff):
    """ddim_sample_loop_progressive of an EPSILON model at eta = 0: the generator evaluates the native denoiser on split operands
    (dc_sampler_set_precise_forward) for exactly such loops - on plain fp16 operands it ended at 1.5e-3."""
    from helpers import O
    from diffusion_conductor_amd import MotionTransformer
    sd, B, T, S, length, xfp, xfo, noise, _ = _g9_setup()
    gd = _diffusion(S, "EPSILON")
    with torch.no_grad():
        ref = O.ddim_sample_loop(O.to_torch_params(sd, torch.float32), noise.cpu(), xfp.cpu(), xfo.cpu(), length, S, clip_denoised=True,
                                 eps_model=True, no_eff=no_eff)
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True, no_eff=no_eff, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    last = None
    for last in gd.ddim_sample_loop_progressive(m, (B, T, 26), noise=noise, clip_denoised=True, progress=False,
                                                model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)}):
        pass
    torch.cuda.synchronize()
    err = rel_l2(last["sample"], ref)
    print(f"EPSILON, eta = 0, stepped through forward(), {prec}{', no_eff' if no_eff else ''}: {err:.3e}")
    assert m.precise_forward is False and err <= TOL


@pytest.mark.parametrize("tag", ["clip", "eta", "eps"])
def test_sampler_branches_no_eff_vs_oracle(tag):
    """The same three branches through the full-attention kernels (k_layer_full carries the same fused update): against the
    oracle's loop on the same draws (the oracle is pinned to the reference on these branches by G9 and on no_eff by G3 / G6b)."""
    from diffusion_conductor_amd import MotionTransformer
    from helpers import O
    sd, B, T, S, length, xfp, xfo, noise, z = _g9_setup()
    clip, eta, mt = _BRANCH[tag]
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True,
                          precision="fp16", no_eff=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    gd = _diffusion(S, mt)
    res = gd.ddim_sample_loop(m, (B, T, 26), noise=noise, clip_denoised=clip, progress=False, eta=eta,
                              model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)}, step_noise=z)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.ddim_sample_loop(O.to_torch_params(sd, torch.float32), noise.cpu(), xfp.cpu(), xfo.cpu(), length, S, no_eff=True,
                                 eta=eta, clip_denoised=clip, eps_model=(mt == "EPSILON"), step_noise=z.cpu())
    err = rel_l2(res, ref)
    print(f"no_eff sampler branch [{tag}] rel-L2 vs oracle {err:.3e}")
    assert torch.isfinite(res).all() and err <= TOL


@pytest.mark.parametrize("tag", ["clip", "eta", "eps"])
def test_sampler_branches_progressive_path_g9(tag):
    """The generator form (one native denoiser call per step, the update on the host side of the ABI) yields the same samples
    and the reference's pred_xstart."""
    g = golden("g9_sampler_branches.npz")
    sd, B, T, S, length, xfp, xfo, noise, z = _g9_setup()
    clip, eta, mt = _BRANCH[tag]
    m = _model(sd, "fp16")
    gd = _diffusion(S, mt)
    outs = list(gd.ddim_sample_loop_progressive(m, (B, T, 26), noise=noise, clip_denoised=clip, progress=False, eta=eta,
                                                model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)},
                                                step_noise=z))
    assert len(outs) == S
    e = {"idx0": rel_l2(outs[0]["sample"], g[f"{tag}_idx0"]), "idx24": rel_l2(outs[24]["sample"], g[f"{tag}_idx24"]),
         "final": rel_l2(outs[-1]["sample"], g[f"{tag}_idx{S}"])}
    e.update({f"pred{it}": rel_l2(outs[it]["pred_xstart"], g[f"{tag}_pred{it}"]) for it in (0, 24, 49)})
    print(f"g9[{tag}] progressive rel-L2 " + "  ".join(f"{k} {v:.3e}" for k, v in e.items()))
    if mt == "EPSILON":
        # pred_xstart = sqrt(1/abar_t) x_t - sqrt(1/abar_t - 1) eps multiplies the denoiser's error by sqrt(1/abar_t - 1): 148 at
        # t = 49, 4.6 at t = 25, 0.03 at t = 0 (DDIM-50) - an early pred_xstart of an epsilon model is not a quantity 11-bit
        # operands can hold to 1e-3 (the SAMPLES are: idx0 above); gate it where the factor is below 1
        e = {k: v for k, v in e.items() if k not in ("pred0", "pred24")}
    assert max(e.values()) <= TOL


def test_eta_needs_noise_and_default_draw_is_finite():
    """eta > 0 without step_noise draws its own noise (as the reference does: the library generates it step by step from a seed);
    the C ABI refuses sigma != 0 with neither a tensor nor a seed."""
    from diffusion_conductor_amd import native
    sd, B, T, S, length, xfp, xfo, noise, z = _g9_setup()
    m = _model(sd, "fp16")
    gd = _diffusion(S, "START_X")
    nat = m.set_conditioning(xfp, xfo, length)
    with pytest.raises(native.DcError, match="needs the per-iteration noise"):
        nat.ddim_loop(noise, gd.native_coefficients(0.5), (), 0, None)              # fresh sampler: no seed was ever set
    out = gd.ddim_sample_loop(m, (B, T, 26), noise=noise, clip_denoised=True, progress=False, eta=1.0,
                              model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)})
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0 + 1e-6          # t = 0: x0 = clamp(pred)


def test_nonfinite_flag_and_auto_fallback():
    """A checkpoint whose activations leave the fp16 range: precision="fp16" reports it (FloatingPointError naming the remedy),
    precision="auto" re-runs the loop in a fresh "mixed" sampler and matches the oracle; the status word clears."""
    from diffusion_conductor_amd import native
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    sd = dict(sd)
    # FFN hidden layer scaled far beyond fp16: GELU(W1 h) ~ 1e6 enters the second FFN GEMM as an fp16 operand -> inf
    for l in range(8):
        k = f"temporal_decoder_blocks.{l}.ffn.linear1.weight"
        sd[k] = (sd[k] * 3.0e5).astype(np.float32)
        k2 = f"temporal_decoder_blocks.{l}.ffn.linear2.weight"
        sd[k2] = (sd[k2] / 3.0e5).astype(np.float32)
    B, T, S = 2, 96, 25
    xfp, xfo = _features(sd, B, T, 50)
    noise = torch.from_numpy(batch_noise(B, T, first=50))
    p = O.to_torch_params(sd)
    with torch.no_grad():
        ref = O.ddim_sample_loop(p, noise, xfp, xfo, [96, 70], S)
    kw = dict(noise=noise.cuda(), clip_denoised=False, progress=False,
              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor([96, 70])})
    gd = make_diffusion(S)
    m16 = _model(sd, "fp16")
    with pytest.raises(FloatingPointError, match="precision='mixed'"):
        gd.ddim_sample_loop(m16, (B, T, 26), **kw)
    assert m16._native.status() == 0                       # the failing call read and cleared the word
    ma = _model(sd, "auto")
    assert ma.active_precision == "fp16"
    out = gd.ddim_sample_loop(ma, (B, T, 26), **kw)
    torch.cuda.synchronize()
    err = rel_l2(out, ref)
    print(f"auto fallback: active precision {ma.active_precision}, rel-L2 vs oracle {err:.3e}")
    assert ma.active_precision == "mixed" and torch.isfinite(out).all() and err <= TOL
    assert ma._native.status() & native.STATUS_NONFINITE == 0


def test_status_word_reports_on_the_loop_that_was_asked_about():
    """The device status word is reset when a loop starts: a non-finite bit left behind by work nobody checked (a loop with
    check_numerics off, a forward()) must not fail - or switch the precision of - the next, healthy loop on the same sampler."""
    from diffusion_conductor_amd import native
    good = synthetic_state_dict(DenoiserConfig(), seed=0)
    bad = dict(good)
    for l in range(8):
        k = f"temporal_decoder_blocks.{l}.ffn.linear1.weight"
        bad[k] = (bad[k] * 3.0e5).astype(np.float32)
    B, T, S = 2, 96, 25
    xfp, xfo = _features(good, B, T, 50)
    noise = torch.from_numpy(batch_noise(B, T, first=50)).cuda()
    kw = dict(noise=noise, clip_denoised=False, progress=False,
              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor([96, 70])})
    gd = make_diffusion(S)
    m = _model(bad, "auto")
    m.check_numerics = False
    out = gd.ddim_sample_loop(m, (B, T, 26), **kw)                    # leaves DC_STATUS_NONFINITE behind, unread
    m(noise, torch.tensor([3, 3]), length=torch.LongTensor([96, 70]), xf_proj=kw["model_kwargs"]["xf_proj"], xf_out=kw["model_kwargs"]["xf_out"])
    torch.cuda.synchronize()
    assert not torch.isfinite(out).all()
    nat = m._native
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in good.items()}, strict=True)    # same sampler object, healthy weights
    m.check_numerics = True
    out = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    torch.cuda.synchronize()
    assert m._native is nat and m.active_precision == "fp16" and torch.isfinite(out).all()
    assert not m.numerics_fallback(native.STATUS_NONFINITE | 4) and not m.numerics_fallback(4)     # only for exactly NONFINITE
    assert m.active_precision == "fp16"


def test_seeded_step_noise_equals_the_explicit_tensor():
    """eta > 0 without a noise tensor: the library draws each iteration's N(0,1) noise at the head of its step from a seed
    (one [B,T,P] buffer, dc_sampler_set_step_noise_seed) - the same result, bit for bit, as handing it the [S,B,T,P] tensor
    assembled from dc_step_noise_fill; the draws are standard normal, differ per iteration and per seed, and a second
    tensor at another address replays the captured graph (the address travels through a device slot)."""
    from diffusion_conductor_amd import native
    sd, B, T, S, length, xfp, xfo, noise, _ = _g9_setup()
    m = _model(sd, "fp16")
    gd = make_diffusion(S)
    nat = m.set_conditioning(xfp, xfo, length)
    coef = gd.native_coefficients(0.5)
    seeded, _ = nat.ddim_loop(noise, coef, noise_seed=1234)
    z = torch.stack([native.step_noise((B, T, 26), 1234, i, noise.device) for i in range(S)])
    explicit, _ = nat.ddim_loop(noise, coef, step_noise=z)
    z2 = z.clone()
    explicit2, _ = nat.ddim_loop(noise, coef, step_noise=z2)
    other, _ = nat.ddim_loop(noise, coef, noise_seed=1235)
    torch.cuda.synchronize()
    assert nat.status() == 0
    assert torch.equal(seeded, explicit) and torch.equal(explicit, explicit2) and not torch.equal(seeded, other)
    zc = z.double().cpu()
    print(f"library draws: mean {zc.mean():.4f} std {zc.std():.4f} kurtosis {((zc - zc.mean()) ** 4).mean() / zc.var() ** 2:.3f}")
    assert abs(float(zc.mean())) < 0.01 and abs(float(zc.std()) - 1.0) < 0.01 and abs(float(((zc - zc.mean()) ** 4).mean() / zc.var() ** 2) - 3.0) < 0.1
    assert not torch.equal(z[0], z[1]) and abs(float((zc[0] * zc[1]).mean())) < 0.02
    # a seed serves one loop: without a new one (and without a tensor) the next eta > 0 loop fails instead of replaying the draws
    with pytest.raises(native.DcError, match="needs the per-iteration noise"):
        nat.ddim_loop(noise, coef)
    # a shard of a larger batch draws the rows the whole batch's draw gives its clips (identically seeded ranks)
    if B >= 2:
        lo = 1
        nat1 = m.set_conditioning(xfp[lo:], xfo[lo:], length[lo:])
        shard, _ = nat1.ddim_loop(noise[lo:].contiguous(), coef, noise_seed=(1234, lo * T * 26))
        same_rows, _ = nat1.ddim_loop(noise[lo:].contiguous(), coef, noise_seed=1234)
        torch.cuda.synchronize()
        assert torch.equal(shard, seeded[lo:]) and not torch.equal(same_rows, seeded[lo:])
        nat = m.set_conditioning(xfp, xfo, length)
    # through the Python surface: torch.manual_seed makes an eta > 0 run reproducible
    kw = dict(noise=noise, clip_denoised=False, progress=False, eta=0.5,
              model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)})
    torch.manual_seed(7)
    a = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    torch.manual_seed(7)
    b = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    c = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(c).all()


def test_smoothing_window_longer_than_the_clip_fails_before_any_work():
    """The window check of the loop's final write runs before anything is enqueued: the call fails cleanly and the next one works."""
    from diffusion_conductor_amd import native
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    B, T, S = 1, 32, 25
    xfp, xfo = _features(sd, B, T, 50)
    noise = torch.from_numpy(batch_noise(B, T, first=50)).cuda()
    m = _model(sd, "fp16")
    gd = make_diffusion(S)
    kw = dict(noise=noise, clip_denoised=False, progress=False,
              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor([32])})
    with pytest.raises(native.DcError, match="smoothing window"):
        gd.ddim_sample_loop(m, (B, T, 26), smooth=(51, 5), **kw)
    a = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    b = gd.ddim_sample_loop(m, (B, T, 26), smooth=(19, 5), **kw)
    c = gd.ddim_sample_loop(m, (B, T, 26), smooth=(19, 5), **kw)          # same (window, order): no table upload, same result
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.equal(b, c) and torch.equal(b, native.savgol_filter(a, 19, 5))


def test_smoothing_in_the_loops_final_write():
    """dc_sampler_set_smoothing: the loop's final write applies the Savitzky-Golay filter (tools/visualization.py:20-26, 126)
    in place of the plain copy - the same numbers as filtering the unsmoothed result afterwards (bit for bit: one kernel on the
    same x0), scipy within 1e-5; snapshots stay unsmoothed; window 0 switches it off."""
    from scipy.signal import savgol_filter
    from diffusion_conductor_amd import native
    sd, B, T, S, length, xfp, xfo, noise, z = _g9_setup()
    m = _model(sd, "fp16")
    gd = make_diffusion(S)
    kw = dict(noise=noise, clip_denoised=False, progress=False, idxs=[10],
              model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(length)})
    plain = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    sm = gd.ddim_sample_loop(m, (B, T, 26), smooth=(19, 5), **kw)
    again = gd.ddim_sample_loop(m, (B, T, 26), **kw)
    torch.cuda.synchronize()
    assert torch.equal(sm[10], plain[10]) and torch.equal(again[S], plain[S])
    assert torch.equal(sm[S], native.savgol_filter(plain[S], 19, 5))
    ref = savgol_filter(plain[S].cpu().numpy().astype(np.float64), 19, 5, axis=1)
    err = rel_l2(sm[S], ref)
    print(f"in-loop smoothing vs scipy: rel-L2 {err:.3e}")
    assert err <= 1e-5 and rel_l2(sm[S], plain[S]) > 1e-3
    with pytest.raises(native.DcError, match="window"):
        m._native.set_smoothing(18, 5)


def test_film_tile_saturation_is_diagnosed():
    """FiLM modulation values beyond the fp16 storage range (emb_layers x 1e6): x0 turns non-finite, dc_sampler_status scans the
    tiles and adds the F16_SATURATED bit, and precision="auto" does NOT retry (no mode stores those tiles wider) - it raises."""
    from diffusion_conductor_amd import native
    sd = dict(synthetic_state_dict(DenoiserConfig(), seed=0))
    for k in list(sd):
        if ".emb_layers.1.weight" in k:
            sd[k] = (sd[k] * 1.0e6).astype(np.float32)
    B, T, S = 2, 96, 25
    xfp, xfo = _features(sd, B, T, 50)
    noise = torch.from_numpy(batch_noise(B, T, first=50)).cuda()
    m = _model(sd, "auto")
    with pytest.raises(FloatingPointError, match="every precision mode stores"):
        make_diffusion(S).ddim_sample_loop(m, (B, T, 26), noise=noise, clip_denoised=False, progress=False,
                                           model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor([96, 70])})
    assert m.active_precision == "fp16"
    assert m._native.status() == 0


def test_encode_music_from_a_pinned_host_batch_is_pipelined_and_identical():
    """A pinned host batch is copied in chunks beside the encoder (denoiser._encode_music_pipelined): same numbers as the
    encode of the batch already on the device, bit for bit (per-clip kernels see the same inputs)."""
    from helpers import batch_mel
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    m = _model(sd, "fp16")
    mel = torch.from_numpy(batch_mel(19, 540))                  # a first chunk of 8 clips + the other 11
    xp_d, x_d = m.encode_music(mel.cuda(), "cuda:0")
    xp_h, x_h = m.encode_music(mel.pin_memory(), "cuda:0")
    torch.cuda.synchronize()
    assert tuple(x_h.shape) == (19, 180, 64) and torch.equal(xp_h, xp_d) and torch.equal(x_h, x_d)
    p = O.to_torch_params(sd)
    with torch.no_grad():
        rxp, rx = O.encode_music(p, mel[16:19])
    assert rel_l2(x_h[16:19], rx) <= 6e-4 and rel_l2(xp_h[16:19], rxp) <= 6e-4      # (fp16 model: one fp16 plane per activation, measured 3.7e-4)


@pytest.mark.parametrize("B,T", [(1, 257), (3, 1000), (2, 1799), (4, 1800), (33, 300), (5, 77)])
def test_mixed_mode_clip_aligned_units_edge_shapes(B, T):
    """The split-bf16 layer path on workgroup records runs clip-aligned 8-wave units on a padded clip stride (T >= 256; below that
    the per-group form): strides that need padding, ragged lengths down to one frame, more clips than the fused embedding launch
    carries (33 x 300 -> 66 units), per-clip timesteps.  One forward and a DDIM-6 loop with an intermediate, vs the oracle."""
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    p = O.to_torch_params(sd)
    m = _model(sd, "mixed")
    xfp, xfo = _features(sd, B, T, 7)
    x = torch.from_numpy(batch_noise(B, T, first=7))
    t = torch.tensor([(131 * b + 5) % 1000 for b in range(B)])
    length = [T if b % 3 == 0 else (1 if b % 3 == 1 else max(1, T - 33 * b - 1)) for b in range(B)]
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo)
    out = m(x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
    torch.cuda.synchronize()
    e_fwd = rel_l2(out, ref)
    S = 25
    with torch.no_grad():
        rl = O.ddim_sample_loop(p, x, xfp, xfo, length, S, idxs=(3,))
    gd = make_diffusion(S)
    res = gd.ddim_sample_loop(m, (B, T, 26), noise=x.cuda(), clip_denoised=False, progress=False, idxs=[3],
                              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
    torch.cuda.synchronize()
    e3, eS = rel_l2(res[3], rl[3]), rel_l2(res[S], rl[S])
    print(f"mixed B={B} T={T} (clip stride {m._native.clip_stride()}): forward {e_fwd:.2e}  ddim-{S} idx3 {e3:.2e} final {eS:.2e}")
    assert torch.isfinite(res[S]).all() and max(e_fwd, e3, eS) <= 2e-4
    again = gd.ddim_sample_loop(m, (B, T, 26), noise=x.cuda(), clip_denoised=False, progress=False, idxs=[3],
                                model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
    assert torch.equal(again[S], res[S]) and torch.equal(again[3], res[3])


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_last_layer_does_the_next_steps_front_work(prec, monkeypatch):
    """Round 6: in the wide plain form the last layer of a step embeds x_{t-1} and runs layer 0's self-attention front half for the next
    step (k_layer, DC_UPD_EMBED_NEXT; dc_api.hip, enqueue_step) - the next FiLM launch is the bare GEMM.  DC_NO_EMBED_NEXT=1 keeps the
    front work in every step's own launch (round 5's form).  Both against the oracle and against each other, on the branches the fused
    tail re-implements: ragged lengths down to one frame, a padded and an unpadded clip stride, eta > 0 with seeded step noise,
    clip_denoised, an EPSILON-free START_X model with snapshots of intermediate iterations (captured graph and eager launches)."""
    monkeypatch.setenv("DC_NO_NARROW", "1")            # the wide 8-wave form at a batch the oracle finishes in seconds
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    p = O.to_torch_params(sd)
    m = _model(sd, prec)
    S = 25
    gd = make_diffusion(S)
    for B, T, eta, clip in [(3, 300, 0.0, False), (2, 777, 0.5, True), (5, 256, 0.0, False)]:
        xfp, xfo = _features(sd, B, T, 21)
        x = torch.from_numpy(batch_noise(B, T, first=21))
        length = [T, 1, max(1, T - 37), T // 2, 2][:B]
        z = torch.from_numpy(batch_step_noise(S, B, T, first=21)) if eta > 0 else None
        with torch.no_grad():
            ref = O.ddim_sample_loop(p, x, xfp, xfo, length, S, idxs=(5, S - 2), eta=eta, clip_denoised=clip, step_noise=z)
        kw = dict(noise=x.cuda(), clip_denoised=clip, progress=False, idxs=[5, S - 2], eta=eta,
                  model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
        if z is not None:
            kw["step_noise"] = z.cuda()
        outs = {}
        for form in ("fused", "own"):
            for graph in (True, False):
                if form == "own":
                    monkeypatch.setenv("DC_NO_EMBED_NEXT", "1")
                else:
                    monkeypatch.delenv("DC_NO_EMBED_NEXT", raising=False)
                if graph:
                    monkeypatch.delenv("DC_DISABLE_GRAPH", raising=False)
                else:
                    monkeypatch.setenv("DC_DISABLE_GRAPH", "1")
                res = gd.ddim_sample_loop(m, (B, T, 26), **kw)
                torch.cuda.synchronize()
                outs[(form, graph)] = res
                for k in (5, S - 2, S):
                    assert torch.isfinite(res[k]).all()
                    e = max(rel_l2(res[k][c:c + 1], ref[k][c:c + 1]) for c in range(B))
                    assert e <= TOL, (prec, B, T, form, graph, k, e)
        for k in (5, S - 2, S):
            assert torch.equal(outs[("fused", True)][k], outs[("fused", False)][k])        # captured == eager, either form
            assert torch.equal(outs[("own", True)][k], outs[("own", False)][k])
            d = rel_l2(outs[("fused", True)][k], outs[("own", True)][k])
            print(f"{prec} B={B} T={T} eta={eta}: iteration {k}: fused vs own front work {d:.2e}")
            assert d <= 5e-4, d
        assert not torch.equal(outs[("fused", True)][S], outs[("own", True)][S])            # (the switch does switch)
    monkeypatch.delenv("DC_NO_EMBED_NEXT", raising=False)
    monkeypatch.delenv("DC_DISABLE_GRAPH", raising=False)


def test_bf16_precision_short_ragged_clips_every_clip_inside_the_bound():
    """Found by tools/fuzz_shapes.py (seed 601, case 117: 39 clips of 36 frames, lengths down to 1, DDIM-50): the worst clip read 1.27e-3 in
    the bf16 precision with its default tail of 6 - a clip's error is then a norm over a few hundred numbers and the worst of dozens of clips
    left the bound (8.6e-4 at 39 frames, <= 7.2e-4 from 100 frames up).  Loops of that precision over clips of fewer than 100 frames run
    every evaluation in the split form (loop_common in dc_api.hip): every clip of three such batches at the split level, and a 100-frame
    batch keeps the tail (the two settings give different results there)."""
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    p = O.to_torch_params(sd)
    m = _model(sd, "bf16")
    gd = make_diffusion(50)
    rng = np.random.default_rng(117)
    worst = 0.0
    for B, T, first in [(39, 36, 11), (32, 39, 64), (29, 67, 150)]:
        xfp, xfo = _features(sd, B, T, first)
        x = torch.from_numpy(batch_noise(B, T, first=first))
        length = [int(rng.integers(1, T + 1)) if rng.random() < 0.6 else T for _ in range(B)]
        length[0] = 1
        with torch.no_grad():
            ref = O.ddim_sample_loop(p, x, xfp, xfo, length, 50)
        out = gd.ddim_sample_loop(m, (B, T, 26), noise=x.cuda(), clip_denoised=False, progress=False,
                                  model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
        torch.cuda.synchronize()
        errs = [rel_l2(out[c:c + 1], ref[c:c + 1]) for c in range(B)]
        print(f"bf16 B={B} T={T}: worst clip {max(errs):.2e}, whole batch {rel_l2(out, ref):.2e}")
        assert torch.isfinite(out).all()
        worst = max(worst, max(errs))
    assert worst <= 3e-4, worst          # the split form's level (9e-5 on the fuzz runs), far inside 1e-3
    B, T = 4, 100                        # from 100 frames on the precision keeps its plain evaluations + tail
    xfp, xfo = _features(sd, B, T, 5)
    x = torch.from_numpy(batch_noise(B, T, first=5))
    kw = {"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor([T] * B)}
    a = gd.ddim_sample_loop(m, (B, T, 26), noise=x.cuda(), clip_denoised=False, progress=False, model_kwargs=kw)
    m._native.set_precise_tail(50)
    b = gd.ddim_sample_loop(m, (B, T, 26), noise=x.cuda(), clip_denoised=False, progress=False, model_kwargs=kw)
    m._native.set_precise_tail(-1)
    torch.cuda.synchronize()
    assert not torch.equal(a, b)


def test_fp16_plane_encoder_overflow_is_resampled_on_split_planes():
    """The fp16-plane MusicEncoder holds activations up to 65504.  A checkpoint whose encoder runs at 3e5 times the usual scale
    inside (conv1.0 scaled up, conv4 scaled down by the same factor: ReLU and the max-pools are positively homogeneous, so its output
    stays of order one) overflows the planes; the loop's numeric check fires, the harness finds the music features non-finite,
    encodes again on the split bf16 planes (fp32 range) and samples again - the caller gets what the oracle computes for that
    checkpoint."""
    import types
    from helpers import batch_mel
    from diffusion_conductor_amd import DDPMTrainer
    sd = dict(synthetic_state_dict(DenoiserConfig(), seed=0))
    k = np.float32(3e5)
    for n in ("music_encoder.conv1.0.conv2d_layer.0.weight", "music_encoder.conv1.0.conv2d_layer.0.bias"):
        sd[n] = sd[n] * k
    sd["music_encoder.conv4.0.weight"] = sd["music_encoder.conv4.0.weight"] / k
    m = _model(sd, "fp16")
    S = 25
    tr = DDPMTrainer(types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=S, is_train=False), m)
    tr.eval_mode()
    mel = batch_mel(1, 540)
    xo = m.encode_music(torch.from_numpy(mel).cuda(), "cuda:0")[1]
    assert not bool(torch.isfinite(xo).all()), "the checkpoint was meant to overflow the fp16 planes"
    noise = torch.from_numpy(batch_noise(1, 180, first=7))
    out = tr.generate_music_motion(mel[0], 26, noise=noise)
    torch.cuda.synchronize()
    assert m.encoder_format == "split" and bool(torch.isfinite(out).all())
    with torch.no_grad():
        ref = O.generate_music_motion(O.to_torch_params(sd), torch.from_numpy(mel), 26, S, noise)
    err = rel_l2(out, ref)
    print(f"overflowing encoder, re-encoded on split planes: rel-L2 {err:.3e}")
    assert err <= TOL


@pytest.mark.parametrize("prec", ["fp16", "mixed"])
@pytest.mark.parametrize("tag,mean,var,cond,clip", [("mt_cond", "START_X", "FIXED_SMALL", True, False),
                                                    ("mt_prevx", "PREVIOUS_X", "FIXED_SMALL", False, True),
                                                    ("mt_learned", "START_X", "LEARNED_RANGE", False, False)])
def test_sampler_off_path_branches_around_the_native_denoiser_g11(tag, mean, var, cond, clip, prec):
    """cond_fn / condition_score (gaussian_diffusion.py:581-603, 806-808), ModelMeanType.PREVIOUS_X (:510-514, 545) and a learned-variance
    (2C-channel) model (:472-486) through the step-through path: the host arithmetic of sampler.py around the HIP denoiser's model call,
    against the imported reference's DDIM-50 output with the reference denoiser (fixture G11, B=2, T=96, ragged)."""
    from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule
    from oracle import toy_models as TM
    g = golden("g11_offpath_branches.npz")
    sd = synthetic_state_dict(DenoiserConfig(), seed=0)
    B, T, S = 2, 96, 50
    xfp, xfo = (t.cuda() for t in _features(sd, B, T, 60))
    noise = torch.from_numpy(batch_noise(B, T, first=60)).cuda()
    m = _model(sd, prec)
    gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=getattr(ModelMeanType, mean),
                           model_var_type=getattr(ModelVarType, var), loss_type=LossType.MSE)

    def learned_wrap(x, t, **kw):
        y = m(x, t, **kw)
        return torch.cat([y, torch.zeros_like(y)], dim=1)

    mdl = learned_wrap if var.startswith("LEARNED") else m
    outs = list(gd.ddim_sample_loop_progressive(mdl, (B, T, 26), noise=noise, clip_denoised=clip, progress=False, eta=0.0, device="cuda",
                                                cond_fn=TM.pose_cond_fn if cond else None,
                                                model_kwargs={"xf_proj": xfp, "xf_out": xfo, "length": torch.LongTensor(g["mt_length"])}))
    torch.cuda.synchronize()
    e_final, e_mid = rel_l2(outs[-1]["sample"], g[f"{tag}_final"]), rel_l2(outs[24]["pred_xstart"], g[f"{tag}_pred24"])
    print(f"g11[{tag}][{prec}] rel-L2 final {e_final:.3e}  pred_xstart@24 {e_mid:.3e}")
    assert len(outs) == S and torch.isfinite(outs[-1]["sample"]).all() and max(e_final, e_mid) <= TOL
