"""Parity of the HIP path (through the C ABI) with the oracle / golden fixtures.  Needs an MI355X.

Tolerances (rel-L2 on x0, ||gpu - ref||_2 / ||ref||_2):
  * TOL_PARITY = 1e-3  - the bound BASELINE.json's north_star states; gates the default "fp16" mode and "mixed"
  * TOL_X3     = 1e-4  - validation mode (split-bf16 everywhere): catches any indexing/layout error;
                         what remains is the fp16 storage of the FiLM outputs and fast-math exp
  * plain "bf16" is reported and only bounded loosely (it is known not to meet 1e-3: SURVEY.md section 7)
"""
import os

import numpy as np
import pytest
import torch

from helpers import (O, batch_mel, batch_noise, golden, make_diffusion, make_model, oracle_params, rel_l2, xf_pair)

pytestmark = pytest.mark.gpu

TOL_PARITY = 1e-3
TOL_X3 = 1e-4
TOL_BF16 = 2e-2


@pytest.fixture(scope="module")
def models():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return {p: make_model(p) for p in ("fp16", "mixed", "bf16x3", "bf16")}


def _ddim(model, S, noise, xfp, xfo, length, idxs=()):
    gd = make_diffusion(S)
    B, T, P = noise.shape
    out = gd.ddim_sample_loop(model, (B, T, P), noise=noise.cuda(), clip_denoised=False, progress=False,
                              model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(),
                                            "length": torch.LongTensor(list(length))}, idxs=list(idxs))
    torch.cuda.synchronize()
    return out


def test_native_library_is_loaded(models):
    """The product path must be the HIP library, not an eager fallback."""
    from diffusion_conductor_amd import native
    assert native.lib().dc_kernel_count() >= 5
    assert "libdc_ddim.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("prec,tol", [("bf16x3", TOL_X3), ("fp16", TOL_PARITY), ("mixed", TOL_PARITY), ("bf16", TOL_BF16)])
def test_forward_golden_blocks(models, prec, tol):
    """G3: one MotionTransformer.forward at B=2, T=64, ragged length, per-clip timesteps."""
    g = golden("g3_blocks.npz")
    m = models[prec]
    out = m(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]), length=torch.from_numpy(g["length"]),
            xf_proj=torch.from_numpy(g["xf_proj"]).cuda(), xf_out=torch.from_numpy(g["xf_out"]).cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, g["forward"])
    print(f"forward[{prec}] rel-L2 {err:.3e}")
    assert torch.isfinite(out).all() and err <= tol


def test_forward_straddling_groups(models):
    """T not a multiple of 32: token groups straddle clip boundaries (two partial records, masked
    two-pass attention apply).  Compared with the oracle directly."""
    B, T = 5, 77
    p = oracle_params()
    xfp, xfo = xf_pair(B, T, first=30)
    x = torch.from_numpy(batch_noise(B, T, first=30))
    t = torch.tensor([0, 49, 13, 999, 500])
    length = [77, 1, 40, 76, 33]
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo)
    out = models["bf16x3"](x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, ref)
    print(f"straddle rel-L2 {err:.3e}")
    assert err <= TOL_X3


@pytest.mark.parametrize("B,T", [(1, 257), (2, 288), (3, 1000), (2, 1799), (9, 1800), (33, 300)])
def test_forward_padded_clip_strides(models, B, T):
    """Clip strides at the edges of the padding policy (dc_api.hip clip_stride): 257 -> 288 (+12 %, small batch only), 288 and
    1800-multiples untouched or padded by < 2 %, a batch just past the narrow-workgroup limit (9 x 1800: 135 narrow units) and
    one that fills the chip with short clips; ragged lengths down to one frame; per-clip timesteps.  Forward vs the oracle."""
    p = oracle_params()
    xfp, xfo = xf_pair(B, T, first=7)
    x = torch.from_numpy(batch_noise(B, T, first=7))
    t = torch.tensor([(131 * b + 5) % 1000 for b in range(B)])
    length = [T if b % 3 == 0 else (1 if b % 3 == 1 else max(1, T - 33 * b - 1)) for b in range(B)]
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo)
    out = models["fp16"](x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, ref)
    print(f"padded stride B={B} T={T}: rel-L2 {err:.3e}")
    assert torch.isfinite(out).all() and err <= TOL_PARITY


@pytest.mark.parametrize("B,T", [(33, 300), (20, 1000), (40, 777), (18, 1800), (70, 511)])
def test_ddim25_odd_shapes_wide_units_vs_oracle(models, B, T):
    """Whole DDIM-25 loops at shapes off the benchmark's grid, with ragged lengths down to one frame, against the oracle: batches
    past the narrow-workgroup limit (flat 256-token units that contain clip edges at every offset; 777 and 1000 are padded to a
    multiple of 32, 300 and 511 are not), 18 x 1800 (the first batch size on wide units); f16 and the bf16-MFMA mode."""
    S = 25
    xfp, xfo = xf_pair(B, T, first=80)
    noise = torch.from_numpy(batch_noise(B, T, first=80))
    length = [T if b % 3 == 0 else (1 if b % 7 == 1 else max(1, T - 29 * b - 1)) for b in range(B)]
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S)
    for mode in ("fp16", "mixed"):
        a = _ddim(models[mode], S, noise, xfp, xfo, length)
        errs = [rel_l2(a[c:c + 1], ref[c:c + 1]) for c in range(B)]
        print(f"ddim25 B={B} T={T} ({mode}): whole batch {rel_l2(a, ref):.3e}, worst clip {max(errs):.3e} (clip {int(np.argmax(errs))}, length {length[int(np.argmax(errs))]})")
        assert torch.isfinite(a).all() and max(errs) <= TOL_PARITY, errs


@pytest.mark.parametrize("prec,tol", [("fp16", TOL_PARITY), ("mixed", TOL_PARITY), ("bf16x3", TOL_X3), ("bf16", TOL_PARITY)])      # (bf16: with its precise tail of 6)
def test_ddim50_config1_golden(models, prec, tol):
    """G5 = BASELINE config 1: single 60 s clip, DDIM-50, with the idxs=[0,24] intermediates."""
    g = golden("g5_ddim50_b1.npz")
    xfp, xfo = xf_pair(1, 1800)
    noise = torch.from_numpy(batch_noise(1, 1800))
    res = _ddim(models[prec], 50, noise, xfp, xfo, [1800], idxs=(0, 24))
    assert set(res.keys()) == {0, 24, 50}
    err = rel_l2(res[50], g["x0"])
    e0 = rel_l2(res[0].cpu()[:, ::20], g["idx0_sub"])
    e24 = rel_l2(res[24].cpu()[:, ::20], g["idx24_sub"])
    rms = float(np.sqrt(np.mean((res[50].cpu().numpy() - g["x0"]) ** 2)))
    print(f"ddim50[{prec}] x0 rel-L2 {err:.3e} (rms-abs {rms:.2e})  idx0 {e0:.2e}  idx24 {e24:.2e}")
    assert err <= tol and e0 <= tol and e24 <= max(tol, TOL_BF16 if prec == "bf16" else 0.0)       # (iteration 24 of the bf16 loop is plain bf16)


def test_precise_tail_halves_the_fp16_error(models):
    """fp16 precision: the loop's last model evaluation(s) run on split fp16 operands (include/dc_ddim.h, dc_sampler_set_precise_tail;
    default 1).  DDIM's last step returns the model's own prediction of x0, and what the fp16 mode loses is mostly the weights' rounding
    there: golden DDIM-50 (G5) with 0 / 1 / 2 split evaluations, eager and captured loops alike."""
    g = golden("g5_ddim50_b1.npz")
    xfp, xfo = xf_pair(1, 1800)
    noise = torch.from_numpy(batch_noise(1, 1800))
    assert "DC_PRECISE_TAIL" not in os.environ
    errs = {}
    for k in (0, 1, 2):
        errs[k] = rel_l2(_with_env({"DC_PRECISE_TAIL": str(k)}, lambda: _ddim(models["fp16"], 50, noise, xfp, xfo, [1800])), g["x0"])
    default = _ddim(models["fp16"], 50, noise, xfp, xfo, [1800])
    eager = _with_env({"DC_DISABLE_GRAPH": "1"}, lambda: _ddim(models["fp16"], 50, noise, xfp, xfo, [1800]))
    print("precise tail 0 / 1 / 2: " + " ".join(f"{errs[k]:.3e}" for k in (0, 1, 2)) + f"; default {rel_l2(default, g['x0']):.3e}")
    assert errs[0] <= TOL_PARITY and errs[1] <= 0.6 * errs[0] and errs[2] <= 0.8 * errs[1]
    assert rel_l2(default, g["x0"]) == errs[1] and torch.equal(default, eager)          # the default is one evaluation; graph == eager
    nat = models["fp16"]._ensure_native("cuda:0")
    try:
        nat.set_precise_tail(2)
        assert rel_l2(_ddim(models["fp16"], 50, noise, xfp, xfo, [1800]), g["x0"]) == errs[2]
        from diffusion_conductor_amd import native
        with pytest.raises(native.DcError, match="precise tail"):
            native._check(native.lib().dc_sampler_set_precise_tail(nat._h, -2))          # (-1: back to the default)
    finally:
        nat.set_precise_tail(1)
    # a clip stride that is not a whole number of 32-frame groups (here forced: DC_NO_PAD) has no clip-aligned units: the split evaluation
    # runs in the per-group record form with its combine launches
    g6 = golden("g6_variants.npz")
    xfp9, xfo9 = xf_pair(2, 900, first=10)
    noise9 = torch.from_numpy(batch_noise(2, 900, first=10))
    e9 = {k: rel_l2(_with_env({"DC_NO_PAD": "1", "DC_PRECISE_TAIL": k}, lambda: _ddim(models["fp16"], 50, noise9, xfp9, xfo9, [900, 700])),
                    g6["t900_x0"]) for k in ("0", "1")}
    print(f"T = 900 unpadded: precise tail 0 / 1: {e9['0']:.3e} {e9['1']:.3e}")
    assert e9["0"] <= TOL_PARITY and e9["1"] <= 0.7 * e9["0"]
    # the bf16 precision: plain bf16 operands cannot meet the bound (8 mantissa bits); its last evaluations in the "mixed" form (128-wide
    # GEMMs on split bf16, FiLM GEMM on f16 operands) can - 6 of 50 by default (4 left 1.1e-3 on one short ragged batch of
    # tools/fuzz_shapes.py, 6: 8.6e-4, 8: 7.3e-4 - profiles/r06_fuzz_bf16_tail.txt); round 5's form (bf16 FiLM operands in the tail too)
    # needed 8 for what 4 give now
    eb = {k: rel_l2(_with_env({"DC_PRECISE_TAIL": str(k)}, lambda: _ddim(models["bf16"], 50, noise, xfp, xfo, [1800])), g["x0"]) for k in (0, 2, 4, 8)}
    print("bf16, precise tail 0 / 2 / 4 / 8: " + " ".join(f"{eb[k]:.3e}" for k in (0, 2, 4, 8)))
    assert eb[0] > TOL_PARITY and eb[8] <= 0.5 * TOL_PARITY and eb[8] < eb[4] < eb[2] < eb[0]
    r5 = rel_l2(_with_env({"DC_PRECISE_TAIL": "8", "DC_TAIL_FILM_BF16": "1"}, lambda: _ddim(models["bf16"], 50, noise, xfp, xfo, [1800])), g["x0"])
    from diffusion_conductor_amd import native as _nat
    kd = _nat.precise_tail_default("bf16")
    ed = rel_l2(_ddim(models["bf16"], 50, noise, xfp, xfo, [1800]), g["x0"])          # its default
    print(f"bf16 default tail {kd}: {ed:.3e}; round 5's form (tail 8, bf16 FiLM operands in the tail): {r5:.3e}")
    assert kd == 6 and eb[8] < ed < eb[4] and ed <= 0.5 * TOL_PARITY and ed < r5


@pytest.mark.parametrize("S", [1, 25])
def test_precise_tail_in_short_loops(models, S):
    """A loop of fewer steps than the tail asks for runs every evaluation on split operands (the tail is clipped to the loop: S = 1 - the only
    length below bf16's default of 6 that the reference's linear schedule admits, betas <= 1 -; S = 25: nineteen plain evaluations in front
    of six split ones), on a short ragged batch (T = 96: the per-group record form) and on clip-aligned units (T = 320), against the
    oracle.  One all-split evaluation: fp16 6 - 7e-5, bf16 the same since round 6 (its split evaluations take the FiLM GEMM's operands in f16;
    round 5: 3.6 - 4.0e-4)."""
    for B, T, length in ((2, 96, [96, 61]), (2, 320, [320, 1])):
        xfp, xfo = xf_pair(B, T, first=40)
        noise = torch.from_numpy(batch_noise(B, T, first=40))
        with torch.no_grad():
            ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S)
        for mode in ("fp16", "bf16"):
            plain = rel_l2(_with_env({"DC_PRECISE_TAIL": "0"}, lambda: _ddim(models[mode], S, noise, xfp, xfo, length)), ref)
            a = _ddim(models[mode], S, noise, xfp, xfo, length)
            asked = rel_l2(_with_env({"DC_PRECISE_TAIL": "50"}, lambda: _ddim(models[mode], S, noise, xfp, xfo, length)), ref)
            err = rel_l2(a, ref)
            print(f"S={S} T={T} ({mode}): plain {plain:.3e}, default tail {err:.3e}, tail 50 {asked:.3e}")
            assert torch.isfinite(a).all() and err <= TOL_PARITY and asked <= TOL_PARITY
            assert err < plain
            if S == 1:
                assert err == asked and err <= 1.5e-4        # every evaluation split


@pytest.mark.parametrize("B,T", [(1, 1), (3, 2), (40, 7), (5, 31), (2, 33)])
def test_clips_shorter_than_one_token_group(models, B, T):
    """The reference takes any T; the token space works in 32-frame groups whose records name at most two clips, so clips of fewer than
    32 frames get a clip stride of one group (padding frames behave like frames past `length`).  Forward and a DDIM-25 loop against the
    oracle, ragged lengths; full attention (`no_eff`) refuses such clips loudly."""
    S = 25
    xfp, xfo = xf_pair(B, T, first=7)
    noise = torch.from_numpy(batch_noise(B, T, first=7))
    length = [max(1, T - (b % 3) * (T // 3)) for b in range(B)]
    t = torch.tensor([(37 * b + 11) % 1000 for b in range(B)], dtype=torch.long)
    with torch.no_grad():
        ref_f = O.denoiser_forward(oracle_params(), noise, t, length, xfp, xfo)
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S)
    for mode in ("fp16", "mixed", "bf16"):
        f = models[mode](noise.cuda(), t, length=torch.LongTensor(length), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
        a = _ddim(models[mode], S, noise, xfp, xfo, length)
        ef, ea = rel_l2(f, ref_f), max(rel_l2(a[c:c + 1], ref[c:c + 1]) for c in range(B))
        print(f"B={B} T={T} ({mode}): forward {ef:.3e}, ddim25 worst clip {ea:.3e}")
        assert torch.isfinite(a).all() and ea <= TOL_PARITY and ef <= (TOL_BF16 if mode == "bf16" else TOL_PARITY)
    if T < 32:
        from diffusion_conductor_amd import native
        with pytest.raises(native.DcError, match="shorter than 32 frames"):
            _with_env({"DC_NO_PAD": "1"}, lambda: _ddim(models["fp16"], S, noise, xfp, xfo, length))


def test_ddim50_t900_ragged_golden(models):
    """G6: 30 s clips (T=900), B=2, ragged lengths."""
    g = golden("g6_variants.npz")
    xfp, xfo = xf_pair(2, 900, first=10)
    noise = torch.from_numpy(batch_noise(2, 900, first=10))
    out = _ddim(models["fp16"], 50, noise, xfp, xfo, [900, 700])
    err = rel_l2(out, g["t900_x0"])
    print(f"ddim50 T=900 rel-L2 {err:.3e}")
    assert err <= TOL_PARITY


def test_ddim1000_graph_replay_golden(models):
    """G6: the full 1000-step schedule (BASELINE config 4): 20 replays of a 50-step hipGraph."""
    g = golden("g6_variants.npz")
    xfp, xfo = xf_pair(1, 1800)
    noise = torch.from_numpy(batch_noise(1, 1800))
    out = _ddim(models["fp16"], 1000, noise, xfp, xfo, [1800])
    err = rel_l2(out, g["ddim1000_x0"])
    print(f"ddim1000 rel-L2 {err:.3e}")
    assert err <= TOL_PARITY
    # (the precise tail lives in the LAST replay's graph only - a second captured graph: 2.3e-4 with it, 5.1e-4 without)
    if "DC_PRECISE_TAIL" not in os.environ:
        assert err <= 3.5e-4


def test_graph_equals_eager(models):
    """hipGraph replay and eager launches run the same kernels: results must be bit-identical."""
    xfp, xfo = xf_pair(2, 96)
    noise = torch.from_numpy(batch_noise(2, 96))
    a = _ddim(models["fp16"], 25, noise, xfp, xfo, [96, 70])
    os.environ["DC_DISABLE_GRAPH"] = "1"
    try:
        b = _ddim(models["fp16"], 25, noise, xfp, xfo, [96, 70])
    finally:
        del os.environ["DC_DISABLE_GRAPH"]
    assert torch.equal(a, b)


def test_captured_loop_table_lookup_equals_begin_step_launches(models):
    """Inside a captured graph the step kernels index the per-iteration tables themselves (no k_begin_step launch; the
    iteration counter advances once per replay).  DDIM-100 = two replays of a 50-step graph, snapshots in both:
    bit-identical to eager launches with the per-step bookkeeping kernel."""
    B, T, S = 2, 512, 100
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T))
    a = _ddim(models["fp16"], S, noise, xfp, xfo, [T, T - 100], idxs=(0, 60, 99))
    os.environ["DC_BEGIN_STEP"] = "1"
    os.environ["DC_DISABLE_GRAPH"] = "1"
    try:
        b = _ddim(models["fp16"], S, noise, xfp, xfo, [T, T - 100], idxs=(0, 60, 99))
    finally:
        del os.environ["DC_BEGIN_STEP"], os.environ["DC_DISABLE_GRAPH"]
    assert sorted(a.keys()) == [0, 60, 99, 100] and all(torch.equal(a[k], b[k]) for k in a)


def test_narrow_workgroups_agree_with_wide_ones(models):
    """Small batches run the layer kernels with 4-wave workgroups (128-token units, one wave per SIMD; default whenever every
    unit gets its own CU), DC_NO_NARROW=1 keeps the 8-wave form.  The two exponentiate the keys against different unit maxima
    before the f16 operand rounding, so they agree at the precision mode's noise level - and both meet the parity bound."""
    B, T = 3, 900
    xfp, xfo = xf_pair(B, T, first=40)
    noise = torch.from_numpy(batch_noise(B, T, first=40))
    length = [900, 512, 333]
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, 25)
    os.environ["DC_DISABLE_GRAPH"] = "1"
    try:
        a = _ddim(models["fp16"], 25, noise, xfp, xfo, length)
        os.environ["DC_NO_NARROW"] = "1"
        b = _ddim(models["fp16"], 25, noise, xfp, xfo, length)
    finally:
        del os.environ["DC_DISABLE_GRAPH"]
        os.environ.pop("DC_NO_NARROW", None)
    ea, eb, d = rel_l2(a, ref), rel_l2(b, ref), rel_l2(a, b.cpu().numpy())
    print(f"narrow vs oracle {ea:.3e}; wide vs oracle {eb:.3e}; narrow vs wide {d:.3e}")
    assert torch.isfinite(a).all() and ea <= TOL_PARITY and eb <= TOL_PARITY and d <= TOL_PARITY


@pytest.mark.parametrize("B,T,length", [(1, 1800, [1800]), (2, 1800, [1800, 1237]), (4, 1800, [1800, 1, 911, 1799]), (5, 1000, [1000, 3, 999, 512, 64]),
                                        (8, 257, [257, 1, 256, 129, 64, 200, 33, 17])])
def test_layer16_small_batches(models, B, T, length):
    """While every clip-aligned 64-token unit gets a CU of its own (bs <= 8 at T = 1800: the reference's one clip per call,
    trainers/ddpm_trainer.py:184) the layers run on 16-token waves (dc_layer16.hip: v_mfma_f32_16x16x32, one wave per SIMD);
    DC_NO_LAYER16=1 keeps the 32-token narrow form.  Against the oracle (DDIM-25, an intermediate and the final sample), against the
    32-token form (different unit maxima before the f16 operand rounding: noise level), re-run identical, and - clip-aligned units -
    every clip bit-identical to sampling it alone."""
    S = 25                       # (the linear schedule needs S > 20: beta_end = 20 / S < 1)
    xfp, xfo = xf_pair(B, T, first=70)
    noise = torch.from_numpy(batch_noise(B, T, first=70))
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S, idxs=(4,))
    m = models["fp16"]
    a = _ddim(m, S, noise, xfp, xfo, length, idxs=(4,))
    a2 = _ddim(m, S, noise, xfp, xfo, length, idxs=(4,))
    os.environ["DC_NO_LAYER16"] = "1"
    try:
        b = _ddim(m, S, noise, xfp, xfo, length, idxs=(4,))
    finally:
        del os.environ["DC_NO_LAYER16"]
    e4, eS, d = rel_l2(a[4], ref[4]), rel_l2(a[S], ref[S]), rel_l2(a[S], b[S].cpu().numpy())
    e32 = rel_l2(b[S], ref[S])
    print(f"layer16 B={B} T={T}: idx4 {e4:.3e} final {eS:.3e} (32-token form {e32:.3e}); 16- vs 32-token form {d:.3e}")
    assert torch.isfinite(a[S]).all() and torch.equal(a[S], a2[S]) and torch.equal(a[4], a2[4])
    assert max(e4, eS) <= TOL_PARITY and d <= TOL_PARITY
    if B > 1:
        k = B - 1
        alone = _ddim(m, S, noise[k:k + 1], xfp[k:k + 1], xfo[k:k + 1], length[k:k + 1])
        assert torch.equal(alone, a[S][k:k + 1])


@pytest.mark.parametrize("B,T,length", [(1, 1800, [1800]), (3, 1800, [1800, 640, 1]), (8, 1800, None), (2, 900, [900, 333]), (7, 257, None)])
def test_layer16_shared_combine_inside_the_launch(models, B, T, length):
    """k_layer16's prologue: the clip's workgroups reduce one slice each of the previous layer's unit records, publish it as tagged
    8-byte granules and gather the clip's operand from each other inside the launch (dc_layer16.hip, round 5) instead of every
    workgroup reading all records.  Against the form in which every workgroup combines alone (DC_L16_OWN_COMBINE=1; another
    summation tree: fp32 rounding level), graph == eager, re-runs identical (tags advance, nothing stale is ever accepted), loops of
    another batch size in between (the granule buffer is cleared when the geometry changes), and the status word stays clean."""
    S = 25
    length = length or [T - 29 * i for i in range(B)]
    xfp, xfo = xf_pair(B, T, first=81)
    noise = torch.from_numpy(batch_noise(B, T, first=81))
    m = models["fp16"]
    a = _ddim(m, S, noise, xfp, xfo, length, idxs=(3,))
    # another geometry in between, then the same loop again
    _ddim(m, S, noise[:1], xfp[:1], xfo[:1], length[:1])
    a2 = _ddim(m, S, noise, xfp, xfo, length, idxs=(3,))
    os.environ["DC_DISABLE_GRAPH"] = "1"
    try:
        c = _ddim(m, S, noise, xfp, xfo, length, idxs=(3,))
    finally:
        del os.environ["DC_DISABLE_GRAPH"]
    os.environ["DC_L16_OWN_COMBINE"] = "1"
    try:
        b = _ddim(m, S, noise, xfp, xfo, length, idxs=(3,))
    finally:
        del os.environ["DC_L16_OWN_COMBINE"]
    d = rel_l2(a[S], b[S].cpu().numpy())
    print(f"shared vs own combine B={B} T={T}: {d:.3e}")
    assert torch.isfinite(a[S]).all() and torch.equal(a[S], a2[S]) and torch.equal(a[3], a2[3]) and torch.equal(a[S], c[S])
    assert d <= TOL_PARITY          # (another summation tree in front of the f16 operand rounding: the same noise level as 16- vs 32-token waves)
    assert m._native.status() == 0


def test_layer16_combine_exchange_timeout_is_bounded_and_reported(models):
    """A workgroup that never publishes its slice (test hook DC_L16_TEST_DROP_SLICE=1; in production: a GPU shared with other work, so
    that the clip's workgroups are not co-resident): the neighbours' wait ends after the poll limit, one forward (eight layer launches)
    returns, dc_sampler_status carries DC_STATUS_TIMEOUT, and the next call without the hook is healthy again."""
    import time
    from diffusion_conductor_amd import native
    B, T = 1, 1800
    xfp, xfo = xf_pair(B, T, first=83)
    x = torch.from_numpy(batch_noise(B, T, first=83)).cuda()
    m = models["fp16"]
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T])
    good = nat.denoise(x, np.array([7], np.int32))
    torch.cuda.synchronize()
    assert nat.status() == 0
    os.environ["DC_L16_TEST_DROP_SLICE"] = "1"
    try:
        t0 = time.perf_counter()
        nat.denoise(x, np.array([7], np.int32))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = nat.status()
    finally:
        del os.environ["DC_L16_TEST_DROP_SLICE"]
    print(f"forward with a dropped slice: {dt:.2f} s, status {st}")
    # one timeout per forward, not one per layer launch: once a workgroup has given up, the launches that follow stop waiting
    # (8 launches x 0.1 s before round 6)
    assert st & native.STATUS_TIMEOUT and dt < 1.0
    # reading the timeout has latched the form without the exchange on this sampler (co-residency cannot be assumed on this GPU): the
    # next call is healthy, and equal to what DC_L16_OWN_COMBINE=1 computes (another summation tree than the exchange: noise level)
    again = nat.denoise(x, np.array([7], np.int32))
    torch.cuda.synchronize()
    assert nat.status() == 0
    own = _with_env({"DC_L16_OWN_COMBINE": "1"}, lambda: nat.denoise(x, np.array([7], np.int32)))
    torch.cuda.synchronize()
    assert torch.equal(again, own) and rel_l2(again, good) <= 1e-3
    nat.set_combine_exchange(True)                     # (module-scoped model: back to the default for the tests that follow)
    back = nat.denoise(x, np.array([7], np.int32))
    torch.cuda.synchronize()
    assert torch.equal(back, good)


@pytest.mark.parametrize("B,T,length", [(1, 1800, [1800]), (3, 1800, [1800, 77, 1500]), (4, 1800, [1800, 1, 911, 1799]), (12, 1800, None), (6, 512, None)])
def test_small_batch_embedding_rides_in_the_film_launch(models, B, T, length):
    """Small batches (narrow clip-aligned units): the embedding's units are extra workgroups of the FiLM launch - beside the GEMM's
    while both fit the chip (film_extra_workgroups; bs <= 3 at T = 1800), dispatched behind them with the chip full (bs = 4, 12).
    Same kernels' bodies either way: bit-identical to DC_NO_FUSE_EMBED=1, graph and eager."""
    S = 25
    length = length or [T - 13 * i for i in range(B)]
    xfp, xfo = xf_pair(B, T, first=74)
    noise = torch.from_numpy(batch_noise(B, T, first=74))
    m = models["fp16"]
    a = _ddim(m, S, noise, xfp, xfo, length, idxs=(3,))
    os.environ["DC_NO_FUSE_EMBED"] = "1"
    try:
        b = _ddim(m, S, noise, xfp, xfo, length, idxs=(3,))
        os.environ["DC_DISABLE_GRAPH"] = "1"
        c = _ddim(m, S, noise, xfp, xfo, length, idxs=(3,))
    finally:
        del os.environ["DC_NO_FUSE_EMBED"]
        os.environ.pop("DC_DISABLE_GRAPH", None)
    assert torch.isfinite(a[S]).all()
    assert torch.equal(a[S], b[S]) and torch.equal(a[3], b[3]) and torch.equal(a[S], c[S])


def test_layer16_bf16_build(models):
    """The bf16 instantiation of the 16-token kernel (plain bf16 operands: loosely bounded like every plain-bf16 result) against the
    oracle and against the 32-token narrow form of the same mode."""
    B, T, S = 2, 1800, 25
    length = [1800, 977]
    xfp, xfo = xf_pair(B, T, first=72)
    noise = torch.from_numpy(batch_noise(B, T, first=72))
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S)
    a = _ddim(models["bf16"], S, noise, xfp, xfo, length)
    os.environ["DC_NO_LAYER16"] = "1"
    try:
        b = _ddim(models["bf16"], S, noise, xfp, xfo, length)
    finally:
        del os.environ["DC_NO_LAYER16"]
    ea, eb = rel_l2(a, ref), rel_l2(b, ref)
    print(f"layer16 bf16: 16-token {ea:.3e}, 32-token {eb:.3e}, apart {rel_l2(a, b.cpu().numpy()):.3e}")
    assert torch.isfinite(a).all() and ea <= 1e-2 and eb <= 1e-2 and ea <= 2 * eb + 1e-3


def test_clip_layouts_agree_and_small_batches_are_batch_invariant(models):
    """Where the workgroup-record kernels run, a clip's stride in the token space is padded to whole 32-frame groups (T = 900 -> 928
    for a small batch) and small batches run clip-aligned 4-wave workgroups: no group and no workgroup spans two clips.  The
    layouts (padded + aligned, padded + flat units, unpadded; DC_ALIGN=1 with 8-wave workgroups; per-group records + combine
    launches on the padded stride) differ only by which unit maxima the keys are exponentiated against: all within the parity bound, 2e-4 apart.  With aligned units a clip never
    shares a workgroup, so its result does not depend on the batch around it: bit-identical to sampling it alone."""
    B, T = 3, 900
    xfp, xfo = xf_pair(B, T, first=50)
    noise = torch.from_numpy(batch_noise(B, T, first=50))
    length = [900, 611, 333]
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, 25)
    outs = {}
    os.environ["DC_DISABLE_GRAPH"] = "1"          # eager launches: the switches are read when a launch is enqueued
    try:
        for name, env in (("aligned", {}), ("flat", {"DC_NO_ALIGN": "1"}), ("unpadded", {"DC_NO_PAD": "1"}),
                          ("aligned-wide", {"DC_NO_NARROW": "1", "DC_ALIGN": "1"}), ("flat-wide", {"DC_NO_NARROW": "1", "DC_FLAT_UNITS": "1"}),
                          ("per-group records", {"DC_NO_WGREC": "1"})):
            os.environ.update(env)
            try:
                outs[name] = _ddim(models["fp16"], 25, noise, xfp, xfo, length)
            finally:
                for k in env:
                    del os.environ[k]
        alone = [_ddim(models["fp16"], 25, noise[b:b + 1], xfp[b:b + 1], xfo[b:b + 1], length[b:b + 1]) for b in range(B)]
    finally:
        del os.environ["DC_DISABLE_GRAPH"]
    for name, o in outs.items():
        e = rel_l2(o, ref)
        print(f"layout {name}: rel-L2 vs oracle {e:.3e}, vs aligned {rel_l2(o, outs['aligned'].cpu().numpy()):.3e}")
        assert torch.isfinite(o).all() and e <= TOL_PARITY
    assert all(torch.equal(outs["aligned"][b:b + 1], alone[b]) for b in range(B))


def test_embedding_fused_into_the_film_launch_is_bit_identical(models):
    """With the chip full (wide flat units: 18 x 1800 frames and up) k_embed_front rides in the FiLM GEMM's launch; DC_NO_FUSE_EMBED=1
    keeps the two launches.  Same arithmetic per workgroup either way: bit-identical."""
    B, T = 18, 1800
    xfp, xfo = xf_pair(B, T, first=3)
    noise = torch.from_numpy(batch_noise(B, T, first=3))
    a = _ddim(models["fp16"], 25, noise, xfp, xfo, [T] * B)
    os.environ["DC_NO_FUSE_EMBED"] = "1"
    os.environ["DC_DISABLE_GRAPH"] = "1"          # eager launches: the switch is read when a launch is enqueued
    try:
        b = _ddim(models["fp16"], 25, noise, xfp, xfo, [T] * B)
    finally:
        del os.environ["DC_NO_FUSE_EMBED"], os.environ["DC_DISABLE_GRAPH"]
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_film_adaptive_shares_do_not_change_results(models):
    """The persistent FiLM GEMM sizes its workgroups' shares by the per-XCD speeds measured in earlier launches (>= 64
    workgroups); which workgroup computes a tile must not matter: bit-identical to equal shares (DC_FILM_STATIC=1) and
    across repeated loops whose shares differ."""
    B, T = 5, 1800
    xfp, xfo = xf_pair(B, T, first=40)
    noise = torch.from_numpy(batch_noise(B, T, first=40))
    a = _ddim(models["fp16"], 25, noise, xfp, xfo, [T] * B)
    a2 = _ddim(models["fp16"], 25, noise, xfp, xfo, [T] * B)
    os.environ["DC_FILM_STATIC"] = "1"
    os.environ["DC_DISABLE_GRAPH"] = "1"          # eager launches: the switch is read when a launch is enqueued
    try:
        b = _ddim(models["fp16"], 25, noise, xfp, xfo, [T] * B)
    finally:
        del os.environ["DC_FILM_STATIC"], os.environ["DC_DISABLE_GRAPH"]
    assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, a2)


def test_progressive_matches_fast_path(models):
    """ddim_sample_loop_progressive (per-step host loop over the native denoiser) ends where the
    graph-replayed loop ends, and yields num_timesteps samples."""
    xfp, xfo = xf_pair(2, 64)
    noise = torch.from_numpy(batch_noise(2, 64))
    m = models["bf16x3"]     # ~fp32-accurate mode: the two paths differ only by fp32 rounding of the update
    fast = _ddim(m, 25, noise, xfp, xfo, [64, 64])
    gd = make_diffusion(25)
    n, last = 0, None
    for s in gd.ddim_sample_loop_progressive(m, (2, 64, 26), noise=noise.cuda(), clip_denoised=False,
                                             model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(),
                                                           "length": torch.LongTensor([64, 64])}):
        n, last = n + 1, s
    assert n == 25 and set(last.keys()) == {"sample", "pred_xstart"}
    assert rel_l2(last["sample"], fast) <= 1e-5


def test_bs32_full_size_properties(models):
    """BASELINE config 2 size (bs=32, T=1800, DDIM-50), checked through size-independent properties:
    (a) re-running is bit-identical (race check: all reductions are ordered);
    (b) batch invariance (the reference's key softmax is per clip, transformer.py:111): the headline shape runs clip-aligned 256-token
        units (dc_ddim.h, dc_sampler_set_clip_aligned: 256 workgroups instead of 228, one round over the chip either way), so a clip's
        result does not depend on the batch around it: both 16-clip shards, a clip sampled ALONE and the batch in reverse order are
        bit-identical to the joint batch (all in the same 8-wave launch form; the narrow small-batch form sums the unit records in
        another tree and agrees at the precision mode's noise level);
    (c) the flat-unit form (DC_FLAT_UNITS=1, the throughput option: a unit with a clip edge exponentiates both clips' keys against one
        maximum) agrees with it at the noise level;
    (d) clip 0 of the batch matches the golden single-clip result within the parity bound."""
    B, T = 32, 1800
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T))
    m = models["fp16"]
    a = _ddim(m, 50, noise, xfp, xfo, [T] * B)
    b = _ddim(m, 50, noise, xfp, xfo, [T] * B)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    # 16-clip shards and single clips fit one 128-token unit per CU and would run narrow workgroups: the bit-for-bit statement is about
    # the same launch form, so they are run in the 8-wave form here; the narrow form is compared at the noise level
    os.environ["DC_NO_NARROW"] = "1"
    try:
        lo = _ddim(m, 50, noise[:16], xfp[:16].contiguous(), xfo[:16].contiguous(), [T] * 16)
        hi = _ddim(m, 50, noise[16:], xfp[16:].contiguous(), xfo[16:].contiguous(), [T] * 16)
        solo = _ddim(m, 50, noise[21:22], xfp[21:22].contiguous(), xfo[21:22].contiguous(), [T])
    finally:
        del os.environ["DC_NO_NARROW"]
    assert torch.equal(lo, a[:16]) and torch.equal(hi, a[16:]) and torch.equal(solo, a[21:22])
    rev = _ddim(m, 50, noise.flip(0).contiguous(), xfp.flip(0).contiguous(), xfo.flip(0).contiguous(), [T] * B)      # other neighbours, other positions
    assert torch.equal(rev.flip(0), a)
    flat = _with_env({"DC_FLAT_UNITS": "1"}, lambda: _ddim(m, 50, noise, xfp, xfo, [T] * B))
    e_flat = max(rel_l2(flat[c:c + 1], a[c:c + 1]) for c in range(B))
    print(f"flat 256-token units vs clip-aligned units, worst clip: rel-L2 {e_flat:.2e}")
    assert 0 < e_flat <= TOL_PARITY
    lo_n = _ddim(m, 50, noise[:8], xfp[:8].contiguous(), xfo[:8].contiguous(), [T] * 8)       # narrow workgroups (B = 8)
    e_n = rel_l2(lo_n, a[:8])
    print(f"shard 0..7 with narrow workgroups vs joint batch: rel-L2 {e_n:.2e}")
    assert e_n <= TOL_PARITY
    err = rel_l2(a[:1], golden("g5_ddim50_b1.npz")["x0"])
    print(f"bs32 clip0 rel-L2 {err:.3e}")
    assert err <= TOL_PARITY


def test_bs32_interior_clips_vs_oracle(models):
    """The headline configuration (bs=32 x 1800, DDIM-50; clip-aligned 256-token units since round 6, flat units before): clips 13, 17
    and 31 from inside the batch, each compared with the CPU oracle's DDIM-50 of that clip ALONE
    (gaussian_diffusion.py:871-915, transformer.py:96-196), in the default f16 mode and in the bf16-MFMA mode ("mixed", which
    always runs clip-aligned units)."""
    B, T = 32, 1800
    clips = (13, 17, 31)
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T))
    p = oracle_params()
    idx = list(clips)
    with torch.no_grad():       # (one oracle call for the three clips: the oracle has no cross-clip operation, a clip of the batch = the clip alone)
        ref3 = O.ddim_sample_loop(p, noise[idx], xfp[idx], xfo[idx], [T] * len(idx), 50)
    refs = {c: ref3[i:i + 1] for i, c in enumerate(clips)}
    worst = {}
    for mode in ("fp16", "mixed"):
        a = _ddim(models[mode], 50, noise, xfp, xfo, [T] * B)
        errs = {c: rel_l2(a[c:c + 1], refs[c]) for c in clips}
        worst[mode] = max(errs.values())
        print(f"bs32 interior clips vs oracle ({mode}): " + "  ".join(f"clip {c}: {e:.3e}" for c, e in errs.items()))
    assert worst["fp16"] <= TOL_PARITY and worst["mixed"] <= TOL_PARITY, worst


def test_clip_stride_avoids_an_extra_round_of_layer_workgroups(models):
    """The clip stride is padded to whole 32-frame groups (1800 -> 1824) only while the padding frames do not tip the layer launches
    into another round of 256-token workgroups on the 256 CUs: 36 clips are 254 workgroups unpadded, 257 padded (51 vs 37 ms per
    loop).  The unpadded batch runs flat units whose groups straddle clip edges: two interior clips against the oracle (DDIM-25)."""
    T = 1800
    m = models["fp16"]
    strides = {}
    for B in (32, 36):
        xfp, xfo = xf_pair(B, T, first=90)
        nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
        strides[B] = nat.clip_stride()
    assert strides == {32: 1824, 36: 1800}, strides
    B, clips = 36, (17, 35)
    xfp, xfo = xf_pair(B, T, first=90)
    noise = torch.from_numpy(batch_noise(B, T, first=90))
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise[list(clips)], xfp[list(clips)], xfo[list(clips)], [T] * len(clips), 25)
    a = _ddim(m, 25, noise, xfp, xfo, [T] * B)
    errs = [rel_l2(a[c:c + 1], ref[i:i + 1]) for i, c in enumerate(clips)]
    print("bs36 (unpadded stride) clips 17 / 35 vs oracle: " + "  ".join(f"{e:.3e}" for e in errs))
    assert max(errs) <= TOL_PARITY


def test_bs32_every_clip_vs_oracle(models):
    """All 32 clips of the headline batch (bs=32 x 1800, DDIM-50, clip-aligned 256-token units) against the CPU oracle's DDIM-50 of the same batch (the oracle has no cross-clip operation), per
    clip, in the default f16 mode and the bf16-MFMA mode."""
    B, T = 32, 1800
    xfp, xfo = xf_pair(B, T)
    noise = torch.from_numpy(batch_noise(B, T))
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, [T] * B, 50)
    for mode in ("fp16", "mixed"):
        a = _ddim(models[mode], 50, noise, xfp, xfo, [T] * B)
        errs = [rel_l2(a[c:c + 1], ref[c:c + 1]) for c in range(B)]
        print(f"bs32, every clip vs oracle ({mode}): min {min(errs):.3e} max {max(errs):.3e} (clip {int(np.argmax(errs))})")
        assert max(errs) <= TOL_PARITY, errs


def test_harness_generate_music_motion_golden(models):
    """G7: DDPMTrainer.generate_music_motion end to end (encode_music + DDIM-50) on the reference's
    own call pattern: np mel [5400,128] in, tensor [1,1800,26] out."""
    import types
    from diffusion_conductor_amd import DDPMTrainer
    g = golden("g7_harness.npz")
    opt = types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=50, is_train=False)
    tr = DDPMTrainer(opt, models["fp16"])
    tr.eval_mode()
    torch.manual_seed(int(g["torch_seed"]))
    noise = torch.randn(1, 1800, 26)
    out = tr.generate_music_motion(batch_mel(1, 5400)[0], 26, noise=noise)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1, 1800, 26)
    err = rel_l2(out, g["x0"])
    print(f"harness rel-L2 {err:.3e}")
    assert err <= TOL_PARITY


TOL_ENC_SPLIT = 1e-4      # two bf16 planes, three MFMAs per product (~16 mantissa bits): measured 6 - 7e-6
TOL_ENC_F16 = 6e-4        # one fp16 plane: every activation rounded to 11 bits once per layer: measured 3.6 - 3.9e-4


def _enc_env(fmt, extra=None):
    env = {"DC_ME_PREC": fmt}
    env.update(extra or {})
    return env


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_encode_music_golden(models):
    """G4: MusicEncoder + proj through dc_sampler_encode_music vs the reference's own outputs, in both activation formats
    (include/dc_ddim.h, dc_sampler_set_encoder_format): split bf16 planes 1e-4 rel-L2, one fp16 plane 6e-4."""
    g = golden("g4_encode_music.npz")
    m = models["fp16"]
    for fmt, tol in (("split", TOL_ENC_SPLIT), ("f16", TOL_ENC_F16)):
        xp, x = _with_env(_enc_env(fmt), lambda: m.encode_music(torch.from_numpy(batch_mel(1, 270)).cuda(), "cuda:0"))
        e1, e2 = rel_l2(xp, g["small_x_proj"]), rel_l2(x, g["small_x"])
        xp, x = _with_env(_enc_env(fmt), lambda: m.encode_music(torch.from_numpy(batch_mel(1, 5400)).cuda(), "cuda:0"))
        e3 = rel_l2(xp.cpu()[:, ::25], g["full_x_proj_sub"])
        print(f"encode_music golden, {fmt}: {e1:.2e} {e2:.2e} {e3:.2e}")
        assert max(e1, e2, e3) <= tol
        if fmt == "f16":
            assert min(e1, e2, e3) > 1e-5          # (the single-plane kernels did run)


def test_encode_music_batched_vs_oracle(models):
    """Clips are encoded in chunks (32 at full length; DC_ME_CHUNK=4 makes a batch of 9 cross two chunk edges).  30-s clips
    (Tm = 2700 -> T = 900) and an odd frame count (271 = 33 x 8 + 7: partial row tiles of the LDS-tiled convolutions, the
    stride-3 pool's floor) against the oracle, in both activation formats; for the split format the fused conv1 / conv2 kernels and
    their separate launches (DC_ME_NO_STEM=1, DC_ME_NO_MID=1) both."""
    from oracle import ddim_oracle as O
    m = models["fp16"]
    p = oracle_params()
    cases = [(9, 5400, {"DC_ME_CHUNK": "4"}, "split"), (3, 2700, {}, "split"), (2, 271, {}, "split"),
             (2, 271, {"DC_ME_NO_STEM": "1", "DC_ME_NO_MID": "1"}, "split"), (2, 2700, {"DC_ME_NO_STEM": "1"}, "split"),
             (2, 2700, {"DC_ME_NO_MID": "1"}, "split"),
             (9, 5400, {"DC_ME_CHUNK": "4"}, "f16"), (3, 2700, {}, "f16"), (2, 271, {}, "f16"), (1, 33, {}, "f16"), (1, 33, {}, "split")]
    for B, Tm, env, fmt in cases:
        mel = torch.from_numpy(batch_mel(B, Tm))

        def run():
            out = m.encode_music(mel.cuda(), "cuda:0")
            torch.cuda.synchronize()
            return out
        xp, x = _with_env(_enc_env(fmt, env), run)
        with torch.no_grad():
            rxp, rx = O.encode_music(p, mel)
        assert tuple(x.shape) == tuple(rx.shape)
        e1, e2 = rel_l2(xp, rxp), rel_l2(x, rx)
        print(f"encode_music B={B} Tm={Tm} {env} {fmt}: rel-L2 x_proj {e1:.2e} x {e2:.2e}")
        tol = TOL_ENC_SPLIT if fmt == "split" else TOL_ENC_F16
        assert e1 <= tol and e2 <= tol


@pytest.mark.parametrize("Tm", [4, 5, 7, 10, 13])
def test_encode_music_of_a_few_mel_frames(models, Tm):
    """Mel spectrograms shorter than the convolutions' row tiles (found by tools/fuzz_encoder.py: the halo rows of a tile that reaches past
    the image were reflected only once and left an image of fewer rows than the tile on the other side - a memory fault at Tm = 4).  Both
    activation formats against the oracle; below 4 frames the reference's reflection padding raises, and so does the library."""
    p = oracle_params()
    mel = torch.from_numpy(batch_mel(3, Tm, first=5))
    with torch.no_grad():
        rp, rx = O.encode_music(p, mel)
    for fmt, tol in (("split", TOL_ENC_SPLIT), ("f16", TOL_ENC_F16)):
        xp, x = _with_env(_enc_env(fmt), lambda: models["fp16"].encode_music(mel.cuda(), "cuda:0"))
        torch.cuda.synchronize()
        e = max(rel_l2(xp, rp), rel_l2(x, rx))
        print(f"encode_music Tm={Tm} ({fmt}): {e:.2e}")
        assert tuple(x.shape) == tuple(rx.shape) and e <= tol
    from diffusion_conductor_amd import native
    with pytest.raises(native.DcError, match="at least 4 mel frames"):
        models["fp16"].encode_music(torch.from_numpy(batch_mel(1, 3)).cuda(), "cuda:0")
    with pytest.raises(RuntimeError):
        O.encode_music(p, torch.from_numpy(batch_mel(1, 3)))


def test_encoder_format_follows_the_precision_and_can_be_set(models):
    """Default format: one fp16 plane beside the fp16 denoiser (whose conditioning pre-pass rounds the features to fp16 operands anyway),
    split planes beside the split-operand precisions; dc_sampler_set_encoder_format overrides it, DC_ME_PREC overrides both; an
    unknown format is refused."""
    from diffusion_conductor_amd import native
    from oracle import ddim_oracle as O
    assert "DC_ME_PREC" not in os.environ
    mel = torch.from_numpy(batch_mel(1, 540))
    with torch.no_grad():
        _, rx = O.encode_music(oracle_params(), mel)
    err = lambda m: rel_l2(m.encode_music(mel.cuda(), "cuda:0")[1], rx)
    e16, esp = err(models["fp16"]), err(models["mixed"])
    print(f"default formats: fp16 model {e16:.2e}, mixed model {esp:.2e}")
    assert 1e-5 < e16 <= TOL_ENC_F16 and esp <= TOL_ENC_SPLIT
    nat = models["fp16"]._ensure_native("cuda:0")
    try:
        nat.set_encoder_format("split")
        assert err(models["fp16"]) <= TOL_ENC_SPLIT
        assert 1e-5 < _with_env({"DC_ME_PREC": "f16"}, lambda: err(models["fp16"])) <= TOL_ENC_F16
        with pytest.raises(native.DcError, match="encoder format"):
            native._check(native.lib().dc_sampler_set_encoder_format(nat._h, 7))
    finally:
        nat.set_encoder_format("f16")
    assert 1e-5 < err(models["fp16"]) <= TOL_ENC_F16
    fresh = make_model("fp16")                      # the attribute, set before the module's sampler exists: applied when the parameters are finalized
    fresh.encoder_format = "split"
    assert err(fresh) <= TOL_ENC_SPLIT


def test_error_behaviour(models):
    from diffusion_conductor_amd import native
    from diffusion_conductor_amd.param_spec import DenoiserConfig
    s = native.NativeSampler(DenoiserConfig(), "fp16", 100, 0)
    with pytest.raises(native.DcError, match="unknown parameter"):
        s.load_state_dict({"not.a.key": np.zeros(3, np.float32)})
    with pytest.raises(native.DcError, match="missing parameter"):
        s.load_state_dict({})
    x = torch.zeros(1, 64, 26, device="cuda")
    s.B, s.T = 1, 64
    with pytest.raises(native.DcError, match="not finalized"):
        s.denoise(x, [0])
    s.close()
    with pytest.raises(native.DcError, match="no_eff"):
        native.NativeSampler(DenoiserConfig(no_eff=True), "bf16x3")     # full attention: fp16 only
    with pytest.raises(RuntimeError, match="no CPU path"):
        make_model("fp16", device="cpu")(torch.zeros(1, 64, 26), torch.zeros(1, dtype=torch.long),
                                          length=[64], xf_proj=torch.zeros(1, 64, 64), xf_out=torch.zeros(1, 64, 64))


# ---- no_eff variant: full T x T attention (transformer.py:198-287) ----------------------------------------
@pytest.fixture(scope="module")
def model_no_eff():
    return make_model("fp16", no_eff=True)


def test_no_eff_forward_golden(model_no_eff):
    """G3 `forward_no_eff`: one forward of the full-attention variant at B=2, T=64, ragged length (the padded
    query rows carry the reference's -1e5 shift), per-clip timesteps - against the reference's own output."""
    g = golden("g3_blocks.npz")
    out = model_no_eff(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["t"]), length=torch.from_numpy(g["length"]),
                       xf_proj=torch.from_numpy(g["xf_proj"]).cuda(), xf_out=torch.from_numpy(g["xf_out"]).cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, g["forward_no_eff"])
    print(f"forward no_eff rel-L2 {err:.3e}")
    assert torch.isfinite(out).all() and err <= 5e-3        # single forward at mixed timesteps; the gate is on DDIM x0


def test_no_eff_ddim50_golden(model_no_eff):
    """G6b: DDIM-50 of the full-attention variant, B=2, T=96, lengths [96, 70] (reference output)."""
    xfp, xfo = xf_pair(2, 96, first=20)
    noise = torch.from_numpy(batch_noise(2, 96, first=20))
    out = _ddim(model_no_eff, 50, noise, xfp, xfo, [96, 70])
    err = rel_l2(out, golden("g6b_no_eff.npz")["x0"])
    print(f"ddim50 no_eff rel-L2 {err:.3e}")
    assert err <= TOL_PARITY


def test_no_eff_ddim50_long_goldens_g10(model_no_eff):
    """G10: the imported reference's no_eff DDIM-50 x0 at production length - (a) B=1, T=1800 (57 key tiles per query: the
    lengths at which the key loop's lazily moved reference point actually moves and 1 800 f16 weights are summed), seed-0
    checkpoint; (b) B=2, T=900, lengths [900, 613] on the trained-like stress checkpoint."""
    from helpers import DenoiserConfig
    from diffusion_conductor_amd import MotionTransformer
    from diffusion_conductor_amd.synthetic import batch_music_features, stress_state_dict
    g = golden("g10_no_eff_long.npz")
    xfp, xfo = xf_pair(1, 1800, first=50)
    noise = torch.from_numpy(batch_noise(1, 1800, first=50))
    out = _ddim(model_no_eff, 50, noise, xfp, xfo, [1800])
    e1 = rel_l2(out, g["t1800_x0"])
    print(f"no_eff DDIM-50 B=1 T=1800 vs reference: rel-L2 {e1:.3e}")
    sd = stress_state_dict(DenoiserConfig(), seed=0)
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True,
                          precision="fp16", no_eff=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    q = O.to_torch_params(sd, torch.float32)
    xf2 = torch.from_numpy(batch_music_features(2, 900, first=52))
    xfp2 = torch.nn.functional.linear(xf2, q["proj.weight"], q["proj.bias"])
    noise2 = torch.from_numpy(batch_noise(2, 900, first=52))
    length = [int(v) for v in g["stress_t900_length"]]
    out2 = _ddim(m, 50, noise2, xfp2, xf2, length)
    e2 = rel_l2(out2, g["stress_t900_x0"])
    print(f"no_eff DDIM-50 B=2 T=900 ragged, stress checkpoint, vs reference: rel-L2 {e2:.3e}")
    assert torch.isfinite(out).all() and torch.isfinite(out2).all()
    assert e1 <= TOL_PARITY and e2 <= TOL_PARITY, (e1, e2)


def _peaky_state_dict(scale=5.0):
    """seed-0 checkpoint with the self-attention query / key projections scaled: scores x scale^2, i.e. attention rows with a few
    dominant keys - later key tiles then hold scores far above tile 0's maximum, which is what makes the no_eff key loop MOVE its
    lazily kept reference point (on the plain synthetic checkpoints it never does: tools/noeff_moves.py counts 0 moves in 2e7 visits)."""
    from helpers import state_dict_np
    sd = dict(state_dict_np())
    for k in list(sd):
        if ".sa_block.query." in k or ".sa_block.key." in k:
            sd[k] = (sd[k] * scale).astype(np.float32)
    return sd


@pytest.mark.parametrize("scale", [3.0, 5.0])
def test_no_eff_peaky_attention_moves_the_reference_point(scale):
    """The key loop's softmax reference point is fixed by key tile 0 and moves only when a later tile's weights sum past 64 in some
    lane.  A checkpoint with sharpened self-attention (scores x scale^2) makes that happen (the diagnostic build counts the moves:
    tools/noeff_moves.py, 1.5 % of the visits at scale 5; never on the plain checkpoints).  Sharp attention is also where 16-bit
    operands lose ABSOLUTE precision in the scores (a score of 50 carries +-0.02), so the gate is the oracle's own f16-operand
    emulation of the same model: the kernels must be as accurate as f16 arithmetic permits (<= 1e-3, or twice the emulation's
    error where that is larger) - a wrong rescale in the move path would be off by orders of magnitude.  One forward at T = 900,
    ragged, per-clip timesteps, and a DDIM-25 at T = 320."""
    from diffusion_conductor_amd import MotionTransformer
    sd = _peaky_state_dict(scale)
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True,
                          precision="fp16", no_eff=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    p = O.to_torch_params(sd, torch.float32)
    emu = O.Emu("fp16")
    B, T = 2, 900
    xfp, xfo = xf_pair(B, T, first=80)
    x = torch.from_numpy(batch_noise(B, T, first=80))
    t = torch.tensor([30, 5])
    length = [900, 433]
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo, no_eff=True)
        ref_emu = O.denoiser_forward(p, x, t, length, xfp, xfo, no_eff=True, emu=emu)
    out = m(x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
    torch.cuda.synchronize()
    e_fwd, emu_fwd = rel_l2(out, ref), rel_l2(ref_emu, ref)
    B2, T2, S = 2, 320, 25
    xfp2, xfo2 = xf_pair(B2, T2, first=82)
    nz = torch.from_numpy(batch_noise(B2, T2, first=82))
    with torch.no_grad():
        ref2 = O.ddim_sample_loop(p, nz, xfp2, xfo2, [320, 211], S, no_eff=True)
        ref2_emu = O.ddim_sample_loop(p, nz, xfp2, xfo2, [320, 211], S, no_eff=True, emu=emu)
    out2 = _ddim(m, S, nz, xfp2, xfo2, [320, 211])
    e_x0, emu_x0 = rel_l2(out2, ref2), rel_l2(ref2_emu, ref2)
    print(f"no_eff peaky attention x{scale}: forward T=900 rel-L2 {e_fwd:.3e} (f16 emulation on the CPU: {emu_fwd:.3e}); "
          f"DDIM-25 T=320 x0 rel-L2 {e_x0:.3e} (emulation {emu_x0:.3e})")
    assert torch.isfinite(out).all() and torch.isfinite(out2).all()
    assert e_fwd <= max(5e-3, 2 * emu_fwd) and e_x0 <= max(TOL_PARITY, 2 * emu_x0)


def test_no_eff_straddling_clips_vs_oracle(model_no_eff):
    """T = 77: clip edges fall inside 32-token groups, so edge groups are computed by both neighbouring
    workgroups (each for its own lanes) and edge key tiles are partly masked; 5 clips, ragged lengths."""
    B, T = 5, 77
    p = oracle_params()
    xfp, xfo = xf_pair(B, T, first=30)
    x = torch.from_numpy(batch_noise(B, T, first=30))
    t = torch.tensor([0, 49, 13, 999, 500])
    length = [77, 1, 40, 76, 33]
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo, no_eff=True)
    out = model_no_eff(x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, ref)
    print(f"no_eff straddle rel-L2 {err:.3e}")
    assert err <= 5e-3


def test_no_eff_long_clip_vs_oracle(model_no_eff):
    """A 60-s clip (T = 1800: 57 key tiles, 8 workgroups per clip) for 3 forwards' worth of DDIM at S=25."""
    xfp, xfo = xf_pair(2, 1800, first=3)
    noise = torch.from_numpy(batch_noise(2, 1800, first=3))
    p = oracle_params()
    t = torch.tensor([24, 3])
    with torch.no_grad():
        ref = O.denoiser_forward(p, noise, t, [1800, 1800], xfp, xfo, no_eff=True)
    out = model_no_eff(noise.cuda(), t, length=torch.tensor([1800, 1800]), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, ref)
    print(f"no_eff T=1800 forward rel-L2 {err:.3e}")
    assert err <= 5e-3


def test_no_eff_stress_checkpoint_ragged_long_clips_vs_oracle():
    """Round 3's key loop keeps the reference point of the exponentials inside the score MFMA, moves it lazily (only when a
    tile's weights sum past 64) and takes the general path on the first tile, clip edges and waves with padded query rows.
    The trained-like stress checkpoint (x3 weights, outlier channels: scores several times larger, reference points that do
    move) on 29-tile clips with ragged lengths exercises all of them; one forward per clip against the oracle."""
    from helpers import DenoiserConfig
    from diffusion_conductor_amd import MotionTransformer
    from diffusion_conductor_amd.synthetic import batch_music_features, stress_state_dict
    sd = stress_state_dict(DenoiserConfig(), seed=0)
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True,
                          precision="fp16", no_eff=True)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    p = O.to_torch_params(sd, torch.float32)
    B, T = 3, 900
    xf = torch.from_numpy(batch_music_features(B, T, first=40))
    xfp = torch.nn.functional.linear(xf, p["proj.weight"], p["proj.bias"])
    x = torch.from_numpy(batch_noise(B, T, first=40))
    t = torch.tensor([49, 7, 0])
    length = [900, 611, 35]
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xf, no_eff=True)
    out = m(x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xf.cuda())
    again = m(x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xf.cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, ref)
    per_clip = [rel_l2(out[b:b + 1], ref[b:b + 1]) for b in range(B)]
    print(f"no_eff stress checkpoint T=900 ragged: rel-L2 {err:.3e} per clip {['%.2e' % e for e in per_clip]}")
    assert torch.isfinite(out).all() and torch.equal(out, again)
    assert err <= 5e-3 and max(per_clip) <= 1e-2


def test_no_eff_bs32_full_size_properties(model_no_eff):
    """bs=32 x 1800 through the full-attention kernels (the `bench.py --no-eff` shape, 25 steps): finite, bit-identical when re-run,
    and a clip sampled alone stays within the parity bound of the same clip inside the batch (its key tiles are cut at the flat
    32-token groups, i.e. at offsets that depend on the clip's position in the batch: another summation order and other f16
    roundings of the softmax weights - the same kind of neighbour dependence DESIGN section 5 notes for flat units)."""
    B, T, S = 32, 1800, 25
    xfp, xfo = xf_pair(B, T, first=60)
    noise = torch.from_numpy(batch_noise(B, T, first=60))
    length = [T if b % 4 else T - 11 * b for b in range(B)]
    a = _ddim(model_no_eff, S, noise, xfp, xfo, length)
    b = _ddim(model_no_eff, S, noise, xfp, xfo, length)
    sub = slice(5, 9)
    c = _ddim(model_no_eff, S, noise[sub], xfp[sub], xfo[sub], length[sub])
    err = rel_l2(a[sub], c)
    print(f"no_eff bs=32: clips 5..8 inside the batch vs alone: equal {torch.equal(a[sub], c)}, rel-L2 {err:.3e}")
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert err <= TOL_PARITY
    # ... and clips from inside the batch (13: shortened, 30: full length) against the oracle's full-attention run of those clips
    idx = [13, 30]
    with torch.no_grad():
        ref = O.ddim_sample_loop(oracle_params(), noise[idx], xfp[idx], xfo[idx], [length[i] for i in idx], S, no_eff=True)
    errs = [rel_l2(a[i:i + 1], ref[k:k + 1]) for k, i in enumerate(idx)]
    print("no_eff bs=32, clips inside the batch vs oracle: " + "  ".join(f"clip {i}: {e:.3e}" for i, e in zip(idx, errs)))
    assert max(errs) <= TOL_PARITY


def test_no_eff_is_offered_in_fp16_only():
    """Full attention with bf16 scores, weights and values sits 1.0 - 1.8e-3 from the reference on x0 whatever the precise tail
    (profiles/r06_fuzz_bf16_tail.txt): outside the 1e-3 parity bound, so the combination is refused at construction - by the Python
    class and by dc_sampler_create - instead of returning such results (round 5 built it, loosely bounded)."""
    from diffusion_conductor_amd import MotionTransformer, native
    from diffusion_conductor_amd.param_spec import DenoiserConfig
    for prec in ("bf16", "mixed", "bf16x3"):
        with pytest.raises(ValueError, match="precision='fp16' only"):
            MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_eff=True, precision=prec)
        with pytest.raises(native.DcError, match="fp16 precision only"):
            native.NativeSampler(DenoiserConfig(no_eff=True), prec)


# ---- post-processing and the batched evaluation driver ------------------------------------------
def test_savgol_smoothing_matches_scipy():
    """smooth_motion (tools/visualization.py:20-26, kernel 19, order 5) on the GPU vs scipy.signal.savgol_filter."""
    from scipy.signal import savgol_filter
    from diffusion_conductor_amd.evaluate import smooth_motion
    x = torch.from_numpy(batch_noise(3, 1800)).view(3, 1800, 13, 2)
    for kernel in (19, 11):
        y = smooth_motion(x.cuda(), kernel=kernel)
        ref = savgol_filter(x.numpy().astype(np.float64), kernel, 5, axis=1)
        err = rel_l2(y, ref)
        print(f"savgol kernel {kernel}: rel-L2 {err:.2e}")
        assert tuple(y.shape) == (3, 1800, 13, 2) and err <= 1e-5
    one = smooth_motion(x[0].cuda(), kernel=19)                     # the reference's [T, 13, 2] form
    assert torch.equal(one, smooth_motion(x.cuda(), kernel=19)[0])


def test_batched_evaluation_driver(models, tmp_path):
    """evaluate_dataset on a 5-clip dataset in the reference's on-disk format: per-clip MSE against the oracle's
    prediction for the same mel and noise, independent of the batch size (30-s clips, DDIM-25)."""
    import types
    from diffusion_conductor_amd import DDPMTrainer
    from diffusion_conductor_amd import evaluate as ev
    rng = np.random.default_rng(5)
    mels = batch_mel(5, 810)                                          # 9 s of mel -> T = 270 frames
    for i in range(5):
        d = tmp_path / f"{i:03d}"
        d.mkdir()
        np.save(d / "mel.npy", mels[i])
        np.save(d / "motion.npy", rng.standard_normal((270, 13, 2)).astype(np.float32))
    opt = types.SimpleNamespace(device=torch.device("cuda:0"), diffusion_steps=25, is_train=False)
    tr = DDPMTrainer(opt, models["fp16"])
    tr.eval_mode()
    a = ev.evaluate_dataset(tr, str(tmp_path), 26, batch_size=4, seed=11, verbose=False)
    b = ev.evaluate_dataset(tr, str(tmp_path), 26, batch_size=5, seed=11, verbose=False)
    assert list(a["per_clip"]) == [f"{i:03d}" for i in range(5)]
    p = oracle_params()
    noise = torch.stack([ev.clip_noise(11, i, 270, 26) for i in range(5)])
    with torch.no_grad():
        ref = O.generate_music_motion(p, torch.from_numpy(mels), 26, 25, noise)
    for i in range(5):
        gt = np.load(tmp_path / f"{i:03d}" / "motion.npy")
        want = float(ev.mse_loss(gt, ref[i].numpy().reshape(270, 13, 2)))
        assert abs(a["per_clip"][f"{i:03d}"] - want) <= 2e-3 * want
        assert abs(b["per_clip"][f"{i:03d}"] - a["per_clip"][f"{i:03d}"]) <= 2e-3 * want
    print(f"evaluation driver: final_mse {a['final_mse']:.5f}, {a['frames_per_s']:.0f} frames/s")


@pytest.mark.parametrize("B,T,length", [(3, 257, [257, 1, 200]), (2, 2500, [2500, 1999]), (1, 4032, [4000]), (5, 300, [300, 299, 256, 255, 31])])
def test_workgroup_record_path_shapes(B, T, length):
    """The T >= 256 path (workgroup-level records combined inside the layer kernel) at awkward shapes: clip edges inside
    workgroups, more than 9 record units per clip (T > 2304: the combine's second loop), the maximum T, ragged lengths.
    One forward against the oracle; fp16 mode (the single-forward error at a late timestep is ~3e-3, see the stage report)."""
    m = make_model("fp16")
    if T > 1800:      # sequence_embedding has num_frames rows: build a longer model from the same seeded weights
        from diffusion_conductor_amd import MotionTransformer
        from helpers import state_dict_np
        sd = {k: np.asarray(v) for k, v in state_dict_np().items()}
        reps = -(-T // 1800)
        sd["sequence_embedding"] = np.concatenate([sd["sequence_embedding"]] * reps)[:T]
        m = MotionTransformer(input_feats=26, num_frames=T, num_layers=8, latent_dim=128, device="cuda", no_clip=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        m = m.to("cuda").eval()
        p = O.to_torch_params(sd)
    else:
        p = oracle_params()
    xfp, xfo = xf_pair(B, T, first=50)
    x = torch.from_numpy(batch_noise(B, T, first=50))
    t = torch.tensor([7, 400, 49, 3, 999][:B])
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo)
    out = m(x.cuda(), t, length=torch.tensor(length), xf_proj=xfp.cuda(), xf_out=xfo.cuda())
    torch.cuda.synchronize()
    err = rel_l2(out, ref)
    print(f"B={B} T={T}: forward rel-L2 {err:.3e}")
    assert torch.isfinite(out).all() and err <= 5e-3
