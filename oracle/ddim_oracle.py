"""CPU ORACLE for the Diffusion-Conductor DDIM sampler path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.  The product path (diffusion-conductor_amd/) never
does: it fails loudly when its HIP library is missing.

What this is: a plain functional restatement, in PyTorch CPU ops on a dict of named
tensors, of the reference's sampler path.  Each function cites the reference lines it
follows (paths relative to /root/reference/Diffusion_Stage).  It is "eager-faithful":
same op order as the reference, nothing hoisted or fused, fp32 by default (fp64 on
request to separate rounding noise from real differences).

Parity pinning: the reference has no tests or golden vectors for this path
(SURVEY.md §4), so this oracle is pinned against the *imported reference itself* in
the build container by ``oracle/make_golden.py`` (bit-level agreement is checked
there) and the resulting vectors are committed under ``tests/golden/``.

``Emu`` optionally rounds GEMM operands the way the HIP kernels do (bf16, split-bf16)
so that the error budget of a precision mode can be measured on CPU.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Schedule tables  (models/gaussian_diffusion.py)
# --------------------------------------------------------------------------------------


def linear_beta_schedule(num_steps: int) -> np.ndarray:
    """get_named_beta_schedule('linear', n)  (gaussian_diffusion.py:228-245)."""
    scale = 1000 / num_steps
    return np.linspace(scale * 0.0001, scale * 0.02, num_steps, dtype=np.float64)


def ddim_tables(betas: np.ndarray) -> dict:
    """The fp64 tables GaussianDiffusion.__init__ builds (gaussian_diffusion.py:342-361)
    that the DDIM path reads."""
    betas = np.asarray(betas, dtype=np.float64)
    ac = np.cumprod(1.0 - betas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
    }


def ddim_step_coefficients(tables: dict, eta: float = 0.0) -> np.ndarray:
    """Per-timestep fp32 scalars exactly as ddim_sample evaluates them
    (gaussian_diffusion.py:812-830): `_extract_into_tensor` casts the fp64 table entry to
    fp32 (`.float()`, :1178) and all further arithmetic is fp32.

    Returns [S, 5] float32: (sqrt_recip, sqrt_recipm1, sqrt(abar_prev), sqrt(1-abar_prev-sigma^2), sigma).
    """
    # torch ops on fp32 tensors, as the reference does (th.sqrt of a .float() tensor); numpy's
    # sqrtf differs from torch's vectorised CPU sqrt by 1 ulp on a handful of entries.
    f = lambda k: torch.from_numpy(tables[k]).float()
    a, ap = f("alphas_cumprod"), f("alphas_cumprod_prev")
    sigma = eta * torch.sqrt((1 - ap) / (1 - a)) * torch.sqrt(1 - a / ap)
    co = torch.stack([f("sqrt_recip_alphas_cumprod"), f("sqrt_recipm1_alphas_cumprod"), torch.sqrt(ap),
                      torch.sqrt(1 - ap - sigma ** 2), sigma], dim=1)
    return co.numpy().astype(np.float32)


# --------------------------------------------------------------------------------------
# Precision emulation of the HIP path's MFMA operands (fp32 accumulate)
# --------------------------------------------------------------------------------------


def _bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


def _f16(x):
    return x.to(torch.float16).to(x.dtype)


class Emu:
    """Operand rounding per GEMM class.

    mode "fp32":  no rounding (the oracle proper).
    mode "bf16":  every GEMM a·w -> bf16(a)·bf16(w).
    mode "mixed": the K=512 FiLM GEMM (emb_layers) on fp16 operands; all other GEMMs
                  split-bf16 (a_hi·w_hi + a_lo·w_hi + a_hi·w_lo) - the HIP default.
    mode "x3":    split-bf16 everywhere.
    mode "fp16":  every GEMM fp16 operands.
    FiLM outputs (scale|shift) are additionally rounded to fp16 when `film_store_f16`
    (the HIP path stores them as fp16 between kernels).
    """

    def __init__(self, mode="fp32", film_store_f16=None):
        assert mode in ("fp32", "bf16", "mixed", "x3", "fp16")
        self.mode = mode
        self.film_store_f16 = (mode != "fp32") if film_store_f16 is None else film_store_f16

    def _kind(self, big):
        if self.mode in ("fp32", "bf16", "x3", "fp16"):
            return self.mode
        return "fp16" if big else "x3"

    def matmul(self, a, w_t, big=False):
        """a [..., K] @ w_t [K, N] with emulated operand rounding."""
        k = self._kind(big)
        if k == "fp32":
            return a @ w_t
        if k == "bf16":
            return _bf16(a) @ _bf16(w_t)
        if k == "fp16":
            return _f16(a) @ _f16(w_t)
        ah, wh = _bf16(a), _bf16(w_t)
        al, wl = _bf16(a - ah), _bf16(w_t - wh)
        return ah @ wh + al @ wh + ah @ wl

    def linear(self, x, w, b, big=False):
        if self._kind(big) == "fp32":
            return F.linear(x, w, b)          # same fused op the reference's nn.Linear calls
        y = self.matmul(x, w.t(), big)
        return y + b if b is not None else y

    def einsum(self, eq, a, b):
        k = self._kind(False)
        if k == "fp32":
            return torch.einsum(eq, a, b)
        if k == "bf16":
            return torch.einsum(eq, _bf16(a), _bf16(b))
        if k == "fp16":
            return torch.einsum(eq, _f16(a), _f16(b))
        ah, bh = _bf16(a), _bf16(b)
        al, bl = _bf16(a - ah), _bf16(b - bh)
        return torch.einsum(eq, ah, bh) + torch.einsum(eq, al, bh) + torch.einsum(eq, ah, bl)


FP32 = Emu("fp32")

# --------------------------------------------------------------------------------------
# Denoiser blocks  (models/transformer.py)
# --------------------------------------------------------------------------------------


def timestep_embedding(timesteps, dim, max_period=10000):
    """transformer.py:8-25."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _ln(x, p, prefix, eps=1e-5):
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def stylization(p, prefix, h, emb, emu=FP32):
    """StylizationBlock.forward (transformer.py:68-81); dropout is identity in eval."""
    e = emu.linear(F.silu(emb), p[prefix + ".emb_layers.1.weight"], p[prefix + ".emb_layers.1.bias"], big=True)
    if emu.film_store_f16:
        e = _f16(e)
    scale, shift = torch.chunk(e, 2, dim=2)
    h = _ln(h, p, prefix + ".norm") * (1 + scale) + shift
    return emu.linear(F.silu(h), p[prefix + ".out_layers.2.weight"], p[prefix + ".out_layers.2.bias"])


def linear_self_attention(p, prefix, x, emb, src_mask, H, emu=FP32):
    """LinearTemporalSelfAttention.forward (transformer.py:96-123)."""
    B, T, D = x.shape
    n = _ln(x, p, prefix + ".norm")
    query = emu.linear(n, p[prefix + ".query.weight"], p[prefix + ".query.bias"])
    key = emu.linear(n, p[prefix + ".key.weight"], p[prefix + ".key.bias"]) + (1 - src_mask) * -1000000
    query = F.softmax(query.view(B, T, H, -1), dim=-1)
    key = F.softmax(key.view(B, T, H, -1), dim=1)
    value = (emu.linear(n, p[prefix + ".value.weight"], p[prefix + ".value.bias"]) * src_mask).view(B, T, H, -1)
    attention = emu.einsum('bnhd,bnhl->bhdl', key, value)
    y = emu.einsum('bnhd,bhdl->bnhl', query, attention).reshape(B, T, D)
    return x + stylization(p, prefix + ".proj_out", y, emb, emu)


def linear_cross_attention(p, prefix, x, xf, emb, H, emu=FP32):
    """LinearTemporalCrossAttention.forward (transformer.py:138-158)."""
    B, T, D = x.shape
    N = xf.shape[1]
    query = emu.linear(_ln(x, p, prefix + ".norm"), p[prefix + ".query.weight"], p[prefix + ".query.bias"])
    tn = _ln(xf, p, prefix + ".text_norm")
    key = emu.linear(tn, p[prefix + ".key.weight"], p[prefix + ".key.bias"], big=True)
    query = F.softmax(query.view(B, T, H, -1), dim=-1)
    key = F.softmax(key.view(B, N, H, -1), dim=1)
    value = emu.linear(tn, p[prefix + ".value.weight"], p[prefix + ".value.bias"], big=True).view(B, N, H, -1)
    attention = emu.einsum('bnhd,bnhl->bhdl', key, value)
    y = emu.einsum('bnhd,bhdl->bnhl', query, attention).reshape(B, T, D)
    return x + stylization(p, prefix + ".proj_out", y, emb, emu)


def full_self_attention(p, prefix, x, emb, src_mask, H, emu=FP32):
    """TemporalSelfAttention.forward (transformer.py:210-229), the --no_eff variant.
    Note the mask is added along the *query* axis (`src_mask.unsqueeze(-1)` on a
    [B,T,T,H] tensor), a no-op under the softmax over keys; V is not masked."""
    B, T, D = x.shape
    n = _ln(x, p, prefix + ".norm")
    query = emu.linear(n, p[prefix + ".query.weight"], p[prefix + ".query.bias"]).view(B, T, H, -1)
    key = emu.linear(n, p[prefix + ".key.weight"], p[prefix + ".key.bias"]).view(B, T, H, -1)
    attention = emu.einsum('bnhd,bmhd->bnmh', query, key) / math.sqrt(D // H)
    attention = attention + (1 - src_mask.unsqueeze(-1)) * -100000
    weight = F.softmax(attention, dim=2)
    value = emu.linear(n, p[prefix + ".value.weight"], p[prefix + ".value.bias"]).view(B, T, H, -1)
    y = emu.einsum('bnmh,bmhd->bnhd', weight, value).reshape(B, T, D)
    return x + stylization(p, prefix + ".proj_out", y, emb, emu)


def full_cross_attention(p, prefix, x, xf, emb, H, emu=FP32):
    """TemporalCrossAttention.forward (transformer.py:244-264)."""
    B, T, D = x.shape
    N = xf.shape[1]
    query = emu.linear(_ln(x, p, prefix + ".norm"), p[prefix + ".query.weight"], p[prefix + ".query.bias"]).view(B, T, H, -1)
    tn = _ln(xf, p, prefix + ".text_norm")
    key = emu.linear(tn, p[prefix + ".key.weight"], p[prefix + ".key.bias"], big=True).view(B, N, H, -1)
    attention = emu.einsum('bnhd,bmhd->bnmh', query, key) / math.sqrt(D // H)
    weight = F.softmax(attention, dim=2)
    value = emu.linear(tn, p[prefix + ".value.weight"], p[prefix + ".value.bias"], big=True).view(B, N, H, -1)
    y = emu.einsum('bnmh,bmhd->bnhd', weight, value).reshape(B, T, D)
    return x + stylization(p, prefix + ".proj_out", y, emb, emu)


def ffn(p, prefix, x, emb, emu=FP32):
    """FFN.forward (transformer.py:170-173); nn.GELU() is the exact-erf form."""
    y = emu.linear(F.gelu(emu.linear(x, p[prefix + ".linear1.weight"], p[prefix + ".linear1.bias"])),
                   p[prefix + ".linear2.weight"], p[prefix + ".linear2.bias"])
    return x + stylization(p, prefix + ".proj_out", y, emb, emu)


def generate_src_mask(T, length):
    """MotionTransformer.generate_src_mask (transformer.py:461-467)."""
    length = torch.as_tensor(length)
    return (torch.arange(T)[None, :] < length[:, None]).float()


def denoiser_forward(p, x, timesteps, length, xf_proj, xf_out, num_layers=8, num_heads=8,
                     no_eff=False, emu=FP32, taps=None):
    """MotionTransformer.forward (transformer.py:469-497).  `taps`, if a dict, receives
    intermediate activations for block-level known-answer tests."""
    B, T = x.shape[0], x.shape[1]
    D = p["joint_embed.weight"].shape[0]
    xf_proj = emu.linear(xf_proj, p["linear.weight"], p["linear.bias"])
    xf_out = emu.linear(xf_out, p["linear.weight"], p["linear.bias"])
    te = timestep_embedding(timesteps, D).to(x.dtype)
    te = emu.linear(F.silu(emu.linear(te, p["time_embed.0.weight"], p["time_embed.0.bias"])),
                    p["time_embed.2.weight"], p["time_embed.2.bias"])
    emb = te.unsqueeze(1) + xf_proj
    h = emu.linear(x, p["joint_embed.weight"], p["joint_embed.bias"])
    h = h + p["sequence_embedding"].unsqueeze(0)[:, :T, :]
    src_mask = generate_src_mask(T, length).to(x.dtype).unsqueeze(-1)
    if taps is not None:
        taps["emb"] = emb
        taps["h0"] = h
    for i in range(num_layers):
        pre = f"temporal_decoder_blocks.{i}"
        if no_eff:
            h = full_self_attention(p, pre + ".sa_block", h, emb, src_mask, num_heads, emu)
            h = full_cross_attention(p, pre + ".ca_block", h, xf_out, emb, num_heads, emu)
        else:
            h = linear_self_attention(p, pre + ".sa_block", h, emb, src_mask, num_heads, emu)
            if taps is not None:
                taps[f"sa{i}"] = h
            h = linear_cross_attention(p, pre + ".ca_block", h, xf_out, emb, num_heads, emu)
            if taps is not None:
                taps[f"ca{i}"] = h
        h = ffn(p, pre + ".ffn", h, emb, emu)
        if taps is not None:
            taps[f"ffn{i}"] = h
    return emu.linear(h, p["out.weight"], p["out.bias"]).view(B, T, -1).contiguous()


# --------------------------------------------------------------------------------------
# Music encoder  (models/transformer.py:289-340, 447-459)
# --------------------------------------------------------------------------------------


def _bn(x, p, prefix, eps=1e-5):
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], False, 0.0, eps)


def _conv_res_layer(p, prefix, x, residual):
    """Conv2dResLayer.forward (transformer.py:308-311): reflect-padded 3x3 conv + BN +
    ReLU, plus identity / 1x1-conv+BN / no residual."""
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), p[prefix + ".conv2d_layer.0.weight"],
                 p[prefix + ".conv2d_layer.0.bias"])
    y = F.relu(_bn(y, p, prefix + ".conv2d_layer.1"))
    if residual == "none":
        return y
    if residual == "identity":
        return y + x
    r = F.conv2d(x, p[prefix + ".residual.0.weight"], p[prefix + ".residual.0.bias"])
    return y + _bn(r, p, prefix + ".residual.1")


def music_encoder(p, mel, prefix="music_encoder"):
    """MusicEncoder.forward (transformer.py:330-340): mel [B,Tm,128] -> [B,Tm/3,64]."""
    x = mel.unsqueeze(1)
    x = _conv_res_layer(p, prefix + ".conv1.0", x, "none")
    x = _conv_res_layer(p, prefix + ".conv1.1", x, "identity")
    x = _conv_res_layer(p, prefix + ".conv1.2", x, "identity")
    x = F.max_pool2d(x, (5, 5), (1, 2), (2, 2))
    x = _conv_res_layer(p, prefix + ".conv2.0", x, "conv")
    x = _conv_res_layer(p, prefix + ".conv2.1", x, "identity")
    x = F.max_pool2d(x, (5, 5), (3, 2), (2, 2))
    x = _conv_res_layer(p, prefix + ".conv3.0", x, "identity")
    x = _conv_res_layer(p, prefix + ".conv3.1", x, "identity")
    x = F.max_pool2d(x, (3, 3), (1, 2), (1, 1))
    x = x.transpose(1, 2).flatten(start_dim=2).transpose(1, 2)          # [B, 512, T]
    x = F.conv1d(x, p[prefix + ".conv4.0.weight"], p[prefix + ".conv4.0.bias"])
    x = _bn(x, p, prefix + ".conv4.1")
    return x.transpose(1, 2)


def encode_music(p, mel):
    """MotionTransformer.encode_music in eval mode (transformer.py:447-459)."""
    x = music_encoder(p, mel)
    return F.linear(x, p["proj.weight"], p["proj.bias"]), x


# --------------------------------------------------------------------------------------
# DDIM loop  (models/gaussian_diffusion.py:783-965)
# --------------------------------------------------------------------------------------


def ddim_sample_loop(p, noise, xf_proj, xf_out, length, num_steps, num_layers=8, num_heads=8,
                     no_eff=False, eta=0.0, idxs=(), emu=FP32, progress=None, clip_denoised=False,
                     eps_model=False, step_noise=None, return_pred=False):
    """ddim_sample_loop / _progressive / ddim_sample (gaussian_diffusion.py:783-831, 871-965) on top of
    p_mean_variance's pred_xstart (:503-521): model_mean_type START_X (the harness, `eps_model=False`) or
    EPSILON (`eps_model=True`: pred = sqrt(1/abar) x_t - sqrt(1/abar - 1) model_out, :539-544), then
    `clip_denoised` (x.clamp(-1, 1), :506-507).  With eta == 0 the noise term is multiplied by zero and no
    RNG is consumed; with eta > 0 iteration `it` adds nonzero_mask * sigma * step_noise[it] (:822-830; the
    reference draws th.randn_like(x) there - the caller supplies the same draws to compare).
    `return_pred`: also return the list of per-iteration pred_xstart (what the progressive generator yields)."""
    if eta != 0.0:
        assert step_noise is not None and len(step_noise) == num_steps, "eta > 0 needs the per-iteration noise [S,B,T,P]"
    co = torch.from_numpy(ddim_step_coefficients(ddim_tables(linear_beta_schedule(num_steps)), eta)).to(noise.dtype)
    img = noise
    B = noise.shape[0]
    result = {}
    preds = []
    it = 0
    for i in reversed(range(num_steps)):
        t = torch.tensor([i] * B)
        out = denoiser_forward(p, img, t, length, xf_proj, xf_out, num_layers, num_heads, no_eff, emu)
        sr, srm1, c_x0, c_eps, sigma = co[i]
        x0 = sr * img - srm1 * out if eps_model else out
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        eps = (sr * img - x0) / srm1
        mean = x0 * c_x0 + c_eps * eps
        if eta != 0.0:
            mask = 0.0 if i == 0 else 1.0
            img = mean + mask * sigma * torch.as_tensor(step_noise[it]).to(noise.dtype)
        else:
            img = mean
        preds.append(x0)
        if it in idxs:
            result[it] = img
        it += 1
        if progress is not None:
            progress(it)
    if len(idxs) == 0:
        return (img, preds) if return_pred else img
    result[it] = img
    return (result, preds) if return_pred else result


def generate_music_motion(p, mel, dim_pose, num_steps, noise, num_layers=8, num_heads=8, no_eff=False):
    """DDPMTrainer.generate_music_motion (trainers/ddpm_trainer.py:183-201), batched."""
    mel = torch.as_tensor(mel)
    if mel.dim() == 2:
        mel = mel.unsqueeze(0)
    xf_proj, xf_out = encode_music(p, mel)
    B, T = mel.shape[0], xf_proj.shape[1]
    assert noise.shape == (B, T, dim_pose)
    return ddim_sample_loop(p, noise, xf_proj, xf_out, [T] * B, num_steps, num_layers, num_heads, no_eff)


def to_torch_params(state_dict_np, dtype=torch.float32):
    """numpy state_dict -> torch tensors (float entries cast to `dtype`)."""
    out = {}
    for k, v in state_dict_np.items():
        t = torch.from_numpy(np.asarray(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out
