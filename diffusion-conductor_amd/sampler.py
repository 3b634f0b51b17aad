"""Drop-in for the sampling half of the reference's ``GaussianDiffusion``.

Mirrors Diffusion_Stage/models/gaussian_diffusion.py: ``get_named_beta_schedule`` (:228-245),
the enums (:275-308), the constructor tables (:328-379) and the DDIM entry points
``ddim_sample`` (:783-831), ``ddim_sample_loop`` (:871-915) and
``ddim_sample_loop_progressive`` (:917-965), with the reference's argument names.

Fast path: when ``model`` is this package's MotionTransformer and no host callback
(``denoised_fn`` / ``cond_fn``) is given, the whole loop runs inside libdc_ddim.so as a replayed
hipGraph - for ``model_mean_type`` START_X (how DDPMTrainer.generate_music_motion calls it) or
EPSILON, with or without ``clip_denoised`` (the reference's default is True) and for any ``eta``
(the per-iteration noise the reference draws with ``th.randn_like`` is drawn up front, or taken
from the extension keyword ``step_noise=`` so that runs can be reproduced).  A host callback
(``denoised_fn``, ``cond_fn`` with the reference's ``condition_score``, :581-603), the
``PREVIOUS_X`` parameterisation (:510-514), a learned-variance model (2C output channels,
:472-486) or the progressive generator run the same update rule step by step with the model
call still going through the native denoiser; ``p_mean_variance`` (:442-536) is provided whole.

Training-time members (losses, VLB terms, ancestral p_sample, schedule samplers) are out of
scope for this path and are not provided.
"""
from __future__ import annotations

import os
import warnings

import enum
import math

import numpy as np
import torch as th

from . import native
from .denoiser import MotionTransformer


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """gaussian_diffusion.py:228-245."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """gaussian_diffusion.py:248-266."""
    betas = []
    for i in range(num_diffusion_timesteps):
        t1 = i / num_diffusion_timesteps
        t2 = (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self == LossType.KL or self == LossType.RESCALED_KL


def _extract(arr, timesteps, broadcast_shape):
    """_extract_into_tensor (gaussian_diffusion.py:1168-1181) without the per-call H2D copy of the
    whole table: the needed scalars are gathered on the host."""
    t = timesteps.detach().cpu().numpy()
    res = th.from_numpy(np.asarray(arr)[t]).float().to(timesteps.device)
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        # posterior q(x_{t-1} | x_t, x_0) (gaussian_diffusion.py:363-379): read by the PREVIOUS_X parameterisation, the learned-range
        # variance and p_mean_variance's "mean" - never by the captured loop
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:])) \
            if self.num_timesteps > 1 else np.log(np.maximum(self.posterior_variance, 1e-20))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self._native_coef = None

    # ---- helpers ------------------------------------------------------------------------
    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return (_extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart) / \
            _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def _predict_xstart_from_eps(self, x_t, t, eps):
        return _extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - \
            _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps

    def _predict_xstart_from_xprev(self, x_t, t, xprev):
        """gaussian_diffusion.py:545-553: (xprev - coef2 * x_t) / coef1."""
        assert x_t.shape == xprev.shape
        return _extract(1.0 / self.posterior_mean_coef1, t, x_t.shape) * xprev - \
            _extract(self.posterior_mean_coef2 / self.posterior_mean_coef1, t, x_t.shape) * x_t

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """gaussian_diffusion.py:418-440."""
        assert x_start.shape == x_t.shape
        mean = _extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + _extract(self.posterior_mean_coef2, t, x_t.shape) * x_t
        return mean, _extract(self.posterior_variance, t, x_t.shape), _extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def _model_output(self, model, x, t, model_kwargs):
        """The model call of p_mean_variance (gaussian_diffusion.py:466-476): returns (mean-parameter output, variance values or None).
        A learned-variance model emits 2C channels along dim 1, split as th.split(model_output, C, dim=1) (:474)."""
        if model_kwargs is None:
            model_kwargs = {}
        B, C = x.shape[:2]
        assert t.shape == (B,)
        model_output = model(x, self._scale_timesteps(t), **model_kwargs)
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            assert model_output.shape == (B, 2 * C, *x.shape[2:])
            return th.split(model_output, C, dim=1)
        return model_output, None

    def _xstart_of(self, model_output, x, t, clip_denoised, denoised_fn):
        """pred_xstart of p_mean_variance (gaussian_diffusion.py:497-521) for the three parameterisations."""
        if self.model_mean_type == ModelMeanType.PREVIOUS_X:
            pred = self._predict_xstart_from_xprev(x_t=x, t=t, xprev=model_output)
        elif self.model_mean_type == ModelMeanType.START_X:
            pred = model_output
        elif self.model_mean_type == ModelMeanType.EPSILON:
            pred = self._predict_xstart_from_eps(x_t=x, t=t, eps=model_output)
        else:
            raise NotImplementedError(self.model_mean_type)
        if denoised_fn is not None:
            pred = denoised_fn(pred)
        if clip_denoised:
            pred = pred.clamp(-1, 1)
        assert pred.shape == x.shape
        return pred

    def _pred_xstart(self, model, x, t, clip_denoised, denoised_fn, model_kwargs):
        """The part of p_mean_variance (gaussian_diffusion.py:442-536) DDIM reads: pred_xstart."""
        model_output, _ = self._model_output(model, x, t, model_kwargs)
        return self._xstart_of(model_output, x, t, clip_denoised, denoised_fn)

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """gaussian_diffusion.py:442-536, whole: {"mean", "variance", "log_variance", "pred_xstart"}.  (DDIM reads pred_xstart only:
        ddim_sample takes the short way through _pred_xstart.)"""
        model_output, var_values = self._model_output(model, x, t, model_kwargs)
        if self.model_var_type == ModelVarType.LEARNED:
            log_variance = var_values
            variance = th.exp(log_variance)
        elif self.model_var_type == ModelVarType.LEARNED_RANGE:
            min_log = _extract(self.posterior_log_variance_clipped, t, x.shape)
            max_log = _extract(np.log(self.betas), t, x.shape)
            frac = (var_values + 1) / 2                       # the model's [-1, 1] covers [min_var, max_var]
            log_variance = frac * max_log + (1 - frac) * min_log
            variance = th.exp(log_variance)
        else:
            var, logvar = {
                ModelVarType.FIXED_LARGE: (np.append(self.posterior_variance[1], self.betas[1:]),
                                           np.log(np.append(self.posterior_variance[1], self.betas[1:]))),
                ModelVarType.FIXED_SMALL: (self.posterior_variance, self.posterior_log_variance_clipped),
            }[self.model_var_type]
            variance, log_variance = _extract(var, t, x.shape), _extract(logvar, t, x.shape)
        pred_xstart = self._xstart_of(model_output, x, t, clip_denoised, denoised_fn)
        if self.model_mean_type == ModelMeanType.PREVIOUS_X:
            mean = model_output
        else:
            mean, _, _ = self.q_posterior_mean_variance(x_start=pred_xstart, x_t=x, t=t)
        assert mean.shape == log_variance.shape == pred_xstart.shape == x.shape
        return {"mean": mean, "variance": variance, "log_variance": log_variance, "pred_xstart": pred_xstart}

    def condition_score(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """gaussian_diffusion.py:581-603 (Song et al. 2020): the prediction the model would have made had its score been conditioned
        by cond_fn = grad log p(y | x).  "mean" is recomputed only when the input carries one (DDIM does not read it)."""
        alpha_bar = _extract(self.alphas_cumprod, t, x.shape)
        eps = self._predict_eps_from_xstart(x, t, p_mean_var["pred_xstart"])
        eps = eps - (1 - alpha_bar).sqrt() * cond_fn(x, self._scale_timesteps(t), **(model_kwargs or {}))
        out = dict(p_mean_var)
        out["pred_xstart"] = self._predict_xstart_from_eps(x, t, eps)
        if "mean" in out:
            out["mean"], _, _ = self.q_posterior_mean_variance(x_start=out["pred_xstart"], x_t=x, t=t)
        return out

    # ---- DDIM ---------------------------------------------------------------------------
    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    eta=0.0, noise=None):
        """gaussian_diffusion.py:783-831.  `noise` (extension): the draw to use in place of th.randn_like(x)."""
        pred_xstart = self._pred_xstart(model, x, t, clip_denoised, denoised_fn, model_kwargs)
        if cond_fn is not None:
            pred_xstart = self.condition_score(cond_fn, {"pred_xstart": pred_xstart}, x, t, model_kwargs=model_kwargs)["pred_xstart"]
        eps = self._predict_eps_from_xstart(x, t, pred_xstart)
        alpha_bar = _extract(self.alphas_cumprod, t, x.shape)
        alpha_bar_prev = _extract(self.alphas_cumprod_prev, t, x.shape)
        sigma = eta * th.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar)) * th.sqrt(1 - alpha_bar / alpha_bar_prev)
        mean_pred = pred_xstart * th.sqrt(alpha_bar_prev) + th.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
        sample = mean_pred
        if eta != 0.0:
            noise = th.randn_like(x) if noise is None else noise.to(x)
            nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
            sample = mean_pred + nonzero_mask * sigma * noise
        return {"sample": sample, "pred_xstart": pred_xstart}

    def _refuse_epsilon_full_attention(self, model, eta):
        """EPSILON model x full attention (`no_eff`) x eta = 0 is outside the 1e-3 parity bound on some loops (2.3e-4 ... 1.26e-3 over 14
        randomized loops of tools/fuzz_sampler.py: the attention's scores, weights and values stay plain fp16 even in the split
        evaluations, and a deterministic EPSILON chain keeps every evaluation's error in x_t).  Refused instead of returned; the library
        refuses the captured loop likewise (DC_ERR_UNSUPPORTED)."""
        import os
        if (isinstance(model, MotionTransformer) and getattr(getattr(model, "cfg", None), "no_eff", False) and self.model_mean_type == ModelMeanType.EPSILON
                and float(eta) == 0.0 and not os.environ.get("DC_ALLOW_EPSILON_NO_EFF_ETA0")):
            raise ValueError("an EPSILON model with full attention (no_eff) at eta = 0 is outside the 1e-3 parity bound of this "
                             "implementation (up to 1.3e-3); use linear attention, or eta > 0")

    def _fast_path_ok(self, model, denoised_fn, cond_fn):
        return (isinstance(model, MotionTransformer)
                and self.model_mean_type in (ModelMeanType.START_X, ModelMeanType.EPSILON)
                and self.model_var_type in (ModelVarType.FIXED_SMALL, ModelVarType.FIXED_LARGE)
                and denoised_fn is None and cond_fn is None and not self.rescale_timesteps)

    def native_coefficients(self, eta=None):
        """eta None: the [S, 4] table of dc_ddim_coefficients (eta = 0); a float: the [S, 8] table of dc_ddim_coefficients_ex."""
        key = None if eta is None else float(eta)
        if self._native_coef is None:
            self._native_coef = {}
        if key not in self._native_coef:
            self._native_coef[key] = native.ddim_coefficients(self.alphas_cumprod, key)
        return self._native_coef[key]

    def _native_loop(self, model, img, mk, clip_denoised, eta, snap, step_noise, smooth=None, step_noise_seed=None):
        """The captured loop on `model`'s sampler; returns (out, snaps).  Numeric health is checked once per call
        (`model.check_numerics`): a non-finite x0 under precision="auto" falls back to the bf16-range mode in a fresh sampler."""
        flags = (native.UPDATE_CLIP_DENOISED if clip_denoised else 0) | \
            (native.UPDATE_EPSILON if self.model_mean_type == ModelMeanType.EPSILON else 0)
        z, zseed = None, None
        if eta != 0.0:
            if step_noise is not None:
                shape = (self.num_timesteps,) + tuple(img.shape)
                z = step_noise.to(device=img.device, dtype=th.float32).contiguous()
                assert tuple(z.shape) == shape, f"step_noise must be {shape}"
            elif step_noise_seed is not None:
                zseed = step_noise_seed
            else:
                # the reference draws th.randn_like(x) once per step (gaussian_diffusion.py:822); so does the library, at the head
                # of each step, from a seed taken off torch's default generator (torch.manual_seed makes a run reproducible) -
                # never the whole [S, B, T, P] tensor up front (6 GB at S = 1000, bs = 32).  Ranks of a sharded run that seed
                # torch alike must pass step_noise_seed=(seed, lo*T*P) instead, or every shard would add the same draws.
                zseed = int(th.randint(0, 2 ** 62, (1,)).item())
        plain = flags == 0 and eta == 0.0
        # (An EPSILON model's final sample carries what the evaluations left in x_t: the library runs EVERY evaluation of such a loop on
        # split operands - in the fp16 and bf16 precisions, linear and (fp16) full attention alike (dc_ddim.h, dc_sampler_set_precise_tail; the bf16
        # precision's split evaluations take the FiLM GEMM's operands in fp16).)
        coef = self.native_coefficients(None if plain else eta)
        retried = False
        while True:
            nat = model.set_conditioning(mk["xf_proj"], mk["xf_out"], mk.get("length"))
            nat.set_smoothing(*(smooth if smooth else (0, 0)))
            out, snaps = nat.ddim_loop(img, coef, snap, flags, z, zseed)
            if not getattr(model, "check_numerics", True):
                return out, snaps
            st = nat.status()
            if st == 0:
                return out, snaps
            if (st & native.STATUS_TIMEOUT) and not retried:
                # small batches: the clip's workgroups exchange their combine slices inside a layer launch and one of them gave up
                # waiting (the GPU is shared with other work, so they were not co-resident).  Whatever else that void run reported
                # does not count; reading the status has latched the form without the exchange on this sampler: run the loop again
                warnings.warn("libdc_ddim: in-launch combine exchange timed out (GPU shared?); this sampler continues without the exchange")
                retried = True
                continue
            if not model.numerics_fallback(st):
                raise FloatingPointError(native.describe_status(st, model.active_precision))

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, idxs=[], step_noise=None, smooth=None,
                         step_noise_seed=None):
        """gaussian_diffusion.py:871-915.  Returns the final sample, or when `idxs` is non-empty a
        dict {iteration: sample} for the listed iterations plus {num_timesteps: final}.
        `step_noise` (extension, eta > 0): [S, B, T, P], the draw for iteration i in place of th.randn_like.
        `step_noise_seed` (extension, eta > 0, native loop only): seed of the library's own per-step draws, or (seed, first_element)
        for a shard that holds clips [lo, hi) of a larger batch (first_element = lo*T*P: the rows the whole batch's draw gives them).
        `smooth` (extension): (window, order) of the Savitzky-Golay filter tools/visualization.py:126 applies to the result,
        folded into the loop's final write (native loop only)."""
        self._refuse_epsilon_full_attention(model, eta)
        if self._fast_path_ok(model, denoised_fn, cond_fn):
            if device is None:
                device = next(model.parameters()).device
            assert isinstance(shape, (tuple, list))
            img = noise if noise is not None else th.randn(*shape, device=device)
            img = img.to(device=device, dtype=th.float32).contiguous()
            mk = model_kwargs or {}
            if mk.get("xf_proj") is None or mk.get("xf_out") is None:
                mk = dict(mk)
                mk["xf_proj"], mk["xf_out"] = model.encode_music(mk["text"], device)
            snap = sorted(int(i) for i in set(idxs) if 0 <= int(i) < self.num_timesteps)
            out, snaps = self._native_loop(model, img, mk, bool(clip_denoised), float(eta), snap, step_noise, smooth, step_noise_seed)
            if len(idxs) == 0:
                return out
            result = {it: snaps[k] for k, it in enumerate(snap)}
            result[self.num_timesteps] = out
            return result
        if smooth:
            raise NotImplementedError("smooth= is folded into the native loop's final write; with host callbacks apply evaluate.smooth_motion")
        final, i, result = None, 0, {}
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                        denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                        model_kwargs=model_kwargs, device=device,
                                                        progress=progress, eta=eta, step_noise=step_noise):
            final = sample
            if i in idxs:
                result[i] = sample["sample"]
            i += 1
        result[i] = final["sample"]
        if len(idxs) == 0:
            return final["sample"]
        return result

    # the north-star wording calls the entry point `GaussianDiffusion.sample`
    sample = ddim_sample_loop

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0,
                                     step_noise=None):
        """gaussian_diffusion.py:917-965: yields {"sample","pred_xstart"} after every step."""
        self._refuse_epsilon_full_attention(model, eta)
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        # The update of an EPSILON / PREVIOUS_X model, and a conditioned score, keep what the evaluations' 16-bit operands left in x_t
        # (fp16, EPSILON, eta = 0: 1.5e-3 on plain operands): these loops evaluate the native denoiser on split operands
        precise = isinstance(model, MotionTransformer) and (self.model_mean_type != ModelMeanType.START_X or cond_fn is not None)
        before = model.precise_forward if precise else None
        if precise:
            model.precise_forward = True
        try:
            for it, i in enumerate(indices):
                t = th.full((shape[0],), i, device=device, dtype=th.long)
                with th.no_grad():
                    out = self.ddim_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                           cond_fn=cond_fn, model_kwargs=model_kwargs, eta=eta,
                                           noise=None if step_noise is None else step_noise[it])
                    yield out
                    img = out["sample"]
        finally:
            if precise:
                model.precise_forward = before
