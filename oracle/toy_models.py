"""TEST INFRASTRUCTURE (never imported by the product).  Small deterministic stand-ins for `model` and `cond_fn`, shared by
oracle/make_golden.py (which runs the imported reference's GaussianDiffusion on them, fixture G11) and tests/ (which run this
package's sampler on them): the sampler's off-path branches - ModelMeanType.PREVIOUS_X (gaussian_diffusion.py:510-514), the
learned-variance output split (:472-486) and cond_fn / condition_score (:581-603, :806-808) - are host arithmetic around the model call,
so they are pinned with callables that need no weights."""
import torch as th

SHAPE = (3, 8, 5)          # (N, C, ...) in the reference's naming: the learned-variance split is along dim 1


def toy_model(x, t, scale=None, **kw):
    """[N, C, W] -> [N, C, W], bounded, depends on x, t and a model_kwargs entry."""
    s = 1.0 if scale is None else scale
    return 0.9 * th.tanh(0.7 * x.roll(1, dims=1) - 0.2 * x + 0.05 * t.view(-1, 1, 1).float()) * s


def toy_model_learned(x, t, scale=None, **kw):
    """[N, C, W] -> [N, 2C, W]: the mean parameter | the variance values in [-1, 1] (ModelVarType.LEARNED / LEARNED_RANGE)."""
    return th.cat([toy_model(x, t, scale=scale), 0.8 * th.sin(1.3 * x + 0.1 * t.view(-1, 1, 1).float())], dim=1)


def toy_cond_fn(x, t, scale=None, **kw):
    """grad log p(y | x) of a Gaussian 'classifier' centred at 0.5 whose width shrinks with t."""
    return -(x - 0.5) * (0.15 + 0.002 * t.view(-1, 1, 1).float())


def pose_cond_fn(x, t, **kw):
    """cond_fn for the [B, T, P] pose path (G11b): pulls every pose feature towards a fixed smooth trajectory."""
    T = x.shape[1]
    target = 0.4 * th.sin(th.arange(T, dtype=th.float32, device=x.device) * 0.05).view(1, T, 1)
    return -0.5 * (x - target)
