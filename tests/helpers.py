"""Shared test utilities (tests may import oracle/; the product may not)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ddim_oracle as O  # noqa: E402
from diffusion_conductor_amd.param_spec import DenoiserConfig  # noqa: E402
from diffusion_conductor_amd.synthetic import (batch_mel, batch_music_features, batch_noise,  # noqa: E402,F401
                                               synthetic_state_dict)

GOLDEN = os.path.join(ROOT, "tests", "golden")
_cache = {}


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def state_dict_np():
    if "sd" not in _cache:
        _cache["sd"] = synthetic_state_dict(DenoiserConfig(), seed=0)
    return _cache["sd"]


def oracle_params(dtype=torch.float32):
    key = ("p", dtype)
    if key not in _cache:
        _cache[key] = O.to_torch_params(state_dict_np(), dtype)
    return _cache[key]


def xf_pair(B, T, first=0):
    """(xf_proj, xf_out) as encode_music would return them, from seeded stand-in features."""
    p = oracle_params()
    xf = torch.from_numpy(batch_music_features(B, T, first=first))
    return torch.nn.functional.linear(xf, p["proj.weight"], p["proj.bias"]), xf


def make_model(precision="fp16", device="cuda", no_eff=False):
    from diffusion_conductor_amd import MotionTransformer
    m = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device=device,
                          no_clip=True, precision=precision, no_eff=no_eff)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np().items()}, strict=True)
    return m.to(device).eval()


def make_diffusion(S):
    from diffusion_conductor_amd.sampler import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,
                                                 get_named_beta_schedule)
    return GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.START_X,
                             model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
