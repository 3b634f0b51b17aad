#!/usr/bin/env python3
"""GPU box: do the layer launches run slower because they follow the power-limited FiLM GEMM?  Eager per-kernel passes (HIP events per
launch) of the benchmark loop (bs=32 x 1800, DDIM-50) as it is, and with the FiLM GEMM launched only once (DC_DIAG_SKIP_FILM=1: the
layers read stale tiles - results invalid, the layers' work and traffic unchanged).  usage: python tools/diag_clock_coupling.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402
from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise  # noqa: E402

B, T, S = 32, 1800, 50
dev = torch.device("cuda", 0)
model = bench.build_model("fp16", False, dev)
gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.START_X,
                       model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
xf = torch.from_numpy(batch_music_features(B, T)).to(dev)
xfp = torch.nn.functional.linear(xf, model.proj.weight, model.proj.bias).contiguous()
noise = torch.from_numpy(batch_noise(B, T)).to(dev)
nat = model.set_conditioning(xfp, xf, [T] * B)
coef = gd.native_coefficients()
nat.set_precise_tail(0)
for rep in range(3):
    for skip in (False, True):
        if skip:
            os.environ["DC_DIAG_SKIP_FILM"] = "1"
        else:
            os.environ.pop("DC_DIAG_SKIP_FILM", None)
        prof, _ = nat.profile_loop(noise, coef)
        print(("without the FiLM GEMM" if skip else "loop as it is        ") + ": " +
              ", ".join(f"{k} {v[0]:.2f} ms / {v[1]} = {1e3 * v[0] / max(v[1], 1):.1f} us" for k, v in prof.items() if v[1]), flush=True)
