#!/usr/bin/env python3
"""GPU box: sha256 of the DDIM-50 result of the benchmark batch (bs x 1800, seeded synthetic inputs) for the library selected by
DC_DDIM_LIB - two builds whose lines agree produce bit-identical poses.  usage: python tools/ab_equal.py [bs]
(DC_NO_EFF=1: the full-attention kernels; DC_RAGGED=1: clip lengths T, T - 37, T - 74, ... instead of T everywhere)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402
from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise  # noqa: E402

B, T, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 1800, 50
dev = torch.device("cuda", 0)
model = bench.build_model(os.environ.get("DC_PREC", "fp16"), bool(os.environ.get("DC_NO_EFF")), dev)
gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.START_X,
                       model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
xf = torch.from_numpy(batch_music_features(B, T)).to(dev)
xfp = torch.nn.functional.linear(xf, model.proj.weight, model.proj.bias).contiguous()
noise = torch.from_numpy(batch_noise(B, T)).to(dev)
nat = model.set_conditioning(xfp, xf, [T - 37 * i for i in range(B)] if os.environ.get("DC_RAGGED") else [T] * B)
out, _ = nat.ddim_loop(noise, gd.native_coefficients())
out2, _ = nat.ddim_loop(noise, gd.native_coefficients())
torch.cuda.synchronize()
assert nat.status() == 0 and torch.equal(out, out2)
print(f"{os.path.basename(os.environ.get('DC_DDIM_LIB', 'default'))}: bs={B} sha256 {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}"
      f" |x0| {float(out.abs().mean()):.6f}")
