"""GPU box: an EPSILON model at eta = 0 (B=2, T=256, DDIM-25) against the oracle with 0 / 1 / 4 / all evaluations on split operands, and the
split precisions beside it: fp16 1.52e-3 / 1.60e-3 / 1.58e-3 / 2.1e-4, mixed 2.1e-4, bf16x3 9.6e-5; full attention (no split kernels) 1.4e-3.  usage: python tools/epsilon_tail_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from helpers import O, batch_noise, make_model, oracle_params, rel_l2, xf_pair
from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule
torch.set_num_threads(16)
B, T, S = 2, 256, 25
xfp, xfo = xf_pair(B, T, first=3)
noise = torch.from_numpy(batch_noise(B, T, first=3))
length = [256, 200]
with torch.no_grad():
    ref = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S, clip_denoised=True, eps_model=True)
gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.EPSILON, model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
for prec in ("fp16", "mixed", "bf16x3"):
    m = make_model(prec)
    for tail in ("0", "1", "4", "25"):
        os.environ["DC_PRECISE_TAIL"] = tail
        out = gd.ddim_sample_loop(m, (B, T, 26), noise=noise.cuda(), clip_denoised=True, progress=False,
                                  model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
        torch.cuda.synchronize()
        print(prec, "tail", tail, f"{rel_l2(out, ref):.3e}", flush=True)
        if prec != "fp16": break
# full attention (`no_eff`) has no split kernels: plain operands whatever the tail
with torch.no_grad():
    ref_ne = O.ddim_sample_loop(oracle_params(), noise, xfp, xfo, length, S, clip_denoised=True, eps_model=True, no_eff=True)
m = make_model("fp16", no_eff=True)
out = gd.ddim_sample_loop(m, (B, T, 26), noise=noise.cuda(), clip_denoised=True, progress=False,
                          model_kwargs={"xf_proj": xfp.cuda(), "xf_out": xfo.cuda(), "length": torch.LongTensor(length)})
torch.cuda.synchronize()
print("fp16 no_eff", f"{rel_l2(out, ref_ne):.3e}", flush=True)
