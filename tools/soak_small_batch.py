#!/usr/bin/env python3
"""GPU box: the small-batch path (k_layer16 with the in-launch combine exchange) run many times - every loop must end with status 0
(no spurious DC_STATUS_TIMEOUT on an unshared GPU) and repeat bit for bit.  usage: python tools/soak_small_batch.py [loops per batch size]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402
from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
model = bench.build_model("fp16", False, dev)
gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", 50), model_mean_type=ModelMeanType.START_X,
                       model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
coef = gd.native_coefficients()
for B in (1, 2, 5, 8):
    T = 1800
    xf = torch.from_numpy(batch_music_features(B, T)).to(dev)
    xfp = torch.nn.functional.linear(xf, model.proj.weight, model.proj.bias).contiguous()
    noise = torch.from_numpy(batch_noise(B, T)).to(dev)
    nat = model.set_conditioning(xfp, xf, [T] * B)
    ref, _ = nat.ddim_loop(noise, coef)
    torch.cuda.synchronize()
    assert nat.status() == 0
    ref = ref.clone()
    t0 = time.perf_counter()
    bad = 0
    for i in range(N):
        out, _ = nat.ddim_loop(noise, coef)
        st = nat.status()
        if st != 0 or not torch.equal(out, ref):
            bad += 1
            print(f"bs={B} loop {i}: status {st}, equal {bool(torch.equal(out, ref))}")
    print(f"bs={B}: {N} loops, {bad} bad, {1e3 * (time.perf_counter() - t0) / N:.2f} ms per loop incl. the status read")
