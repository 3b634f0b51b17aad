#!/usr/bin/env python3
"""GPU box: do two independent DDIM-50 loops (two samplers, two batches of 32 clips) finish sooner side by side on two streams than one
after the other?  The layer launches leave 28 of 256 CUs and every kernel boundary idle; the other loop's launches could fill them.
usage: python tools/two_streams.py [bs]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402
import bench  # noqa: E402
from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise  # noqa: E402

B, T, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 1800, 50
dev = torch.device("cuda", 0)
gd = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.START_X,
                       model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
coef = gd.native_coefficients()
nats, noises = [], []
for i in range(2):
    model = bench.build_model("fp16", False, dev)
    xf = torch.from_numpy(batch_music_features(B, T, first=i * B)).to(dev)
    xfp = torch.nn.functional.linear(xf, model.proj.weight, model.proj.bias).contiguous()
    noises.append(torch.from_numpy(batch_noise(B, T, first=i * B)).to(dev))
    nats.append((model, model.set_conditioning(xfp, xf, [T] * B)))
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def run(concurrent, n=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for _ in range(n):
        for i in range(2):
            with torch.cuda.stream(streams[i if concurrent else 0]):
                outs.append(nats[i][1].ddim_loop(noises[i], coef)[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n) * 1e3, outs


run(False, 2); run(True, 2)
for rep in range(3):
    a, oa = run(False)
    b, ob = run(True)
    same = all(torch.equal(x, y) for x, y in zip(oa[-2:], ob[-2:]))
    print(f"bs={B}: one stream {a:.3f} ms per loop, two streams {b:.3f} ms per loop ({100 * (b / a - 1):+.1f} %), identical results: {same}", flush=True)
