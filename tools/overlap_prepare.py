#!/usr/bin/env python3
"""GPU box: can the NEXT batch's preparation (encode_music of a pinned mel batch + set_conditioning, 5.9 ms) hide beside the CURRENT batch's
DDIM-50 loop (34 ms) on a second stream and a second sampler?  Sequential vs side by side, ms per (prepare + loop) pair.
usage: python tools/overlap_prepare.py"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402
import bench  # noqa: E402
from diffusion_conductor_amd.sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule  # noqa: E402
from diffusion_conductor_amd.synthetic import batch_mel, batch_noise  # noqa: E402

B, T, S = 32, 1800, 50
dev = torch.device("cuda", 0)
coef = GaussianDiffusion(betas=get_named_beta_schedule("linear", S), model_mean_type=ModelMeanType.START_X,
                         model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE).native_coefficients()
models = [bench.build_model("fp16", False, dev) for _ in range(2)]
mel = torch.from_numpy(batch_mel(B, 3 * T)).pin_memory()
noise = torch.from_numpy(batch_noise(B, T)).to(dev)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
nats = []
for m in models:
    xp, x = m.encode_music(mel, dev)
    nats.append(m.set_conditioning(xp, x, [T] * B))
torch.cuda.synchronize()


def prepare(i):
    xp, x = models[i].encode_music(mel, dev)
    return models[i].set_conditioning(xp, x, [T] * B), (xp, x)


def run(side_by_side, n=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = []
    for k in range(n):
        i = k & 1
        if side_by_side:
            with torch.cuda.stream(streams[0]):
                out = nats[i].ddim_loop(noise, coef)[0]          # loop of the current batch on sampler i
            with torch.cuda.stream(streams[1]):
                nats[i ^ 1], kp = prepare(i ^ 1)                  # preparation of the next batch on the other sampler
            streams[0].wait_stream(streams[1])
            streams[1].wait_stream(streams[0])
        else:
            with torch.cuda.stream(streams[0]):
                out = nats[i].ddim_loop(noise, coef)[0]
                nats[i ^ 1], kp = prepare(i ^ 1)
        keep.append((out, kp))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(False, 2); run(True, 2)
for rep in range(3):
    a, b = run(False), run(True)
    print(f"prepare + loop per batch: one stream {a:.2f} ms, two streams {b:.2f} ms ({100 * (b / a - 1):+.1f} %)", flush=True)
