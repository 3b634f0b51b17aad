#!/usr/bin/env python3
"""GPU box, with a -DDC_DIAG_FULL_MOVES build (tools/ab.sh build M -DDC_DIAG_FULL_MOVES; DC_DDIM_LIB=.../libdc_ddim_M.alt): how often the
no_eff key loop's lazily moved reference point moves, on the G10 cases (B=1, T=1800 seed-0 checkpoint; B=2, T=900 ragged, stress
checkpoint), with the rel-L2 of x0 against the reference's golden."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import DenoiserConfig, O, batch_noise, golden, make_diffusion, make_model, rel_l2, xf_pair
from diffusion_conductor_amd import MotionTransformer
from diffusion_conductor_amd.synthetic import batch_music_features, stress_state_dict
g = golden("g10_no_eff_long.npz")
gd = make_diffusion(50)
def run(m, noise, xfp, xfo, length, ref, tag):
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), length)
    nat.debug_read("full_moves", np.uint64, 2)                       # reset
    out, _ = nat.ddim_loop(noise.cuda(), gd.native_coefficients())
    torch.cuda.synchronize()
    v, mv = [int(x) for x in nat.debug_read("full_moves", np.uint64, 2)]
    print(f"{tag}: rel-L2 vs the reference {rel_l2(out, ref):.3e};  (key tile > 0, head, wave) visits {v}, reference point moved in {mv} = {100.0 * mv / max(v, 1):.3f} %")
m = make_model("fp16", no_eff=True)
xfp, xfo = xf_pair(1, 1800, first=50)
run(m, torch.from_numpy(batch_noise(1, 1800, first=50)), xfp, xfo, [1800], g["t1800_x0"], "seed-0 checkpoint, B=1, T=1800, DDIM-50")
sd = stress_state_dict(DenoiserConfig(), seed=0)
m2 = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True, precision="fp16", no_eff=True)
m2.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
m2 = m2.to("cuda").eval()
q = O.to_torch_params(sd, torch.float32)
xf2 = torch.from_numpy(batch_music_features(2, 900, first=52))
xfp2 = torch.nn.functional.linear(xf2, q["proj.weight"], q["proj.bias"])
run(m2, torch.from_numpy(batch_noise(2, 900, first=52)), xfp2, xf2, [int(v) for v in g["stress_t900_length"]], g["stress_t900_x0"],
    "stress checkpoint, B=2, T=900 ragged, DDIM-50")
# sharpened self-attention (tests/test_gpu_parity.py::_peaky_state_dict): the case that does move the reference point
from test_gpu_parity import _peaky_state_dict
sd3 = _peaky_state_dict()
m3 = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, device="cuda", no_clip=True, precision="fp16", no_eff=True)
m3.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd3.items()}, strict=True)
m3 = m3.to("cuda").eval()
q3 = O.to_torch_params(sd3, torch.float32)
xfp3, xfo3 = xf_pair(2, 320, first=82)
nz3 = torch.from_numpy(batch_noise(2, 320, first=82))
gd25 = make_diffusion(25)
with torch.no_grad():
    ref3 = O.ddim_sample_loop(q3, nz3, xfp3, xfo3, [320, 211], 25, no_eff=True)
nat = m3.set_conditioning(xfp3.cuda(), xfo3.cuda(), [320, 211])
nat.debug_read("full_moves", np.uint64, 2)
out3, _ = nat.ddim_loop(nz3.cuda(), gd25.native_coefficients())
torch.cuda.synchronize()
v, mv = [int(x) for x in nat.debug_read("full_moves", np.uint64, 2)]
print(f"sharpened self-attention (query / key x 5), B=2, T=320 ragged, DDIM-25: rel-L2 vs the oracle {rel_l2(out3, ref3):.3e};  visits {v}, reference point moved in {mv} = {100.0 * mv / max(v, 1):.3f} %")
