"""Stage-by-stage check of the 16-token layer kernel (DC_LAYER16=1) against the oracle's taps and the 32-token kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from helpers import O, oracle_params, xf_pair, batch_noise, make_model, rel_l2

os.environ["DC_NO_NARROW"] = "1"
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
m = make_model(prec)
p = oracle_params()
for B, T in ((2, 900), (3, 700), (5, 1800)):
    xfp, xfo = xf_pair(B, T)
    x = torch.from_numpy(batch_noise(B, T))
    t = torch.tensor([(7 * b + 3) % 50 for b in range(B)])
    length = [T if b % 2 == 0 else max(1, T - 17 - b) for b in range(B)]
    taps = {}
    with torch.no_grad():
        ref = O.denoiser_forward(p, x, t, length, xfp, xfo, taps=taps)
    nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), length)
    xd = x.cuda()
    M = B * T
    for mode in ("32", "16"):
        if mode == "16":
            os.environ["DC_LAYER16"] = "1"
        else:
            os.environ.pop("DC_LAYER16", None)
        line = []
        for i in range(8):
            for stage, tap in ((1, f"sa{i}"), (2, f"ca{i}"), (3, f"ffn{i}")):
                nat.debug_denoise(xd, t.numpy(), i + 1, stage)
                torch.cuda.synchronize()
                h = nat.read_h()[:M].reshape(B, T, 128)
                e = rel_l2(h, taps[tap])
                line.append(f"{tap}:{e:.1e}")
        out = nat.denoise(xd, t.numpy())
        torch.cuda.synchronize()
        print(f"B={B} T={T} [{prec}] layer{mode}: forward {rel_l2(out, ref):.3e} finite={bool(torch.isfinite(out).all())}\n   " + " ".join(line), flush=True)
os.environ.pop("DC_LAYER16", None)
