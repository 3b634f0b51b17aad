#!/usr/bin/env python3
"""CPU study (not collected by pytest): which GEMM's fp16 operand rounding carries the fp16 mode's x0 error.

The oracle's arithmetic with an `Emu` that rounds the operands of ONE class of GEMM to fp16 (and stores the FiLM tiles as fp16 when the
class is "film") while every other GEMM stays fp32, DDIM-50 on one clip; rel-L2 of x0 against the fp32 oracle.  Then all classes
together (what the fp16 kernels do, up to the order of their sums) and all but one.
usage: python tests/study_operand_rounding.py [frames, default 450]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_conductor_amd.synthetic import batch_music_features, batch_noise, synthetic_state_dict  # noqa: E402
from oracle import ddim_oracle as O  # noqa: E402

torch.set_num_threads(8)
p = {k: torch.as_tensor(v) for k, v in synthetic_state_dict().items()}
name_of = {id(v): k for k, v in p.items()}
T = int(sys.argv[1]) if len(sys.argv) > 1 else 450
xf = torch.from_numpy(batch_music_features(1, T))
xfp = F.linear(xf, p["proj.weight"], p["proj.bias"])
noise = torch.from_numpy(batch_noise(1, T))
f16 = lambda x: x.half().float()


def site_of_linear(w):
    n = name_of.get(id(w), "?")
    if "emb_layers" in n:
        return "film"
    if "out_layers" in n:
        return "styl_out"
    for s in ("sa_block.query", "sa_block.key", "sa_block.value", "ca_block.query", "ffn.linear1", "ffn.linear2"):
        if s in n:
            return s.replace("_block", "")
    return "other"            # linear / time_embed / joint_embed / out / ca key, value: split or fp32 in the kernels


class SiteEmu(O.Emu):
    def __init__(self, sites):
        super().__init__("fp32", film_store_f16="film" in sites)
        self.sites = set(sites)
        self.block = "sa"

    def linear(self, x, w, b, big=False):
        s = site_of_linear(w)
        n = name_of.get(id(w), "")
        if "sa_block" in n:
            self.block = "sa"
        elif "ca_block" in n:
            self.block = "ca"
        if s in self.sites:
            y = f16(x) @ f16(w.t())
            return y + b if b is not None else y
        return F.linear(x, w, b)

    def einsum(self, eq, a, b):
        s = ("%s.kv" if eq == "bnhd,bnhl->bhdl" else "%s.attend") % self.block
        if s in self.sites:            # (ca.kv is the pre-pass: split in the kernels - listed for completeness)
            return torch.einsum(eq, f16(a), f16(b))
        return torch.einsum(eq, a, b)


ALL = ["film", "sa.query", "sa.key", "sa.value", "sa.kv", "sa.attend", "ca.query", "ca.attend", "ffn.linear1", "ffn.linear2", "styl_out"]
rl2 = lambda a, b: float((a - b).norm() / b.norm())
with torch.no_grad():
    ref = O.ddim_sample_loop(p, noise, xfp, xf, [T], 50)
    run = lambda sites: rl2(O.ddim_sample_loop(p, noise, xfp, xf, [T], 50, emu=SiteEmu(sites)), ref)
    print(f"all classes fp16 (the fp16 kernels' roundings): {run(ALL):.3e}")
    for s in ALL:
        print(f"  only {s:12s}: {run([s]):.3e}    all but it: {run([t for t in ALL if t != s]):.3e}")


class HalfSiteEmu(SiteEmu):
    """one class with only its weights ("w") or only its activations ("a") rounded"""

    def __init__(self, site, which):
        super().__init__([site])
        self.which = which

    def linear(self, x, w, b, big=False):
        if site_of_linear(w) in self.sites:
            y = (f16(x) if self.which == "a" else x) @ (f16(w.t()) if self.which == "w" else w.t())
            return y + b if b is not None else y
        return super().linear(x, w, b, big)


with torch.no_grad():
    for s in ("styl_out", "ffn.linear1", "ffn.linear2", "sa.value", "film"):
        ea = rl2(O.ddim_sample_loop(p, noise, xfp, xf, [T], 50, emu=HalfSiteEmu(s, "a")), ref)
        ew = rl2(O.ddim_sample_loop(p, noise, xfp, xf, [T], 50, emu=HalfSiteEmu(s, "w")), ref)
        print(f"  {s:12s}: activations only {ea:.3e}   weights only {ew:.3e}")


class StepEmu(SiteEmu):
    """every class on fp16 operands except in the LAST `k` model evaluations of the loop, which run exact: what the precise tail
    (include/dc_ddim.h, dc_sampler_set_precise_tail) was derived from - 5.0e-4 / 2.4e-4 / 1.6e-4 / 1.1e-4 / 6.9e-5 for k = 0, 1, 2, 4, 8"""

    def __init__(self, k, S=50):
        super().__init__(ALL)
        self.k, self.S, self.n = k, S, {}

    def linear(self, x, w, b, big=False):
        s = site_of_linear(w)
        if s in self.sites:
            c = self.n.get(id(w), 0)
            self.n[id(w)] = c + 1
            if c >= self.S - self.k:                      # (every weight matrix is used once per evaluation)
                n = name_of.get(id(w), "")
                self.block = "sa" if "sa_block" in n else "ca" if "ca_block" in n else self.block
                return F.linear(x, w, b)
        return super().linear(x, w, b, big)

    def einsum(self, eq, a, b):
        i = self.n.get((eq, self.block), 0)
        self.n[(eq, self.block)] = i + 1
        if i // 8 >= self.S - self.k:                     # (8 layers per evaluation)
            return torch.einsum(eq, a, b)
        return super().einsum(eq, a, b)


with torch.no_grad():
    for k in (0, 1, 2, 4, 8):
        e = StepEmu(k)
        print(f"  last {k} model evaluations exact, the others on fp16 operands: {rl2(O.ddim_sample_loop(p, noise, xfp, xf, [T], 50, emu=e), ref):.3e}")
