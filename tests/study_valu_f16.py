#!/usr/bin/env python3
"""CPU study (not collected by pytest): what packed-fp16 vector arithmetic in the StylizationBlocks' elementwise part would cost in x0.

Round 5's verdict (item 4): in the PLAIN evaluations of the fp16 mode, compute n-hat, the FiLM affine n-hat G' + H', SiLU's 1 + e and
its final product in packed fp16 (v_pk_fma_f16 / v_pk_add_f16 / v_pk_mul_f16, v_exp_f16 / v_rcp_f16) - the result is rounded to fp16 for
the MFMA anyway - for about -290 of k_layer's 4 537 vector instructions per wave and layer.  Bounded here first: the oracle's
arithmetic with every GEMM class on fp16 operands (what the fp16 kernels do) and, on top, the stylization's elementwise chain rounded to
fp16 after every instruction; the last k evaluations exact (the precise tail).  DDIM-50, one clip, rel-L2 of x0 against the fp32 oracle.
usage: python tests/study_valu_f16.py [frames, default 450]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import study_operand_rounding as R  # noqa: E402  (runs its own tables first)

O, p, f16 = R.O, R.p, R.f16
LOG2E = 1.4426950408889634


class ValuEmu(R.StepEmu):
    """StepEmu (all GEMM classes fp16, last k evaluations exact) + the stylization's vector chain in fp16 in the plain evaluations.
    variant: "chain" = every instruction's result rounded to fp16; "affine" = only n-hat and the affine in fp16, SiLU in fp32."""

    def __init__(self, k, variant, S=50):
        super().__init__(k, S)
        self.variant, self.calls = variant, 0


_orig = O.stylization


def stylization(pp, prefix, h, emb, emu=O.FP32):
    if not isinstance(emu, ValuEmu):
        return _orig(pp, prefix, h, emb, emu)
    i = emu.calls
    emu.calls += 1
    if i // 24 >= emu.S - emu.k:              # 24 StylizationBlocks per evaluation: the tail is exact, as in StepEmu
        return _orig(pp, prefix, h, emb, emu)
    e = emu.linear(F.silu(emb), pp[prefix + ".emb_layers.1.weight"], pp[prefix + ".emb_layers.1.bias"], big=True)
    scale, shift = torch.chunk(e, 2, dim=2)
    g, b = pp[prefix + ".norm.weight"], pp[prefix + ".norm.bias"]
    G = f16(g * (1 + scale))                                   # the FiLM tiles as the GEMM stores them (LayerNorm affine folded, fp16)
    H = f16(LOG2E * (b * (1 + scale) + shift))
    mean = h.mean(-1, keepdim=True)
    var = h.var(-1, unbiased=False, keepdim=True)
    rstd = LOG2E / torch.sqrt(var + 1e-5)
    sh = -mean * rstd
    y16 = f16(h)                                               # the packed y tile
    n = f16(y16 * f16(rstd) + f16(sh))                         # v_pk_fma_f16
    u = f16(G * n + H)                                         # v_pk_fma_f16
    if emu.variant == "chain":
        ex = f16(torch.exp2(-u))                               # v_exp_f16
        d = f16(1 + ex)                                        # v_pk_add_f16
        r = f16(1 / d)                                         # v_rcp_f16
        z = f16(u * r)                                         # v_pk_mul_f16: the MFMA operand
    else:
        z = u / (1 + torch.exp2(-u))
    w = pp[prefix + ".out_layers.2.weight"] / LOG2E            # (ln 2 folded into W_o on the host)
    return f16(z) @ f16(w.t()) + pp[prefix + ".out_layers.2.bias"]


O.stylization = stylization
T = R.T
with torch.no_grad():
    ref = R.ref
    for k in (0, 1, 2):
        base = R.rl2(O.ddim_sample_loop(p, R.noise, R.xfp, R.xf, [T], 50, emu=R.StepEmu(k)), ref)
        ch = R.rl2(O.ddim_sample_loop(p, R.noise, R.xfp, R.xf, [T], 50, emu=ValuEmu(k, "chain")), ref)
        af = R.rl2(O.ddim_sample_loop(p, R.noise, R.xfp, R.xf, [T], 50, emu=ValuEmu(k, "affine")), ref)
        print(f"tail {k}: fp32 vector arithmetic {base:.3e}   n-hat + affine in fp16 {af:.3e}   whole chain in fp16 {ch:.3e}")
