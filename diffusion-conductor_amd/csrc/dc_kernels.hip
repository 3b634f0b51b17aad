// dc_kernels.hip - hand-written gfx950 (CDNA4) kernels of the DDIM denoising step.
//
// Design: every activation lives in "FT" form - a stack of 32x32 fp32
// tiles in the v_mfma_f32_32x32x16_bf16 accumulator layout with the TOKEN on the lane
// and the FEATURES in registers.  Consequences:
//   * per-token reductions over features (LayerNorm, softmax over head_dim, FiLM) are
//     in-lane register reductions plus ONE exchange with lane^32;
//   * an accumulator tile converts in registers (v_cvt_pk_bf16_f32) into the B operand
//     of the next MFMA (W * X) or the A operand (X^T * W), so the whole
//     LN -> QKV -> softmax -> attention -> FiLM -> out-proj -> FFN chain of one token
//     group never leaves the register file; weights are pre-packed host-side in the
//     matching fragment order so each operand is one coalesced 16-B-per-lane load;
//   * the only cross-token dependency per layer (softmax over the sequence of K and
//     K^T V of the linear attention) is a two-phase reduction: per-group partial
//     records (k_embed_front / k_layer) -> k_attn_combine.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "dc_common.h"

#define DEV __device__ __forceinline__

namespace dc {

DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
DEV f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// value of the partner lane (lane ^ 32) combined with this lane's: one v_permlane32_swap, no LDS
DEV float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
DEV float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// row (feature in FT, token in TF) held by register r of lane-half hh
DEV int tile_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

DEV f32x16 splat(float v) {
    f32x16 x;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = v;
    return x;
}

// One 32-row tile as MFMA operand fragments for its two 16-deep k-steps (hi [+ lo]).
template <bool SPLIT>
struct XFrag {
    bf16x8 hi[2];
    bf16x8 lo[SPLIT ? 2 : 1];
};

template <bool SPLIT>
DEV void make_frag(const f32x16& x, XFrag<SPLIT>& f) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[8 * s + j];
            const __bf16 h = (__bf16)v;
            f.hi[s][j] = h;
            if constexpr (SPLIT) f.lo[s][j] = (__bf16)(v - (float)h);
        }
}

template <bool SPLIT>
DEV void mask_frag(XFrag<SPLIT>& f, bool keep) {
    if (!keep) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f.hi[s][j] = (__bf16)0.f;
                if constexpr (SPLIT) f.lo[s][j] = (__bf16)0.f;
            }
    }
}

// acc[ot] (rows = output features, cols = tokens) += W[ot][kt] * X[kt]; weights are the A operand.
template <int OT, int KT, bool SPLIT>
DEV void gemm_wa(f32x16 (&acc)[OT], const bf16x8* __restrict__ w, const XFrag<SPLIT> (&x)[KT], int lane) {
    constexpr int NF = OT * KT * 2;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int fi = (ot * KT + kt) * 2 + s;
                const bf16x8 a = w[fi * 64 + lane];
                acc[ot] = mfma(a, x[kt].hi[s], acc[ot]);
                if constexpr (SPLIT) {
                    acc[ot] = mfma(a, x[kt].lo[s], acc[ot]);
                    const bf16x8 al = w[(NF + fi) * 64 + lane];
                    acc[ot] = mfma(al, x[kt].hi[s], acc[ot]);
                }
            }
}

// acc[oc] (rows = tokens, cols = output features) += X^T[kt] * W[oc][kt]; weights are the B operand.
template <int OC, int KT, bool SPLIT>
DEV void gemm_wb(f32x16 (&acc)[OC], const bf16x8* __restrict__ w, const XFrag<SPLIT> (&x)[KT], int lane) {
    constexpr int NF = OC * KT * 2;
#pragma unroll
    for (int oc = 0; oc < OC; ++oc)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int fi = (oc * KT + kt) * 2 + s;
                const bf16x8 b = w[fi * 64 + lane];
                acc[oc] = mfma(x[kt].hi[s], b, acc[oc]);
                if constexpr (SPLIT) {
                    acc[oc] = mfma(x[kt].lo[s], b, acc[oc]);
                    const bf16x8 bl = w[(NF + fi) * 64 + lane];
                    acc[oc] = mfma(x[kt].hi[s], bl, acc[oc]);
                }
            }
}

// per-feature vector stored as [tile][lane-half][16] so a lane reads its 16 values with one 64-B load
DEV f32x16 ld_ft(const float* __restrict__ p, int tile, int hh) {
    return *reinterpret_cast<const f32x16*>(p + (tile * 2 + hh) * 16);
}

// nn.LayerNorm(128) over the feature axis of an FT activation (transformer.py:79,104,147)
template <int NT>
DEV void layernorm_ft(const f32x16 (&x)[NT], f32x16 (&y)[NT], const float* __restrict__ g,
                      const float* __restrict__ b, int hh) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x[t][r];
    const float mean = xhalf_sum(s) * (1.f / (32 * NT));
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = x[t][r] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(xhalf_sum(q) * (1.f / (32 * NT)) + 1e-5f);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x16 gg = ld_ft(g, t, hh), bb = ld_ft(b, t, hh);
#pragma unroll
        for (int r = 0; r < 16; ++r) y[t][r] = (x[t][r] - mean) * rstd * gg[r] + bb[r];
    }
}

// F.softmax(query.view(B,T,H,-1), dim=-1) (transformer.py:109,150): a head = 16 features
// = registers 8p..8p+7 of this lane and of lane^32.
DEV void softmax_heads_ft(f32x16 (&q)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float m = q[t][8 * p];
#pragma unroll
            for (int j = 1; j < 8; ++j) m = fmaxf(m, q[t][8 * p + j]);
            m = xhalf_max(m);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float e = __expf(q[t][8 * p + j] - m);
                q[t][8 * p + j] = e;
                s += e;
            }
            const float inv = fast_rcp(xhalf_sum(s));
#pragma unroll
            for (int j = 0; j < 8; ++j) q[t][8 * p + j] *= inv;
        }
}

DEV float silu(float z) { return z * fast_rcp(1.f + __expf(-z)); }
DEV float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// y = softmax(Q) . A per head, i.e. FT tile oc of y = A_frag[oc]^T-as-A-operand * Q tile oc.
// afrag: [4 oc][2 s][64 lanes] hi frags, followed by the same count of lo frags.
template <bool SPLIT>
DEV void attn_apply(f32x16 (&y)[4], const bf16x8* __restrict__ afrag, const XFrag<SPLIT> (&q)[4], int lane) {
#pragma unroll
    for (int oc = 0; oc < 4; ++oc)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 a = afrag[(oc * 2 + s) * 64 + lane];
            y[oc] = mfma(a, q[oc].hi[s], y[oc]);
            if constexpr (SPLIT) {
                y[oc] = mfma(a, q[oc].lo[s], y[oc]);
                const bf16x8 al = afrag[(8 + oc * 2 + s) * 64 + lane];
                y[oc] = mfma(al, q[oc].hi[s], y[oc]);
            }
        }
}

// token group geometry shared by the per-group kernels
struct GroupCtx {
    int g, lane, c, hh;
    int tok;          // this lane's token (flat)
    int b0, b1;       // first / last clip touched by the group
    int boundary;     // first token of clip b1 (== (b0+1)*T when straddling)
    bool straddle;
    bool lane_in_b0;
};

DEV GroupCtx make_ctx(int g, int lane, int M, int T) {
    GroupCtx x;
    x.g = g;
    x.lane = lane;
    x.c = lane & 31;
    x.hh = lane >> 5;
    x.tok = 32 * g + x.c;
    const int first = 32 * g;
    const int last = min(first + 31, M - 1);
    x.b0 = first / T;
    x.b1 = last / T;
    x.straddle = x.b1 != x.b0;
    x.boundary = (x.b0 + 1) * T;
    x.lane_in_b0 = x.tok < x.boundary;
    return x;
}

// StylizationBlock.forward (transformer.py:68-81) given the precomputed FiLM tiles
// E (fp16 FT tiles: 4 scale tiles then 4 shift tiles for this block and group):
//   o = W_o * SiLU( LN(y) * (1 + scale) + shift ) + b_o
template <bool SPLIT>
DEV void stylization(f32x16 (&o)[4], const f32x16 (&y)[4], const f16x16* __restrict__ E,
                     const DcStyl& w, int lane, int hh) {
    f32x16 n[4];
    layernorm_ft<4>(y, n, w.ln_g, w.ln_b, hh);
    XFrag<SPLIT> a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f16x16 sc = E[t * 64 + lane];
        const f16x16 sh = E[(4 + t) * 64 + lane];
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = silu(n[t][r] * (1.f + (float)sc[r]) + (float)sh[r]);
        make_frag<SPLIT>(z, a[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = ld_ft(w.bo, t, hh);
    gemm_wa<4, 4, SPLIT>(o, w.wo, a, lane);
}

// "Front half" of LinearTemporalSelfAttention (transformer.py:104-117) for one group:
// n = LN(h); K = Wk n + bk; V = Wv n + bv in TF form; then the group's partial record
// of softmax_T(K + mask) and K^T V, one record per clip the group touches.
template <bool SPLIT>
DEV void sa_front(const f32x16 (&h)[4], const DcLayer& L, const GroupCtx& cx, int M, int T,
                  const int* __restrict__ length, float* __restrict__ rec /* this group's 2 records */) {
    f32x16 n[4];
    layernorm_ft<4>(h, n, L.sa_ln_g, L.sa_ln_b, cx.hh);
    XFrag<SPLIT> nf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) make_frag<SPLIT>(n[t], nf[t]);
    f32x16 K[4], V[4];
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
        K[oc] = splat(L.sa_bk[32 * oc + cx.c]);
        V[oc] = splat(L.sa_bv[32 * oc + cx.c]);
    }
    gemm_wb<4, 4, SPLIT>(K, L.sa_wk, nf, cx.lane);
    gemm_wb<4, 4, SPLIT>(V, L.sa_wv, nf, cx.lane);

    const int nslot = cx.straddle ? 2 : 1;
    for (int slot = 0; slot < nslot; ++slot) {
        const int bs = slot == 0 ? cx.b0 : cx.b1;
        const int len = length[bs];
        const int base = bs * T;
        unsigned valid = 0;   // bit r: token of register r belongs to clip bs and is unmasked
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tk = 32 * cx.g + tile_row(r, cx.hh);
            const int nn = tk - base;
            if (tk < M && nn >= 0 && nn < T && nn < len) valid |= 1u << r;
        }
        float* R = rec + (size_t)slot * DC_REC_FLOATS;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (valid & (1u << r)) m = fmaxf(m, K[oc][r]);
            m = xhalf_max(m);
            if (m == -INFINITY) m = 0.f;
            f32x16 Ee, Vm;
            float ssum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = valid & (1u << r);
                const float e = ok ? __expf(K[oc][r] - m) : 0.f;
                Ee[r] = e;
                ssum += e;
                Vm[r] = ok ? V[oc][r] : 0.f;
            }
            ssum = xhalf_sum(ssum);
            XFrag<SPLIT> ef, vf;
            make_frag<SPLIT>(Ee, ef);
            make_frag<SPLIT>(Vm, vf);
            f32x16 P = splat(0.f);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                P = mfma(ef.hi[s], vf.hi[s], P);
                if constexpr (SPLIT) {
                    P = mfma(ef.lo[s], vf.hi[s], P);
                    P = mfma(ef.hi[s], vf.lo[s], P);
                }
            }
            if (cx.hh == 0) {
                R[32 * oc + cx.c] = m;
                R[128 + 32 * oc + cx.c] = ssum;
            }
            reinterpret_cast<f32x16*>(R + 256)[oc * 64 + cx.lane] = P;
        }
    }
}

DEV void load_h(f32x16 (&h)[4], const float* __restrict__ hbuf, int g, int lane) {
    const f32x16* p = reinterpret_cast<const f32x16*>(hbuf) + (size_t)g * 256 + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) h[t] = p[t * 64];
}
DEV void store_h(const f32x16 (&h)[4], float* __restrict__ hbuf, int g, int lane) {
    f32x16* p = reinterpret_cast<f32x16*>(hbuf) + (size_t)g * 256 + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) p[t * 64] = h[t];
}

}  // namespace dc
using namespace dc;

// ------------------------------------------------------------------------------------
// per-step bookkeeping: iteration counter -> timestep per clip, DDIM scalars, snapshot slot
// ------------------------------------------------------------------------------------
__global__ void k_begin_step(int* __restrict__ iter, const int* __restrict__ t_of_iter,
                             const float* __restrict__ coef_of_t, const int* __restrict__ snap_of_iter,
                             int* __restrict__ t_clip, float* __restrict__ coef_cur, int* __restrict__ snap_cur, int B) {
    const int it = *iter;
    const int t = t_of_iter[it];
    for (int b = threadIdx.x; b < B; b += blockDim.x) t_clip[b] = t;
    if (threadIdx.x < 4) coef_cur[threadIdx.x] = coef_of_t[t * 4 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        *snap_cur = snap_of_iter ? snap_of_iter[it] : -1;
        *iter = it + 1;
    }
}

// ------------------------------------------------------------------------------------
// timestep_embedding + time_embed MLP table (transformer.py:8-25, 410-414): one block per t
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_temb_table(const float* __restrict__ freqs /*[64]*/,
                                                    const float* __restrict__ w0t /*[128][512]*/, const float* __restrict__ b0,
                                                    const float* __restrict__ w2t /*[512][512]*/, const float* __restrict__ b2,
                                                    float* __restrict__ temb /*[nt][512]*/) {
    __shared__ float te[128];
    __shared__ float hid[512];
    const int t = blockIdx.x, k = threadIdx.x;
    if (k < 64) {
        const float a = (float)t * freqs[k];
        te[k] = cosf(a);
        te[64 + k] = sinf(a);
    }
    __syncthreads();
    float acc = b0[k];
    for (int i = 0; i < 128; ++i) acc = fmaf(te[i], w0t[i * 512 + k], acc);
    hid[k] = acc / (1.f + expf(-acc));
    __syncthreads();
    float o = b2[k];
    for (int i = 0; i < 512; ++i) o = fmaf(hid[i], w2t[i * 512 + k], o);
    temb[(size_t)t * 512 + k] = o;
}

// ------------------------------------------------------------------------------------
// conditioning (step-invariant, once per batch)
// ------------------------------------------------------------------------------------
// y[tok][k] = b[k] + sum_i xf[tok][i] * Wt[i][k]   (`self.linear`, transformer.py:479-480)
__global__ void k_cond_linear(const float* __restrict__ xf /*[M][64]*/, const float* __restrict__ wt,
                              const float* __restrict__ b, float* __restrict__ y /*[Mpad][512]*/, int M, int Mpad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Mpad * 512) return;
    const int k = idx & 511;
    const size_t tok = idx >> 9;
    if (tok >= (size_t)M) {
        y[idx] = 0.f;
        return;
    }
    float acc = b[k];
    const float* x = xf + tok * 64;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) acc = fmaf(x[i], wt[i * 512 + k], acc);
    y[idx] = acc;
}

// per-row mean / rstd over 512 (text_norm without its affine, transformer.py:149); one wave per token
__global__ void k_row_stats512(const float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int Mpad) {
    const int tok = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= Mpad) return;
    const float* r = y + (size_t)tok * 512;
    float v[8], s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = r[lane + 64 * i];
        s += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s * (1.f / 512.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += (v[i] - mu) * (v[i] - mu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) {
        mean[tok] = mu;
        rstd[tok] = rsqrtf(q * (1.f / 512.f) + 1e-5f);
    }
}

// row-major [Mpad][512] fp32 -> fragment-major: element (g, ks, lane, j) = y[32g + (lane&31)][16ks + 8(lane>>5) + j]
// MODE 0: fp32 image (the xf_proj' term of emb);  MODE 1: normalised, bf16 hi (+lo) operand image.
template <int MODE>
__global__ void k_cond_pack(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ rstd,
                            float* __restrict__ out_f32, bf16x8* __restrict__ out_hi, bf16x8* __restrict__ out_lo, int G) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (g*32+ks)*64 + lane
    if (idx >= (size_t)G * 32 * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) & 31;
    const size_t g = idx >> 11;
    const size_t tok = g * 32 + (lane & 31);
    const float* src = y + tok * 512 + 16 * ks + 8 * (lane >> 5);
    f32x8 v = *reinterpret_cast<const f32x8*>(src);
    if constexpr (MODE == 0) {
        reinterpret_cast<f32x8*>(out_f32)[idx] = v;
    } else {
        const float mu = mean[tok], rs = rstd[tok];
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float n = (v[j] - mu) * rs;
            hi[j] = (__bf16)n;
            lo[j] = (__bf16)(n - (float)hi[j]);
        }
        out_hi[idx] = hi;
        if (out_lo) out_lo[idx] = lo;
    }
}

// Cross-attention K/V for every layer + their partial records (transformer.py:149-155):
// K = Wk' nhat + bk', V = Wv' nhat + bv' with text_norm's affine folded into Wk'/Wv'.
// grid (ceil(G/4), L); one wave per (group, layer).  CA has no mask: every real token is valid.
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_cond_ca_partials(const DcModel* __restrict__ dm, const bf16x8* __restrict__ nh_hi,
                                                          const bf16x8* __restrict__ nh_lo, float* __restrict__ recs,
                                                          int M, int T, int G) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.y;
    const DcLayer& L = dm->layer[l];
    const GroupCtx cx = make_ctx(g, lane, M, T);
    f32x16 K[4], V[4];
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
        K[oc] = splat(L.ca_bk[32 * oc + cx.c]);
        V[oc] = splat(L.ca_bv[32 * oc + cx.c]);
    }
    constexpr int NF = 4 * DC_KS_E;
    for (int ks = 0; ks < DC_KS_E; ++ks) {
        const bf16x8 a = nh_hi[((size_t)g * DC_KS_E + ks) * 64 + lane];
        bf16x8 al;
        if constexpr (SPLIT) al = nh_lo[((size_t)g * DC_KS_E + ks) * 64 + lane];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            const bf16x8 bk = L.ca_wk[(oc * DC_KS_E + ks) * 64 + lane];
            const bf16x8 bv = L.ca_wv[(oc * DC_KS_E + ks) * 64 + lane];
            K[oc] = mfma(a, bk, K[oc]);
            V[oc] = mfma(a, bv, V[oc]);
            if constexpr (SPLIT) {
                K[oc] = mfma(al, bk, K[oc]);
                V[oc] = mfma(al, bv, V[oc]);
                K[oc] = mfma(a, L.ca_wk[((NF + oc * DC_KS_E + ks)) * 64 + lane], K[oc]);
                V[oc] = mfma(a, L.ca_wv[((NF + oc * DC_KS_E + ks)) * 64 + lane], V[oc]);
            }
        }
    }
    float* rec = recs + ((size_t)l * G + g) * 2 * DC_REC_FLOATS;
    const int nslot = cx.straddle ? 2 : 1;
    for (int slot = 0; slot < nslot; ++slot) {
        const int bs = slot == 0 ? cx.b0 : cx.b1;
        const int base = bs * T;
        unsigned valid = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tk = 32 * g + tile_row(r, cx.hh);
            const int nn = tk - base;
            if (tk < M && nn >= 0 && nn < T) valid |= 1u << r;
        }
        float* R = rec + (size_t)slot * DC_REC_FLOATS;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (valid & (1u << r)) m = fmaxf(m, K[oc][r]);
            m = xhalf_max(m);
            if (m == -INFINITY) m = 0.f;
            f32x16 Ee, Vm;
            float ssum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = valid & (1u << r);
                const float e = ok ? __expf(K[oc][r] - m) : 0.f;
                Ee[r] = e;
                ssum += e;
                Vm[r] = ok ? V[oc][r] : 0.f;
            }
            ssum = xhalf_sum(ssum);
            // cross-attention K^T V always in split precision: it is step-invariant (one-time cost)
            XFrag<true> ef, vf;
            make_frag<true>(Ee, ef);
            make_frag<true>(Vm, vf);
            f32x16 P = splat(0.f);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                P = mfma(ef.hi[s], vf.hi[s], P);
                P = mfma(ef.lo[s], vf.hi[s], P);
                P = mfma(ef.hi[s], vf.lo[s], P);
            }
            if (cx.hh == 0) {
                R[32 * oc + cx.c] = m;
                R[128 + 32 * oc + cx.c] = ssum;
            }
            reinterpret_cast<f32x16*>(R + 256)[oc * 64 + lane] = P;
        }
    }
}

// ------------------------------------------------------------------------------------
// combine partial records of one (set, clip, 32-feature tile) into attention operand frags
//   A[d][l] = sum_g w_g[d] P_g[d][l] / sum_g w_g[d] ssum_g[d],   w_g = exp(m_g - max_g m_g)
// recs: [nset][G][2][DC_REC_FLOATS]; afrag out: [nset][B][16 frags (8 hi, 8 lo)][64] bf16x8.
// grid (B, 4, nset), 256 threads: thread (lane, rq) owns registers 4rq..4rq+3 of the tile.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ recs, bf16x8* __restrict__ afrag,
                                                      int T, int G, int B) {
    extern __shared__ float sm[];   // w[ng][32], z[32], red[8][32]
    const int b = blockIdx.x, oc = blockIdx.y, set = blockIdx.z;
    const int g_lo = (b * T) / 32, g_hi = ((b + 1) * T - 1) / 32;
    const int ng = g_hi - g_lo + 1;
    const float* base = recs + (size_t)set * G * 2 * DC_REC_FLOATS;
    float* w = sm;
    float* z = sm + ng * 32;
    float* red = z + 32;
    const int tid = threadIdx.x;
    auto rec_of = [&](int gi) -> const float* {
        const int g = g_lo + gi;
        const int slot = ((32 * g) / T == b) ? 0 : 1;
        return base + ((size_t)g * 2 + slot) * DC_REC_FLOATS;
    };
    // phase 1: column max over the clip's groups; 8 groups in flight per feature
    const int f = tid & 31, part = tid >> 5;
    {
        float mloc = -INFINITY;
        for (int gi = part; gi < ng; gi += 8) {
            const float* R = rec_of(gi);
            const float ss = R[128 + 32 * oc + f], mm = R[32 * oc + f];
            if (ss > 0.f) mloc = fmaxf(mloc, mm);
        }
        red[part * 32 + f] = mloc;
    }
    __syncthreads();
    float mstar = red[f];
#pragma unroll
    for (int k = 1; k < 8; ++k) mstar = fmaxf(mstar, red[k * 32 + f]);
    __syncthreads();
    // phase 2: weights w_g = exp(m_g - m*) and the normaliser
    {
        float zloc = 0.f;
        for (int gi = part; gi < ng; gi += 8) {
            const float* R = rec_of(gi);
            const float ss = R[128 + 32 * oc + f];
            const float ww = ss > 0.f ? __expf(R[32 * oc + f] - mstar) : 0.f;
            w[gi * 32 + f] = ww;
            zloc += ww * ss;
        }
        red[part * 32 + f] = zloc;
    }
    __syncthreads();
    if (tid < 32) {
        float zz = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) zz += red[k * 32 + tid];   // fixed order: deterministic
        z[tid] = zz;
    }
    __syncthreads();
    // phase 3: weighted sum of the partial K^T V tiles, groups in order (deterministic)
    const int lane = tid & 63, rq = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int row0 = 8 * rq + 4 * hh;   // tile_row(4rq + i, hh) = i + 8rq + 4hh
    int gi = 0;
    for (; gi + 4 <= ng; gi += 4) {
        f32x4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = reinterpret_cast<const f32x4*>(rec_of(gi + u) + 256)[(oc * 64 + lane) * 4 + rq];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[(gi + u) * 32 + row0 + i], p[u][i], acc[i]);
    }
    for (; gi < ng; ++gi) {
        const f32x4 p = reinterpret_cast<const f32x4*>(rec_of(gi) + 256)[(oc * 64 + lane) * 4 + rq];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[gi * 32 + row0 + i], p[i], acc[i]);
    }
    const bool keep = (rq >> 1) == (c >> 4);   // same head on both sides
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float zz = z[row0 + i];
        const float a = (keep && zz > 0.f) ? acc[i] / zz : 0.f;
        hi[i] = (__bf16)a;
        lo[i] = (__bf16)(a - (float)hi[i]);
    }
    // register r = 4rq+i of the tile  ->  k-step s = r>>3, element j = r&7 of the A-operand frag
    bf16x8* out = afrag + ((size_t)set * B + b) * 16 * 64;
    const int s = rq >> 1, j0 = (rq & 1) * 4;
    reinterpret_cast<bf16x4*>(out + ((oc * 2 + s) * 64 + lane))[j0 >> 2] = hi;
    reinterpret_cast<bf16x4*>(out + ((8 + oc * 2 + s) * 64 + lane))[j0 >> 2] = lo;
}

// ------------------------------------------------------------------------------------
// per step: S = SiLU(time_embed[t] + xf_proj')  (transformer.py:482 + StylizationBlock's nn.SiLU, :57-58)
// as the bf16 B-operand image of the FiLM GEMM, [G][32 ks][64 lanes][8]
// ------------------------------------------------------------------------------------
// FMODE: 0 = bf16, 1 = split bf16 (hi + lo images), 2 = f16
template <int FMODE>
__global__ void k_silu_emb(const float* __restrict__ pp /*frag-major fp32*/, const float* __restrict__ temb,
                           const int* __restrict__ t_clip, bf16x8* __restrict__ s_hi, bf16x8* __restrict__ s_lo,
                           int G, int T, int B) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)G * 32 * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) & 31;
    const int g = (int)(idx >> 11);
    const int tok = g * 32 + (lane & 31);
    const int b = min(tok / T, B - 1);
    const float* te = temb + (size_t)t_clip[b] * 512 + 16 * ks + 8 * (lane >> 5);
    const f32x8 p = reinterpret_cast<const f32x8*>(pp)[idx];
    const f32x8 tv = *reinterpret_cast<const f32x8*>(te);
    if constexpr (FMODE == 2) {
        f16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)silu(p[j] + tv[j]);
        reinterpret_cast<f16x8*>(s_hi)[idx] = h;
    } else {
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = silu(p[j] + tv[j]);
            hi[j] = (__bf16)v;
            if constexpr (FMODE == 1) lo[j] = (__bf16)(v - (float)hi[j]);
        }
        s_hi[idx] = hi;
        if constexpr (FMODE == 1) s_lo[idx] = lo;
    }
}

// ------------------------------------------------------------------------------------
// FiLM GEMM: E[g][ot] = Wf[ot] * S[g] + bf  for all 3*L StylizationBlocks at once
// (StylizationBlock.emb_layers, transformer.py:57-60,74), output fp16 FT tiles.
// v1: operands straight from L2 into registers; wave tile 4 feature tiles x 2 groups.
// grid (NT/8, ceil(G/4)), 256 threads = 2x2 waves.
// ------------------------------------------------------------------------------------
template <int FMODE>
__global__ __launch_bounds__(256) void k_film_gemm(const void* __restrict__ Wv, const float* __restrict__ bias_ft,
                                                   const void* __restrict__ S_hiv, const void* __restrict__ S_lov,
                                                   f16x16* __restrict__ E, int G, int NT) {
    using OP = typename std::conditional<FMODE == 2, f16x8, bf16x8>::type;
    constexpr bool SPLIT = FMODE == 1;
    const OP* __restrict__ W = reinterpret_cast<const OP*>(Wv);
    const OP* __restrict__ S_hi = reinterpret_cast<const OP*>(S_hiv);
    const OP* __restrict__ S_lo = reinterpret_cast<const OP*>(S_lov);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ot0 = (blockIdx.x * 2 + (wave >> 1)) * 4;
    const int g0 = (blockIdx.y * 2 + (wave & 1)) * 2;
    if (g0 >= G) return;
    const bool two = g0 + 1 < G;
    const int g1 = two ? g0 + 1 : g0;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = splat(0.f);
    const size_t nfw = (size_t)NT * DC_KS_E;
#pragma unroll 2
    for (int ks = 0; ks < DC_KS_E; ++ks) {
        OP a[4], al[4], b[2], bl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = W[((size_t)(ot0 + i) * DC_KS_E + ks) * 64 + lane];
            if constexpr (SPLIT) al[i] = W[(nfw + (size_t)(ot0 + i) * DC_KS_E + ks) * 64 + lane];
        }
        b[0] = S_hi[((size_t)g0 * DC_KS_E + ks) * 64 + lane];
        b[1] = S_hi[((size_t)g1 * DC_KS_E + ks) * 64 + lane];
        if constexpr (SPLIT) {
            bl[0] = S_lo[((size_t)g0 * DC_KS_E + ks) * 64 + lane];
            bl[1] = S_lo[((size_t)g1 * DC_KS_E + ks) * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mfma(a[i], b[j], acc[i][j]);
                if constexpr (SPLIT) {
                    acc[i][j] = mfma(a[i], bl[j], acc[i][j]);
                    acc[i][j] = mfma(al[i], b[j], acc[i][j]);
                }
            }
    }
    const int hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x16 bb = ld_ft(bias_ft, ot0 + i, hh);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two) continue;
            f16x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = (_Float16)(acc[i][j][r] + bb[r]);
            E[((size_t)(g0 + j) * NT + ot0 + i) * 64 + lane] = o;
        }
    }
}

// ------------------------------------------------------------------------------------
// step prologue: h = joint_embed(x) + sequence_embedding[:T] (transformer.py:488-490),
// then layer 0's self-attention front half.  One wave per group.
// ------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_embed_front(const DcModel* __restrict__ dm, const float* __restrict__ x /*[M][P]*/,
                                                     float* __restrict__ hbuf, float* __restrict__ recs,
                                                     const int* __restrict__ length, int M, int T, int G) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int lane = threadIdx.x & 63;
    const GroupCtx cx = make_ctx(g, lane, M, T);
    const int P = dm->input_feats;
    const bool live = cx.tok < M;
    const int n = live ? cx.tok % T : 0;
    // x as chained-order operand: element j of k-step s <-> pose feature 16s + 8(j>>2) + 4hh + (j&3)
    XFrag<SPLIT> xf[1];
    {
        f32x16 xv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            xv[r] = (live && f < P) ? x[(size_t)cx.tok * P + f] : 0.f;
        }
        make_frag<SPLIT>(xv, xf[0]);
    }
    f32x16 h[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) h[t] = ld_ft(dm->je_b, t, cx.hh);
    gemm_wa<4, 1, SPLIT>(h, dm->je_w, xf, lane);
    const float* se = dm->seq_emb + (size_t)n * DC_D;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(se + 32 * t + 8 * q + 4 * cx.hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) h[t][4 * q + i] += v[i];
        }
    store_h(h, hbuf, g, lane);
    sa_front<SPLIT>(h, dm->layer[0], cx, M, T, length, recs + (size_t)g * 2 * DC_REC_FLOATS);
}

// ------------------------------------------------------------------------------------
// one decoder layer for one token group, from the attention matrices onward:
//   SA back half (transformer.py:104,109,119-121) -> CA (:147,150,156-157) -> FFN (:170-173)
//   then either the next layer's SA front half, or the output projection (:496) fused with the
//   DDIM update (gaussian_diffusion.py:812-830).
// out_mode 0: write pred_xstart to xout;  1: DDIM update of xio in place (+ snapshot).
// ------------------------------------------------------------------------------------
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_layer(const DcModel* __restrict__ dm, int l, float* __restrict__ hbuf,
                                               const f16x16* __restrict__ E, int NT,
                                               const bf16x8* __restrict__ a_sa /*[B][16][64]*/,
                                               const bf16x8* __restrict__ a_ca /*[L][B][16][64]*/,
                                               float* __restrict__ recs, const int* __restrict__ length,
                                               const float* __restrict__ xin, float* __restrict__ xout, int out_mode,
                                               const float* __restrict__ coef_cur, const int* __restrict__ snap_cur,
                                               float* __restrict__ snaps, int M, int T, int G, int B, int dbg) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int lane = threadIdx.x & 63;
    const GroupCtx cx = make_ctx(g, lane, M, T);
    const DcLayer& L = dm->layer[l];
    const int nl = dm->num_layers;
    const f16x16* Eg = E + ((size_t)g * NT + (size_t)l * 24) * 64;   // this layer's 3 blocks x 8 tiles

    f32x16 h[4];
    load_h(h, hbuf, g, lane);

    // ---------------- self-attention, back half ----------------
    {
        f32x16 q[4];
        {
            f32x16 n[4];
            layernorm_ft<4>(h, n, L.sa_ln_g, L.sa_ln_b, cx.hh);
            XFrag<SPLIT> nf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) make_frag<SPLIT>(n[t], nf[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = ld_ft(L.sa_bq, t, cx.hh);
            gemm_wa<4, 4, SPLIT>(q, L.sa_wq, nf, lane);
        }
        softmax_heads_ft(q);
        XFrag<SPLIT> qf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) make_frag<SPLIT>(q[t], qf[t]);
        f32x16 y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = splat(0.f);
        if (!cx.straddle) {
            attn_apply<SPLIT>(y, a_sa + (size_t)cx.b0 * 16 * 64, qf, lane);
        } else {
            XFrag<SPLIT> qm[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { qm[t] = qf[t]; mask_frag<SPLIT>(qm[t], cx.lane_in_b0); }
            attn_apply<SPLIT>(y, a_sa + (size_t)cx.b0 * 16 * 64, qm, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) { qm[t] = qf[t]; mask_frag<SPLIT>(qm[t], !cx.lane_in_b0); }
            attn_apply<SPLIT>(y, a_sa + (size_t)cx.b1 * 16 * 64, qm, lane);
        }
        f32x16 o[4];
        stylization<SPLIT>(o, y, Eg, L.sa_styl, lane, cx.hh);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] += o[t];
    }
    if (dbg == 1) { store_h(h, hbuf, g, lane); return; }   // test hook: stop after self-attention
    // ---------------- cross-attention ----------------
    {
        f32x16 q[4];
        {
            f32x16 n[4];
            layernorm_ft<4>(h, n, L.ca_ln_g, L.ca_ln_b, cx.hh);
            XFrag<SPLIT> nf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) make_frag<SPLIT>(n[t], nf[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = ld_ft(L.ca_bq, t, cx.hh);
            gemm_wa<4, 4, SPLIT>(q, L.ca_wq, nf, lane);
        }
        softmax_heads_ft(q);
        XFrag<SPLIT> qf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) make_frag<SPLIT>(q[t], qf[t]);
        f32x16 y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = splat(0.f);
        const bf16x8* acl = a_ca + (size_t)l * B * 16 * 64;
        if (!cx.straddle) {
            attn_apply<SPLIT>(y, acl + (size_t)cx.b0 * 16 * 64, qf, lane);
        } else {
            XFrag<SPLIT> qm[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { qm[t] = qf[t]; mask_frag<SPLIT>(qm[t], cx.lane_in_b0); }
            attn_apply<SPLIT>(y, acl + (size_t)cx.b0 * 16 * 64, qm, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) { qm[t] = qf[t]; mask_frag<SPLIT>(qm[t], !cx.lane_in_b0); }
            attn_apply<SPLIT>(y, acl + (size_t)cx.b1 * 16 * 64, qm, lane);
        }
        f32x16 o[4];
        stylization<SPLIT>(o, y, Eg + 8 * 64, L.ca_styl, lane, cx.hh);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] += o[t];
    }
    if (dbg == 2) { store_h(h, hbuf, g, lane); return; }   // test hook: stop after cross-attention
    // ---------------- FFN ----------------
    {
        XFrag<SPLIT> hf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) make_frag<SPLIT>(h[t], hf[t]);
        f32x16 u[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) u[t] = ld_ft(L.ffn_b1, t, cx.hh);
        gemm_wa<2, 4, SPLIT>(u, L.ffn_w1, hf, lane);
        XFrag<SPLIT> uf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) u[t][r] = gelu_erf(u[t][r]);
            make_frag<SPLIT>(u[t], uf[t]);
        }
        f32x16 y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = ld_ft(L.ffn_b2, t, cx.hh);
        gemm_wa<4, 2, SPLIT>(y, L.ffn_w2, uf, lane);
        f32x16 o[4];
        stylization<SPLIT>(o, y, Eg + 16 * 64, L.ffn_styl, lane, cx.hh);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] += o[t];
    }

    if (dbg == 3) { store_h(h, hbuf, g, lane); return; }   // test hook: stop after the FFN
    if (l + 1 < nl) {
        store_h(h, hbuf, g, lane);
        sa_front<SPLIT>(h, dm->layer[l + 1], cx, M, T, length, recs + (size_t)g * 2 * DC_REC_FLOATS);
        return;
    }
    // ---------------- output projection + DDIM update ----------------
    XFrag<SPLIT> hf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) make_frag<SPLIT>(h[t], hf[t]);
    f32x16 x0[1];
    x0[0] = ld_ft(dm->out_b, 0, cx.hh);
    gemm_wa<1, 4, SPLIT>(x0, dm->out_w, hf, lane);
    if (cx.tok >= M) return;
    const int P = dm->input_feats;
    if (out_mode == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            if (f < P) xout[(size_t)cx.tok * P + f] = x0[0][r];
        }
    } else {
        const float sr = coef_cur[0], srm1 = coef_cur[1], cx0 = coef_cur[2], ceps = coef_cur[3];
        const int snap = *snap_cur;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            if (f < P) {
                const size_t o = (size_t)cx.tok * P + f;
                const float xt = xin[o];
                const float eps = (sr * xt - x0[0][r]) / srm1;
                const float xn = x0[0][r] * cx0 + ceps * eps;
                xout[o] = xn;
                if (snap >= 0) snaps[(size_t)snap * M * P + o] = xn;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// host-callable launchers (declared in dc_launch.h)
// ------------------------------------------------------------------------------------
#include "dc_launch.h"

#define LAUNCH_CHECK() (hipGetLastError())

hipError_t dc_launch_begin_step(hipStream_t st, int* iter, const int* t_of_iter, const float* coef_of_t,
                                const int* snap_of_iter, int* t_clip, float* coef_cur, int* snap_cur, int B) {
    hipLaunchKernelGGL(k_begin_step, dim3(1), dim3(64), 0, st, iter, t_of_iter, coef_of_t, snap_of_iter, t_clip,
                       coef_cur, snap_cur, B);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_temb_table(hipStream_t st, const float* freqs, const float* w0t, const float* b0,
                                const float* w2t, const float* b2, float* temb, int nt) {
    hipLaunchKernelGGL(k_temb_table, dim3(nt), dim3(512), 0, st, freqs, w0t, b0, w2t, b2, temb);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_cond_linear(hipStream_t st, const float* xf, const float* wt, const float* b, float* y, int M, int Mpad) {
    const size_t n = (size_t)Mpad * 512;
    hipLaunchKernelGGL(k_cond_linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xf, wt, b, y, M, Mpad);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_row_stats(hipStream_t st, const float* y, float* mean, float* rstd, int Mpad) {
    hipLaunchKernelGGL(k_row_stats512, dim3((Mpad + 3) / 4), dim3(256), 0, st, y, mean, rstd, Mpad);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_cond_pack(hipStream_t st, int mode, const float* y, const float* mean, const float* rstd,
                               float* out_f32, void* out_hi, void* out_lo, int G) {
    const size_t n = (size_t)G * 32 * 64;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (mode == 0)
        hipLaunchKernelGGL(k_cond_pack<0>, grid, dim3(256), 0, st, y, mean, rstd, out_f32, (bf16x8*)out_hi, (bf16x8*)out_lo, G);
    else
        hipLaunchKernelGGL(k_cond_pack<1>, grid, dim3(256), 0, st, y, mean, rstd, out_f32, (bf16x8*)out_hi, (bf16x8*)out_lo, G);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_ca_partials(hipStream_t st, bool split, const DcModel* dm, const void* nh_hi, const void* nh_lo,
                                 float* recs, int M, int T, int G, int L) {
    const dim3 grid((G + 3) / 4, L);
    if (split)
        hipLaunchKernelGGL(k_cond_ca_partials<true>, grid, dim3(256), 0, st, dm, (const bf16x8*)nh_hi, (const bf16x8*)nh_lo, recs, M, T, G);
    else
        hipLaunchKernelGGL(k_cond_ca_partials<false>, grid, dim3(256), 0, st, dm, (const bf16x8*)nh_hi, (const bf16x8*)nh_lo, recs, M, T, G);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_attn_combine(hipStream_t st, const float* recs, void* afrag, int T, int G, int B, int nset) {
    const int ng_max = T / 32 + 2;
    const size_t shm = (size_t)(ng_max * 32 + 32 + 256) * sizeof(float);
    hipLaunchKernelGGL(k_attn_combine, dim3(B, 4, nset), dim3(256), shm, st, recs, (bf16x8*)afrag, T, G, B);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_silu_emb(hipStream_t st, int fmode, const float* pp, const float* temb, const int* t_clip,
                              void* s_hi, void* s_lo, int G, int T, int B) {
    const size_t n = (size_t)G * 32 * 64;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (fmode == 1)
        hipLaunchKernelGGL(k_silu_emb<1>, grid, dim3(256), 0, st, pp, temb, t_clip, (bf16x8*)s_hi, (bf16x8*)s_lo, G, T, B);
    else if (fmode == 2)
        hipLaunchKernelGGL(k_silu_emb<2>, grid, dim3(256), 0, st, pp, temb, t_clip, (bf16x8*)s_hi, (bf16x8*)s_lo, G, T, B);
    else
        hipLaunchKernelGGL(k_silu_emb<0>, grid, dim3(256), 0, st, pp, temb, t_clip, (bf16x8*)s_hi, (bf16x8*)s_lo, G, T, B);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_film_gemm(hipStream_t st, int fmode, const void* W, const float* bias_ft, const void* s_hi,
                               const void* s_lo, void* E, int G, int NT) {
    const dim3 grid(NT / 8, (G + 3) / 4);
    if (fmode == 1)
        hipLaunchKernelGGL(k_film_gemm<1>, grid, dim3(256), 0, st, W, bias_ft, s_hi, s_lo, (f16x16*)E, G, NT);
    else if (fmode == 2)
        hipLaunchKernelGGL(k_film_gemm<2>, grid, dim3(256), 0, st, W, bias_ft, s_hi, s_lo, (f16x16*)E, G, NT);
    else
        hipLaunchKernelGGL(k_film_gemm<0>, grid, dim3(256), 0, st, W, bias_ft, s_hi, s_lo, (f16x16*)E, G, NT);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_embed_front(hipStream_t st, bool split, const DcModel* dm, const float* x, float* hbuf, float* recs,
                                 const int* length, int M, int T, int G) {
    const dim3 grid((G + 3) / 4);
    if (split)
        hipLaunchKernelGGL(k_embed_front<true>, grid, dim3(256), 0, st, dm, x, hbuf, recs, length, M, T, G);
    else
        hipLaunchKernelGGL(k_embed_front<false>, grid, dim3(256), 0, st, dm, x, hbuf, recs, length, M, T, G);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_layer(hipStream_t st, bool split, const DcModel* dm, int l, float* hbuf, const void* E, int NT,
                           const void* a_sa, const void* a_ca, float* recs, const int* length, const float* xin,
                           float* xout, int out_mode, const float* coef_cur, const int* snap_cur, float* snaps,
                           int M, int T, int G, int B, int dbg) {
    const dim3 grid((G + 3) / 4);
    if (split)
        hipLaunchKernelGGL(k_layer<true>, grid, dim3(256), 0, st, dm, l, hbuf, (const f16x16*)E, NT, (const bf16x8*)a_sa,
                           (const bf16x8*)a_ca, recs, length, xin, xout, out_mode, coef_cur, snap_cur, snaps, M, T, G, B, dbg);
    else
        hipLaunchKernelGGL(k_layer<false>, grid, dim3(256), 0, st, dm, l, hbuf, (const f16x16*)E, NT, (const bf16x8*)a_sa,
                           (const bf16x8*)a_ca, recs, length, xin, xout, out_mode, coef_cur, snap_cur, snaps, M, T, G, B, dbg);
    return LAUNCH_CHECK();
}
