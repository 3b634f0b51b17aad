// dc_dev.h - device-side building blocks shared by dc_kernels.hip and dc_fused.hip (gfx950 only): MFMA fragment
// helpers, LayerNorm / softmax / SiLU / GELU on accumulator-layout tiles, LDS-DMA, the stylization tile math.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "dc_common.h"

#define DEV __device__ __forceinline__

namespace dc {

template <class T> struct V8;
template <> struct V8<__bf16> { using type = bf16x8; };
template <> struct V8<_Float16> { using type = f16x8; };
template <class T> using v8 = typename V8<T>::type;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <class T> struct V4;
template <> struct V4<__bf16> { using type = bf16x4; };
template <> struct V4<_Float16> { using type = f16x4; };
template <class T> using v4 = typename V4<T>::type;

DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEV f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
DEV f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// value of the partner lane (lane ^ 32) combined with this lane's: one v_permlane32_swap, no LDS
DEV float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_max_f32 as the hardware does it.  fmaxf() makes hipcc canonicalise both inputs first (v_max_f32 x, x, x - two extra instructions per
// maximum, and a third for a negated result): the values compared here are accumulators or maxima of accumulators, where a signalling
// NaN cannot occur and a quiet one propagates to the result either way (it ends in the status word's NONFINITE bit).
DEV float max2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
DEV float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV float exp2f_fast(float x) { return __builtin_amdgcn_exp2f(x); }     // v_exp_f32

// acc + the sum of the 8 elements of an operand fragment register group (v_dot2c_f32_f16 / _bf16 against ones)
DEV float sum8(f16x8 a, float acc) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 one = {(_Float16)1.f, (_Float16)1.f};
    acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 0, 1), one, acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 2, 3), one, acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 4, 5), one, acc, false);
    return __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 6, 7), one, acc, false);
}
DEV float sum8(bf16x8 a, float acc) {
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 one = {(__bf16)1.f, (__bf16)1.f};
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), one, acc, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), one, acc, false);
}

// the sum of the 16 elements of two operand fragment register groups by a packed-f16 tree (7 v_pk_add_f16 + the last pair in f32)
DEV float sum16_pk(f16x8 a, f16x8 b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const f16x8 s8 = a + b;                                                       // 4 v_pk_add_f16
    const f16x4 s4 = __builtin_shufflevector(s8, s8, 0, 1, 2, 3) + __builtin_shufflevector(s8, s8, 4, 5, 6, 7);
    const h2 s2 = __builtin_shufflevector(s4, s4, 0, 1) + __builtin_shufflevector(s4, s4, 2, 3);
    return (float)s2[0] + (float)s2[1];
}

// row (feature in FT, token in TF) held by register r of lane-half hh
DEV int tile_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

DEV f32x16 splat(float v) {
    f32x16 x;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = v;
    return x;
}

// One 32-row tile as MFMA operand fragments for its two 16-deep k-steps (hi [+ lo]).
template <class T16, bool SPLIT>
struct XFrag {
    v8<T16> hi[2];
    v8<T16> lo[SPLIT ? 2 : 1];
};

template <class T16, bool SPLIT>
DEV void make_frag(const f32x16& x, XFrag<T16, SPLIT>& f) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[8 * s + j];
            const T16 h = (T16)v;
            f.hi[s][j] = h;
            if constexpr (SPLIT) f.lo[s][j] = (T16)(v - (float)h);
        }
}

template <class T16, bool SPLIT>
DEV void mask_frag(XFrag<T16, SPLIT>& f, bool keep) {
    if (!keep) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f.hi[s][j] = (T16)0.f;
                if constexpr (SPLIT) f.lo[s][j] = (T16)0.f;
            }
    }
}

// Weight image order (chained pack): [hi | lo][kt][ot][s][64 lanes][8]  (kt-major).  NF = OT*KT*2 frags per half.
//
// acc[ot] (rows = output features, cols = tokens) += W[ot][kt] * X[kt] for ONE k-tile; weights are the A operand.
template <int OT, int KT, class T16, bool SPLIT>
DEV void mma_kt(f32x16 (&acc)[OT], const v8<T16>* __restrict__ w, int kt, const XFrag<T16, SPLIT>& x, int lane) {
    constexpr int NF = OT * KT * 2;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fi = (kt * OT + ot) * 2 + s;
            const v8<T16> a = w[fi * 64 + lane];
            acc[ot] = mfma(a, x.hi[s], acc[ot]);
            if constexpr (SPLIT) {
                acc[ot] = mfma(a, x.lo[s], acc[ot]);
                const v8<T16> al = w[(NF + fi) * 64 + lane];
                acc[ot] = mfma(al, x.hi[s], acc[ot]);
            }
        }
}
template <int OT, int KT, class T16, bool SPLIT>
DEV void gemm_wa(f32x16 (&acc)[OT], const v8<T16>* __restrict__ w, const XFrag<T16, SPLIT> (&x)[KT], int lane) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        mma_kt<OT, KT, T16, SPLIT>(acc, w, kt, x[kt], lane);
        __builtin_amdgcn_sched_barrier(0);      // bounds the operand-read lookahead to one k-tile (register pressure)
    }
}

// one output tile: acc (rows = output features of tile ot) += sum_kt W[ot][kt] * X[kt]
template <int OT, int KT, class T16, bool SPLIT>
DEV void mma_ot(f32x16& acc, const v8<T16>* __restrict__ w, int ot, const XFrag<T16, SPLIT> (&x)[KT], int lane) {
    constexpr int NF = OT * KT * 2;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fi = (kt * OT + ot) * 2 + s;
            const v8<T16> a = w[fi * 64 + lane];
            acc = mfma(a, x[kt].hi[s], acc);
            if constexpr (SPLIT) {
                acc = mfma(a, x[kt].lo[s], acc);
                const v8<T16> al = w[(NF + fi) * 64 + lane];
                acc = mfma(al, x[kt].hi[s], acc);
            }
        }
}

// acc (rows = tokens, cols = output features of tile oc) += X^T * W[oc] over all k-tiles; weights are the B operand.
template <int OC, int KT, class T16, bool SPLIT>
DEV void mmb_oc(f32x16& acc, const v8<T16>* __restrict__ w, int oc, const XFrag<T16, SPLIT> (&x)[KT], int lane) {
    constexpr int NF = OC * KT * 2;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fi = (kt * OC + oc) * 2 + s;
            const v8<T16> b = w[fi * 64 + lane];
            acc = mfma(x[kt].hi[s], b, acc);
            if constexpr (SPLIT) {
                acc = mfma(x[kt].lo[s], b, acc);
                const v8<T16> bl = w[(NF + fi) * 64 + lane];
                acc = mfma(x[kt].hi[s], bl, acc);
            }
        }
}

// two independent accumulator chains interleaved (non-split): tiles (wa, oca) and (wb, ocb) of the same operand x.
// A single tile is a chain of 8 dependent MFMAs (~70 cycles each); two chains keep the matrix pipe busy.
template <int OC, int KT, class T16>
DEV void mmb_oc_pair(f32x16& acca, f32x16& accb, const v8<T16>* __restrict__ wa, int oca, const v8<T16>* __restrict__ wb, int ocb,
                     const XFrag<T16, false> (&x)[KT], int lane) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const v8<T16> ba = wa[((kt * OC + oca) * 2 + s) * 64 + lane];
            const v8<T16> bb = wb[((kt * OC + ocb) * 2 + s) * 64 + lane];
            acca = mfma(x[kt].hi[s], ba, acca);
            accb = mfma(x[kt].hi[s], bb, accb);
        }
}

// all four output tiles of one 128 x 128 image at once: four independent chains (split: three products per fragment)
template <int OC, int KT, class T16, bool SPLIT = false>
DEV void mmb_oc_quad(f32x16& a, f32x16& b, f32x16& c, f32x16& d, const v8<T16>* __restrict__ w, const XFrag<T16, SPLIT> (&x)[KT], int lane) {
    constexpr int NF = OC * KT * 2;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const v8<T16> fa = w[((kt * OC + 0) * 2 + s) * 64 + lane], fb = w[((kt * OC + 1) * 2 + s) * 64 + lane];
            const v8<T16> fc = w[((kt * OC + 2) * 2 + s) * 64 + lane], fd = w[((kt * OC + 3) * 2 + s) * 64 + lane];
            a = mfma(x[kt].hi[s], fa, a);
            b = mfma(x[kt].hi[s], fb, b);
            c = mfma(x[kt].hi[s], fc, c);
            d = mfma(x[kt].hi[s], fd, d);
            if constexpr (SPLIT) {
                a = mfma(x[kt].lo[s], fa, a);
                b = mfma(x[kt].lo[s], fb, b);
                c = mfma(x[kt].lo[s], fc, c);
                d = mfma(x[kt].lo[s], fd, d);
                const v8<T16> la = w[(NF + (kt * OC + 0) * 2 + s) * 64 + lane], lb = w[(NF + (kt * OC + 1) * 2 + s) * 64 + lane];
                const v8<T16> lc = w[(NF + (kt * OC + 2) * 2 + s) * 64 + lane], ld = w[(NF + (kt * OC + 3) * 2 + s) * 64 + lane];
                a = mfma(x[kt].hi[s], la, a);
                b = mfma(x[kt].hi[s], lb, b);
                c = mfma(x[kt].hi[s], lc, c);
                d = mfma(x[kt].hi[s], ld, d);
            }
        }
}

// per-feature vector stored as [tile][lane-half][16] so a lane reads its 16 values with one 64-B load
DEV f32x16 ld_ft(const float* __restrict__ p, int tile, int hh) {
    return *reinterpret_cast<const f32x16*>(p + (tile * 2 + hh) * 16);
}

#ifndef DC_SCALAR_F32
typedef float f32x2 __attribute__((ext_vector_type(2)));   // operand of the packed-fp32 VALU ops (v_pk_add/mul/fma_f32)
DEV f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
// A/B build (tools/ab.sh build S -DDC_SCALAR_F32 -fno-slp-vectorize): the same arithmetic as scalar v_add / v_mul / v_fma_f32 -
// MI355X_MICROARCH.md prices packed fp32 beside MFMAs as an anti-lever; measured on this kernel in DESIGN.md section 4
struct f32x2 {
    float x, y;
};
DEV f32x2 operator+(f32x2 a, f32x2 b) { return {a.x + b.x, a.y + b.y}; }
DEV f32x2 operator-(f32x2 a, f32x2 b) { return {a.x - b.x, a.y - b.y}; }
DEV f32x2 operator*(f32x2 a, f32x2 b) { return {a.x * b.x, a.y * b.y}; }
DEV f32x2 operator+(f32x2 a, float b) { return {a.x + b, a.y + b}; }
DEV f32x2 operator*(f32x2 a, float b) { return {a.x * b, a.y * b}; }
DEV f32x2& operator+=(f32x2& a, f32x2 b) { a.x += b.x; a.y += b.y; return a; }
DEV f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#endif
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;
// nn.LayerNorm(128) over the feature axis of an FT activation (transformer.py:79,104,147)
template <int NT>
DEV void ln_stats(const f32x16 (&x)[NT], float& mean, float& rstd) {
    // one pass: sum and sum of squares (fp32, 128 terms: the cancellation in E[x^2]-mean^2 stays ~1e-7*mean^2/var)
    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x2 v = {x[t][2 * r], x[t][2 * r + 1]};
            s2 += v;
            q2 = fma2(v, v, q2);
        }
    const float s = xhalf_sum(s2.x + s2.y);
    const float q = xhalf_sum(q2.x + q2.y);
    mean = s * (1.f / (32 * NT));
    const float var = fmaxf(fmaf(-mean, mean, q * (1.f / (32 * NT))), 0.f);
    rstd = __builtin_amdgcn_rsqf(var + 1e-5f);       // v_rsq_f32 (1 ulp; var + eps >= 1e-5 is never subnormal: rsqrtf()'s range scaling - five more instructions - buys nothing)
}
// operand fragments of the normalised x (the LayerNorm affine is folded into the projection that follows)
template <class T16, bool SPLIT>
DEV void ln_frags(XFrag<T16, SPLIT> (&nf)[4], const f32x16 (&x)[4]) {
    float mean, rstd;
    ln_stats<4>(x, mean, rstd);
    const float shift = -mean * rstd;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        f32x16 n;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x2 v = {x[kt][2 * r], x[kt][2 * r + 1]};
            const f32x2 w = fma2(v, (f32x2){rstd, rstd}, (f32x2){shift, shift});
            n[2 * r] = w.x;
            n[2 * r + 1] = w.y;
        }
        make_frag<T16, SPLIT>(n, nf[kt]);
    }
}

// F.softmax(query.view(B,T,H,-1), dim=-1) (transformer.py:109,150): a head = 16 features
// = registers 8p..8p+7 of this lane and of lane^32.
// Packed-fp32 formulation (v_pk_add / v_pk_mul_f32 on register pairs, 3-input maxima): 29 instead of 43 VALU instructions per
// head half - the layer kernel is bound by instruction issue.
DEV float max3(float a, float b, float c) {       // (the compiler forms v_max3_f32 from nested fmaxf only now and then)
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
DEV float absmax3(float a, float b, float c) {    // max(|a|, |b|, |c|)
    float d;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
DEV float max8(const f32x16& q, int o) {
    return max3(max3(q[o], q[o + 1], q[o + 2]), max3(q[o + 3], q[o + 4], q[o + 5]), max2(q[o + 6], q[o + 7]));
}
DEV void softmax_heads_ft(f32x16 (&q)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float m = xhalf_max(max8(q[t], 8 * p));
            const f32x2 nm = {-m, -m};
            f32x2 e[4], s2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 d = (f32x2){q[t][8 * p + 2 * j], q[t][8 * p + 2 * j + 1]} + nm;
                e[j] = (f32x2){exp2f_fast(d.x), exp2f_fast(d.y)};      // q carries log2(e): folded into Wq, bq
                s2 += e[j];
            }
            const float inv = fast_rcp(xhalf_sum(s2.x + s2.y));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 r = e[j] * inv;
                q[t][8 * p + 2 * j] = r.x;
                q[t][8 * p + 2 * j + 1] = r.y;
            }
        }
}
// column maximum of a 32-row tile held in 16 registers (this lane's half of the rows)
DEV float max16(const f32x16& k) {
    return max3(max3(max3(k[0], k[1], k[2]), max3(k[3], k[4], k[5]), max3(k[6], k[7], k[8])),
                max3(k[9], k[10], k[11]), max3(max3(k[12], k[13], k[14]), k[15], k[15]));
}
// e[r] = exp2(k[r] - m) for the 16 registers, and their sum (packed subtractions / additions)
DEV float exp_rows(f32x16& e, const f32x16& k, float m) {
    const f32x2 nm = {-m, -m};
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x2 d = (f32x2){k[2 * j], k[2 * j + 1]} + nm;
        const f32x2 x = {exp2f_fast(d.x), exp2f_fast(d.y)};
        e[2 * j] = x.x;
        e[2 * j + 1] = x.y;
        s2 += x;
    }
    return s2.x + s2.y;
}

DEV float silu(float z) { return z * fast_rcp(1.f + exp2f_fast(-1.4426950408889634f * z)); }
// SiLU on log2(e)-scaled arguments, two elements at a time: u = log2(e) x  ->  u / (1 + 2^-u) = log2(e) SiLU(x).
// The StylizationBlocks run in this scaling (the host folds log2(e) into the H' tiles and ln 2 into W_o; the caller scales
// rstd / shift), which removes the per-element multiply in front of v_exp_f32; the add and the product are packed-fp32 ops.
DEV f32x2 silu_pair(float x0, float x1) {          // plain SiLU of two elements, packed-fp32 products and sums
    const f32x2 x = {x0, x1};
    const f32x2 u = x * 1.4426950408889634f;
    f32x2 e = {exp2f_fast(-u.x), exp2f_fast(-u.y)};
    e = e + 1.f;
    const f32x2 r = {fast_rcp(e.x), fast_rcp(e.y)};
    return x * r;
}
DEV f32x2 silu_l2_pair(float u0, float u1) {
    const f32x2 u = {u0, u1};
    f32x2 e = {exp2f_fast(-u0), exp2f_fast(-u1)};
    e = e + 1.f;
    const f32x2 r = {fast_rcp(e.x), fast_rcp(e.y)};
    return u * r;
}
// nn.GELU() (exact-erf form).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 noise of the
// surrounding GEMMs): erf(a) = 1 - (a1 t + ... + a5 t^5) exp(-a^2), t = 1/(1 + p a), a >= 0.
DEV float gelu_erf(float x) {
    const float a = fabsf(x) * 0.70710678118654752440f;
    const float t = fast_rcp(fmaf(0.3275911f, a, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erfa = 1.f - poly * t * __expf(-a * a);
    return 0.5f * x * (1.f + copysignf(erfa, x));
}
// the same on two elements with packed-fp32 arithmetic: gelu(x) = x/2 + |x|/2 * erf(|x| / sqrt 2)  (x sign(x) = |x|).
// erf by Abramowitz-Stegun 7.1.28: 1 - erf(a) = (1 + a1 a + ... + a6 a^6)^-16 (|error| <= 3e-7), the 1/sqrt 2 of a = |x| / sqrt 2 folded
// into the coefficients: one v_rcp_f32 per element and no v_exp_f32 (transcendentals issue at a quarter of the packed-fp32 rate; 7.1.26
// - one rcp AND one exp - stays for the scalar form and under -DDC_GELU_AS26).  Against fp64 over |x| <= 12: max |error| of the GELU
// 8.8e-7 (7.1.26: 4.7e-7); beyond |x| ~ 19 the sixteenth power overflows to +inf and its reciprocal is the exact limit 0.
DEV f32x2 gelu_erf_pair(float x0, float x1) {
    const f32x2 x = {x0, x1};
    const f32x2 ax = {fabsf(x0), fabsf(x1)};
    const f32x2 half_ax = ax * 0.5f;
#ifdef DC_GELU_AS26
    const f32x2 den = fma2(ax, (f32x2){0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f},
                                                (f32x2){1.f, 1.f});
    const f32x2 t = {fast_rcp(den.x), fast_rcp(den.y)};
    f32x2 poly = fma2(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    poly = fma2(poly, t, (f32x2){1.421413741f, 1.421413741f});
    poly = fma2(poly, t, (f32x2){-0.284496736f, -0.284496736f});
    poly = fma2(poly, t, (f32x2){0.254829592f, 0.254829592f});
    const f32x2 xx = x * x * (-0.5f * 1.4426950408889634f);              // exp(-a^2) = 2^(-x^2 log2(e) / 2)
    const f32x2 ex = {exp2f_fast(xx.x), exp2f_fast(xx.y)};
    const f32x2 pe = poly * t * ex;                                    // 1 - erf(a)
#else
    constexpr float c1 = 0.0705230784f * 0.70710678118654752440f, c2 = 0.0422820123f * 0.5f, c3 = 0.0092705272f * 0.35355339059327376220f,
                    c4 = 0.0001520143f * 0.25f, c5 = 0.0002765672f * 0.17677669529663688110f, c6 = 0.0000430638f * 0.125f;
    f32x2 p = fma2(ax, (f32x2){c6, c6}, (f32x2){c5, c5});
    p = fma2(p, ax, (f32x2){c4, c4});
    p = fma2(p, ax, (f32x2){c3, c3});
    p = fma2(p, ax, (f32x2){c2, c2});
    p = fma2(p, ax, (f32x2){c1, c1});
    p = fma2(p, ax, (f32x2){1.f, 1.f});
    f32x2 pe = {fast_rcp(p.x), fast_rcp(p.y)};
    pe = pe * pe;
    pe = pe * pe;
    pe = pe * pe;
    pe = pe * pe;                                                       // 1 - erf(a)
#endif
    // x/2 + |x|/2 (1 - pe)
    return fma2(half_ax, (f32x2){1.f, 1.f} - pe, x * 0.5f);
}

// XCD-aware workgroup index: blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so logical workgroup
// ids are handed out in contiguous runs per XCD - the ~8 workgroups of a clip then share one L2 for the clip's unit
// records and attention fragments.  Bijective for any grid size; affects speed only (placement is not guaranteed).
DEV int wg_index() {
    const int n = gridDim.x, b = blockIdx.x;
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return x * q + min(x, r) + i;
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

// Rows (tokens, TF layout) of a group that belong to the slot's clip and are unmasked form one interval
// [lo, lo + span) of group-local row indices (src_mask of transformer.py:107,114; `length` == nullptr means no
// mask, as in cross-attention).  Stored per lane with the lane-half offset folded in: register r is valid
// iff (unsigned)(crow(r) - lo) < span, crow(r) = (r&3) + 8*(r>>2).
struct RowRange {
    int lo;
    unsigned span;      // wave-uniform; span == 32 means every row of the group is valid (the common case: no masking code)
};
DEV RowRange valid_rows(const GroupCtx& cx, int slot, int M, int T, const int* __restrict__ length, int len_all = -1) {
    const int bs = slot == 0 ? cx.b0 : cx.b1;
    const int len = length ? length[bs] : (len_all >= 0 ? len_all : T);      // len_all: frames per clip when the clip stride T is padded
    const int first = max(bs * T, 32 * cx.g);                         // first valid token
    const int end = min(min(bs * T + min(len, T), M), 32 * cx.g + 32);   // one past the last valid token
    RowRange rr;
    rr.lo = first - 32 * cx.g - 4 * cx.hh;
    rr.span = end > first ? (unsigned)(end - first) : 0u;
    return rr;
}
DEV bool row_ok(const RowRange& rr, int r) { return (unsigned)(((r & 3) + 8 * (r >> 2)) - rr.lo) < rr.span; }

// lanes of the upper 16 columns keep registers 8..15 of an accumulator tile, the others 0..7.  Written as a bit-select:
// as `up ? P[8+i] : P[i]` the compiler forms a dynamic vector index and expands it into a 16-way compare/select chain
// per element (~400 instructions per tile; it was most of the record stage's time).
DEV f32x8 keep_head_block(const f32x16& P, int c) {
    const unsigned mask = (c >> 4) ? 0xffffffffu : 0u;
    f32x8 k;
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = __uint_as_float((__float_as_uint(P[i]) & ~mask) | (__float_as_uint(P[8 + i]) & mask));
    return k;
}

// One 32-feature tile of a group's partial record: column max m, column sum of exp(K-m), and
// exp2(K-m)^T V (32x32; only the two diagonal 16x16 head blocks are stored).
template <class T16, bool SPLIT>
DEV void emit_partial(const f32x16& K, const f32x16& V, int oc, const RowRange& rr, float* __restrict__ R,
                      const GroupCtx& cx) {
    float m = -INFINITY;
    const bool full = __builtin_amdgcn_readfirstlane(rr.span) == 32u;     // no predicates for whole groups (3 instructions per element)
    if (full) {
        m = max16(K);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) m = row_ok(rr, r) ? fmaxf(m, K[r]) : m;
    }
    m = xhalf_max(m);
    if (m == -INFINITY) m = 0.f;
    f32x16 Ee, Vm;
    float ssum = 0.f;
    if (full) {
        ssum = exp_rows(Ee, K, m);                                  // K carries log2(e): folded into Wk, bk
        Vm = V;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = row_ok(rr, r);
            const float e = ok ? exp2f_fast(K[r] - m) : 0.f;
            Ee[r] = e;
            ssum += e;
            Vm[r] = ok ? V[r] : 0.f;
        }
    }
    ssum = xhalf_sum(ssum);
    XFrag<T16, SPLIT> ef, vf;
    make_frag<T16, SPLIT>(Ee, ef);
    make_frag<T16, SPLIT>(Vm, vf);
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
    // keep the diagonal head blocks only: rows 16*(c>>4) .. +15 of this lane's column = registers 8*(c>>4) .. +7
    reinterpret_cast<f32x8*>(R + 256)[oc * 64 + cx.lane] = keep_head_block(P, cx.c);
}

// fp32 fragment image of the step-invariant emb term: [g][ks][2 halves][64 lanes][4 floats] - element j of lane l in half
// j>>2, so each 16-byte access of a wave covers one contiguous KiB
DEV f32x8 ld_pp(const float* __restrict__ pp, size_t frag /* g*32 + ks */, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(pp) + frag * 128 + lane;
    const f32x4 a = p[0], b = p[64];
    f32x8 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = a[i];
        v[4 + i] = b[i];
    }
    return v;
}

// FiLM tile image: [2 halves][64 lanes][8 fp16] - registers 0..7 then 8..15 of each lane, so that both a
// register load and an LDS-DMA copy of the tile are lane-linear 16-B accesses.
DEV void store_etile(f16x16* __restrict__ E, size_t tile, int lane, const f16x16& v) {
    f16x8* p = reinterpret_cast<f16x8*>(E) + tile * 128 + lane;
    f16x8 lo8, hi8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        lo8[i] = v[i];
        hi8[i] = v[8 + i];
    }
    // written once and read by a later kernel: non-temporal stores keep the 708 MB per step from evicting the FiLM
    // weights (and later the layer kernels' own lines) from L2 - measured -1.8 % on the whole loop (tools/ab.sh)
#if defined(DC_DIAG_NO_ESTORE)
    // diagnostic build (timing only, results invalid): conversion and lane swaps kept, the stores never execute (E is never 1)
    if (reinterpret_cast<size_t>(E) == 1) {
        __builtin_nontemporal_store(lo8, p);
        __builtin_nontemporal_store(hi8, p + 64);
    }
#elif defined(DC_E_PLAIN)
    p[0] = lo8;          // experiment: default cache policy for the FiLM tiles (a chunk that is meant to stay in the Infinity Cache)
    p[64] = hi8;
#elif defined(DC_E_SC1)
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(lo8) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p + 64), "v"(hi8) : "memory");
#else
    __builtin_nontemporal_store(lo8, p);
    __builtin_nontemporal_store(hi8, p + 64);
#endif
}
DEV f16x16 load_etile(const f16x8* __restrict__ p /* tile base + lane; global or LDS */) {
    const f16x8 lo8 = p[0], hi8 = p[64];
    f16x16 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = lo8[i];
        v[8 + i] = hi8[i];
    }
    return v;
}

// Residual-stream image: [group][tile t][quarter q][64 lanes][4 floats] - register 4q+i of tile t of lane l.  Each 16-byte
// access of a wave covers one contiguous KiB (the former [t][lane][16] order made every store instruction touch 32
// cache lines a quarter each: the eight waves' stores queued for ~7 us behind one another).
DEV void load_h(f32x16 (&h)[4], const float* __restrict__ hbuf, int g, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(hbuf) + (size_t)g * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[(t * 4 + q) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) h[t][4 * q + i] = v[i];
        }
}
DEV void store_h(const f32x16 (&h)[4], float* __restrict__ hbuf, int g, int lane) {
    f32x4* p = reinterpret_cast<f32x4*>(hbuf) + (size_t)g * 1024 + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = h[t][4 * q + i];
#ifdef DC_H_SC1
            // write-through: the 29 MB of residual stream are not left dirty in the L2s for the end-of-kernel write-back
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + (t * 4 + q) * 64), "v"(v) : "memory");
#else
            // non-temporal: the next kernel reads h after its start-of-kernel invalidate anyway (k_embed_front -6 %, loop -0.3 %)
            __builtin_nontemporal_store(v, p + (t * 4 + q) * 64);
#endif
        }
}

// "Front half" of LinearTemporalSelfAttention (transformer.py:104-117) for one token group, given the operand
// fragments nf of n = LN(h): K = Wk n + bk, V = Wv n + bv in TF form, then the group's partial record(s) of
// softmax_T(K + mask) and K^T V - one per clip the group touches.  bk/bv: plain bias[128] (any address space).
// (A workgroup-level pre-reduction of these records through LDS was measured: it shortened the combine by 30 %
// but cost 2x that in this kernel - 8 waves x 128 live K/V registers - so records stay per group.)
template <class T16, bool SPLIT>
DEV void front_stage(const XFrag<T16, SPLIT> (&nf)[4], const v8<T16>* wk, const v8<T16>* wv, const float* bk,
                     const float* bv, const GroupCtx& cx, int M, int T, const int* __restrict__ length,
                     float* __restrict__ recs, bool active) {
    float* rec = recs + (size_t)cx.g * 2 * DC_REC_FLOATS;
    const RowRange valid0 = valid_rows(cx, 0, M, T, length);
    const RowRange valid1 = valid_rows(cx, cx.straddle ? 1 : 0, M, T, length);
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
        f32x16 K = splat(bk[32 * oc + cx.c]);
        f32x16 V = splat(bv[32 * oc + cx.c]);
        mmb_oc<4, 4, T16, SPLIT>(K, wk, oc, nf, cx.lane);
        mmb_oc<4, 4, T16, SPLIT>(V, wv, oc, nf, cx.lane);
        if (active) {
            emit_partial<T16, SPLIT>(K, V, oc, valid0, rec, cx);
            if (cx.straddle) emit_partial<T16, SPLIT>(K, V, oc, valid1, rec + DC_REC_FLOATS, cx);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the four feature tiles sequential: bounds register pressure
    }
}


// ------------------------------------------------------------------------------------
// Workgroup-level partial records (non-split formats, T >= 256 so that the 256 tokens of a workgroup touch at most
// two clips, "slots" 0/1 = clips ub0, ub0+1).  Instead of one record per 32-token group the 8 waves reduce theirs
// through LDS: column maxima first (so every wave exponentiates against the workgroup's maximum and no rescaling is
// needed), then the exp(K-m)^T V blocks and column sums are summed over the waves in wave order (deterministic).
// One record per workgroup and slot: 8x fewer record bytes to write and to combine, and the combine itself moves into
// the consuming kernel's prologue (wg_combine_attn) - no separate combine launch.
// LDS: mx [4 oc][2 slots][32 cols][8 waves] floats; pst [8 waves][4 oc][64 lanes] f32x8 (each wave's primary slot);
//      xp [4 oc][64 lanes] f32x8 (second slot of the one wave that straddles the clip edge); ss [(8+1)][4 oc][32] floats.
// ------------------------------------------------------------------------------------
DEV RowRange valid_rows_clip(const GroupCtx& cx, int clip, int B, int M, int T, const int* __restrict__ length, bool active) {
    RowRange rr;
    rr.lo = 0;
    rr.span = 0u;
    if (!active || clip >= B) return rr;
    const int len = length ? length[clip] : T;
    const int first = max(clip * T, 32 * cx.g);
    const int end = min(min(clip * T + min(len, T), M), 32 * cx.g + 32);
    rr.lo = first - 32 * cx.g - 4 * cx.hh;
    rr.span = end > first ? (unsigned)(end - first) : 0u;
    return rr;
}
// Workgroup -> token groups of the workgroup-record kernels.
// Flat (upc == 0): workgroup wg owns groups wg*NW .. of the flat token space; a unit (and a group) may span two clips.
// Clip-aligned (upc > 0; the clip stride T is a multiple of 32): clip b = wg / upc owns upc workgroups = units of NW*32 tokens, the
// last one partial - no unit and no group spans two clips, and a clip's result no longer depends on its place in the batch.
// The unit arithmetic of the wg_* helpers (which units make up a clip, record slots) runs in a "unit space" in which every clip is
// Tu = upc*NW*32 tokens long; group / token indices of h, E, pp stay compact (clip stride T).
struct WgMap {
    int g;            // this wave's group (clamped to a valid one when the wave is idle)
    bool active;
    int ub0;          // first clip of the unit
    int nact;         // active waves of the workgroup (a prefix)
    int Mu, Tu;       // unit space: total tokens, clip stride
};
DEV WgMap wg_map(int wg, int wave, int NW, int G, int M, int T, int B, int upc) {
    WgMap w;
    if (upc == 0) {
        w.g = wg * NW + wave;
        w.active = w.g < G;
        if (!w.active) w.g = G - 1;
        w.ub0 = (wg * NW * 32) / T;
        w.nact = min(NW, G - wg * NW);
        w.Mu = M;
        w.Tu = T;
    } else {
        const int gc = T >> 5, b = wg / upc, u = wg - b * upc, gl = NW * u + wave;
        w.active = gl < gc;
        w.g = b * gc + min(gl, gc - 1);
        w.ub0 = b;
        w.nact = min(NW, gc - NW * u);
        w.Tu = upc * NW * 32;
        w.Mu = B * w.Tu;
    }
    return w;
}
template <int NW = 8>
DEV void wg_put_maxes(const f32x16 (&K)[4], const GroupCtx& cx, const RowRange (&vr)[2], float* mx, int wave) {
#pragma unroll
    for (int oc = 0; oc < 4; ++oc)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            float m = -INFINITY;
            const unsigned span = __builtin_amdgcn_readfirstlane(vr[sl].span);
            if (span == 32u) {
                m = max16(K[oc]);
            } else if (span != 0u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) m = row_ok(vr[sl], r) ? fmaxf(m, K[oc][r]) : m;
            }
            m = xhalf_max(m);
            if (cx.hh == 0) mx[((oc * 2 + sl) * 32 + cx.c) * NW + wave] = m;
        }
}
template <int NW = 8>
DEV float wg_colmax(const float* mx, int oc, int sl, int c) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(mx + ((oc * 2 + sl) * 32 + c) * NW);
    float m;
    if constexpr (NW == 8) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(mx + ((oc * 2 + sl) * 32 + c) * 8 + 4);
        m = max3(max3(a[0], a[1], a[2]), max3(a[3], b[0], b[1]), max2(b[2], b[3]));
    } else {
        m = max2(max3(a[0], a[1], a[2]), a[3]);
    }
    return m == -INFINITY ? 0.f : m;
}
// one 32-feature tile against a GIVEN column maximum: column sums of exp2(K-m) and the kept head blocks of exp2(K-m)^T V
template <class T16, bool SPLIT = false>
DEV void partial_tile(const f32x16& K, const f32x16& V, const RowRange& rr, float m, const GroupCtx& cx, float& ssum, f32x8& keep) {
    f32x16 Ee, Vm;
    float s = 0.f;
    if (__builtin_amdgcn_readfirstlane(rr.span) == 32u) {       // all 32 rows valid: no predicates (they cost 3 instructions per element)
        s = exp_rows(Ee, K, m);
        Vm = V;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = row_ok(rr, r);
            const float e = ok ? exp2f_fast(K[r] - m) : 0.f;
            Ee[r] = e;
            s += e;
            Vm[r] = ok ? V[r] : 0.f;
        }
    }
    ssum = xhalf_sum(s);
    XFrag<T16, SPLIT> ef, vf;
    make_frag<T16, SPLIT>(Ee, ef);
    make_frag<T16, SPLIT>(Vm, vf);
    f32x16 P = splat(0.f);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        P = mfma(ef.hi[s2], vf.hi[s2], P);
        if constexpr (SPLIT) {
            P = mfma(ef.lo[s2], vf.hi[s2], P);
            P = mfma(ef.hi[s2], vf.lo[s2], P);
        }
    }
    keep = keep_head_block(P, cx.c);
}

// after the barrier that follows the last partial_tile: wave w sums tile oc = w & 3 of slot w >> 2 over the waves
// NW = 8: wave w sums tile oc = w & 3 of slot w >> 2; NW = 4 (narrow workgroups): wave w sums tile oc = w of both slots in turn.
template <int NW = 8>
DEV void wg_write_record(float* __restrict__ recs, const float* mx, const f32x8* pst, const f32x8* xp, const float* ss,
                         int wave, int lane, int ub0, int nact /* active waves of the workgroup (a prefix) */, int M, int T, int wg) {
    const int oc = wave & 3, c = lane & 31;
#pragma unroll
    for (int pass = 0; pass < (NW == 8 ? 1 : 2); ++pass) {
        const int sl = NW == 8 ? wave >> 2 : pass;
        f32x8 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        float ssum = 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const int gv = wg * NW + v;                           // (M, T: the unit space of WgMap)
            if (v >= nact) continue;
            const int edge = (ub0 + 1) * T;                       // first token of slot 1's clip
            const int s0v = 32 * gv >= edge ? 1 : 0;              // the wave's primary slot
            const bool strad = !s0v && min(32 * gv + 31, M - 1) >= edge;
            if (s0v == sl) {
                const f32x8 p = pst[(v * 4 + oc) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += p[i];
                ssum += ss[(v * 4 + oc) * 32 + c];
            }
            if (strad && sl == 1) {
                const f32x8 p = xp[oc * 64 + lane];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += p[i];
                ssum += ss[(NW * 4 + oc) * 32 + c];
            }
        }
        float* R = recs + ((size_t)wg * 2 + sl) * DC_REC_FLOATS;
        if (lane < 32) {
            R[32 * oc + c] = wg_colmax<NW>(mx, oc, sl, c);
            R[128 + 32 * oc + c] = ssum;
        }
        reinterpret_cast<f32x8*>(R + 256)[oc * 64 + lane] = acc;
    }
}
// The workgroup's own combine (512 threads): attention operand fragments A[d][l] of clips ub0, ub0+1 from the unit
// records of the previous kernel -> af [2 clips][8 frags][64 lanes] in LDS (the 8 hi fragments k_attn_combine makes).
// SPLIT (clip-aligned units only: the workgroup has ONE clip): af [8 hi frags | 8 lo frags][64 lanes] of clip ub0.
// scratch (LDS): w [2][NU][128] floats, z [2][128] floats.  Summation order is fixed.
template <class T16, bool SPLIT = false>
DEV void wg_combine_attn(const float* __restrict__ recs, v8<T16>* af, float* scratch, int ub0, int B, int M, int T, int tid, int wg,
                         unsigned long long* st = nullptr) {
#define CSTAMP(k) do { if (st && (tid & 63) == 0) st[(tid >> 6) * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    constexpr int NU = 17, PRE = 9;               // units per clip: T <= 4032; the first PRE are loaded before the weights exist
    float* wsc = scratch;
    float* zsc = scratch + 2 * NU * 128;
    const int ci = tid >> 8, t = tid & 255, ln = t & 63, oc = t >> 6;
    const int c = ln & 31, hh = ln >> 5;
    const int b = ub0 + ci;
    const int ub1 = (min((wg + 1) * 256, M) - 1) / T;          // last clip this workgroup touches
    const bool live = b <= ub1;
    const int u_lo = live ? (b * T) / 256 : 0;
    const int u_hi = live ? (min((b + 1) * T, M) - 1) / 256 : -1;
    const int nu = u_hi - u_lo + 1;
    // unit u (tokens 256u..256u+255) intersects the clip; the clip is the unit's slot 0 iff the unit starts inside it
    auto rec_of = [&](int clip, int u) { return recs + ((size_t)u * 2 + ((u * 256 >= clip * T) ? 0 : 1)) * DC_REC_FLOATS; };
    // All loads of the combine are issued up front, branch-free (indices clamped to a valid unit, results predicated), so that
    // one memory round trip covers them: phase A's scalars first (they are needed first), then the K^T V blocks.
    const int ca = (tid >> 7) & 1, f = tid & 127, ba = ub0 + ca;
    const bool la = ba <= ub1;
    const int bav = la ? ba : ub0;                                   // a clip that certainly has units
    const int a_lo = (bav * T) / 256, a_hi = (min((bav + 1) * T, M) - 1) / 256;
    const int na = la ? a_hi - a_lo + 1 : 0;
    float mr[PRE], sr[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const float* R = rec_of(bav, min(a_lo + k, a_hi));
        mr[k] = R[f];
        sr[k] = R[128 + f];
    }
    const int bv = live ? b : ub0;
    const int v_lo = (bv * T) / 256, v_hi = (min((bv + 1) * T, M) - 1) / 256;
    auto blk = [&](const float* R) {           // this thread's 32-byte piece of a record's K^T V blocks
        return reinterpret_cast<const f32x8*>(R + 256)[oc * 64 + ln];
    };
    f32x8 pre[PRE];
    if (live) {                      // wave-uniform (ci is the wave's half of the workgroup): idle halves issue nothing
#pragma unroll
        for (int k = 0; k < PRE; ++k) pre[k] = blk(rec_of(bv, min(v_lo + k, v_hi)));
    }
    CSTAMP(22);
    if (tid < 256) {                              // phase A: per feature f of clip ca: m*, weights, normaliser
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            if (k >= na) sr[k] = 0.f;
        float mstar = -INFINITY;
#pragma unroll
        for (int k = 0; k < PRE; ++k)
        {
            const float cand = max2(mstar, mr[k]);      // (unconditional + a select: an asm statement under `if` becomes a branch)
            mstar = sr[k] > 0.f ? cand : mstar;
        }
        for (int k = PRE; k < na; ++k) {          // clips longer than 9 workgroups (T > 2048)
            const float* R = rec_of(ba, a_lo + k);
            if (R[128 + f] > 0.f) mstar = fmaxf(mstar, R[f]);
        }
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const float ww = sr[k] > 0.f ? exp2f_fast(mr[k] - mstar) : 0.f;
            if (k < na) wsc[(ca * NU + k) * 128 + f] = ww;
            z += ww * sr[k];
        }
        for (int k = PRE; k < na; ++k) {
            const float* R = rec_of(ba, a_lo + k);
            const float su = R[128 + f];
            const float ww = su > 0.f ? exp2f_fast(R[f] - mstar) : 0.f;
            wsc[(ca * NU + k) * 128 + f] = ww;
            z += ww * su;
        }
        zsc[ca * 128 + f] = z;
    }
    CSTAMP(23);
    __syncthreads();
    CSTAMP(24);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int rowb = 32 * oc + 16 * (c >> 4) + 4 * hh;       // kept value j <-> feature row rowb + (j&3) + 8*(j>>2)
    auto wrow = [&](const float* base, float (&w8)[8]) {       // rows rowb..+3 and rowb+8..+11: two 16-byte LDS reads
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + rowb), c2 = *reinterpret_cast<const f32x4*>(base + rowb + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w8[j] = a[j];
            w8[4 + j] = c2[j];
        }
    };
#pragma unroll
    for (int k = 0; k < PRE; ++k)
        if (k < nu) {
            float w8[8];
            wrow(wsc + (ci * NU + k) * 128, w8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w8[j], pre[k][j], acc[j]);
        }
    for (int k = PRE; k < nu; ++k) {
        const f32x8 pv = blk(rec_of(b, u_lo + k));
        float w8[8];
        wrow(wsc + (ci * NU + k) * 128, w8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(w8[j], pv[j], acc[j]);
    }
    CSTAMP(25);
    v8<T16> out, outl, zero;
    {
        float z8[8];
        wrow(zsc + ci * 128, z8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (split: an exact division - the reciprocal approximation's 1 ulp would be the largest error of the mode)
            const float a = z8[j] > 0.f ? (SPLIT ? acc[j] / z8[j] : acc[j] * fast_rcp(z8[j])) : 0.f;
            out[j] = (T16)a;
            outl[j] = (T16)(a - (float)out[j]);
            zero[j] = (T16)0.f;
        }
    }
    const int s = c >> 4;
    if constexpr (SPLIT) {
        if (ci == 0) {
            af[(oc * 2 + s) * 64 + ln] = out;
            af[(oc * 2 + (s ^ 1)) * 64 + ln] = zero;
            af[(8 + oc * 2 + s) * 64 + ln] = outl;
            af[(8 + oc * 2 + (s ^ 1)) * 64 + ln] = zero;
        }
    } else {
        af[(ci * 8 + oc * 2 + s) * 64 + ln] = out;
        af[(ci * 8 + oc * 2 + (s ^ 1)) * 64 + ln] = zero;
    }
}


// The same combine for NARROW workgroups (4 waves = 128-token units, 256 threads; small batches, one wave per SIMD).
// scratch (LDS): w [2][32][128] floats, z [2][128] floats: <= 32 units per clip (T <= 3840).  The K^T V blocks of the
// workgroup's first clip are loaded up front (one memory round trip together with phase A's scalars); a second clip
// (the one unit per clip that contains a clip edge) is loaded in batches behind the barrier.
template <class T16>
DEV void wg_combine_attn_narrow(const float* __restrict__ recs, v8<T16>* af, float* scratch, int ub0, int M, int T, int tid, int wg) {
    constexpr int UT = 128, NU = 32, PRE = 16;
    float* wsc = scratch;
    float* zsc = scratch + 2 * NU * 128;
    const int ub1 = (min((wg + 1) * UT, M) - 1) / T;          // last clip this unit touches
    auto rec_of = [&](int clip, int u) { return recs + ((size_t)u * 2 + ((u * UT >= clip * T) ? 0 : 1)) * DC_REC_FLOATS; };
    const int ca = tid >> 7, f = tid & 127, ba = ub0 + ca;
    const bool la = ba <= ub1;
    const int bav = la ? ba : ub0;
    const int a_lo = (bav * T) / UT, a_hi = (min((bav + 1) * T, M) - 1) / UT;
    const int na = la ? a_hi - a_lo + 1 : 0;
    float mr[PRE], sr[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {                            // branch-free: indices clamped, results predicated
        const float* R = rec_of(bav, min(a_lo + k, a_hi));
        mr[k] = R[f];
        sr[k] = R[128 + f];
    }
    const int oc = tid >> 6, ln = tid & 63, c = ln & 31, hh = ln >> 5;
    const int v_lo = (ub0 * T) / UT, v_hi = (min((ub0 + 1) * T, M) - 1) / UT;
    f32x8 pre[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) pre[k] = reinterpret_cast<const f32x8*>(rec_of(ub0, min(v_lo + k, v_hi)) + 256)[oc * 64 + ln];
    {   // phase A: per feature f of clip ca: m*, weights, normaliser
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            if (k >= na) sr[k] = 0.f;
        float mstar = -INFINITY;
#pragma unroll
        for (int k = 0; k < PRE; ++k)
        {
            const float cand = max2(mstar, mr[k]);      // (unconditional + a select: an asm statement under `if` becomes a branch)
            mstar = sr[k] > 0.f ? cand : mstar;
        }
        for (int k = PRE; k < na; ++k) {                       // clips longer than 16 units (T > 1920)
            const float* R = rec_of(ba, a_lo + k);
            if (R[128 + f] > 0.f) mstar = fmaxf(mstar, R[f]);
        }
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const float ww = sr[k] > 0.f ? exp2f_fast(mr[k] - mstar) : 0.f;
            if (k < na) wsc[(ca * NU + k) * 128 + f] = ww;
            z += ww * sr[k];
        }
        for (int k = PRE; k < na; ++k) {
            const float* R = rec_of(ba, a_lo + k);
            const float su = R[128 + f];
            const float ww = su > 0.f ? exp2f_fast(R[f] - mstar) : 0.f;
            wsc[(ca * NU + k) * 128 + f] = ww;
            z += ww * su;
        }
        zsc[ca * 128 + f] = z;
    }
    __syncthreads();
    const int rowb = 32 * oc + 16 * (c >> 4) + 4 * hh;        // kept value j <-> feature row rowb + (j&3) + 8*(j>>2)
    auto wrow = [&](const float* base, float (&w8)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + rowb), c2 = *reinterpret_cast<const f32x4*>(base + rowb + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w8[j] = a[j];
            w8[4 + j] = c2[j];
        }
    };
    auto emit = [&](int ci, const float (&acc)[8], bool live) {
        v8<T16> out, zero;
        float z8[8];
        wrow(zsc + ci * 128, z8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            out[j] = (T16)((live && z8[j] > 0.f) ? acc[j] * fast_rcp(z8[j]) : 0.f);
            zero[j] = (T16)0.f;
        }
        const int s = c >> 4;
        af[(ci * 8 + oc * 2 + s) * 64 + ln] = out;
        af[(ci * 8 + oc * 2 + (s ^ 1)) * 64 + ln] = zero;
    };
    {   // clip ub0 (always live)
        const int nu = v_hi - v_lo + 1;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            if (k < nu) {
                float w8[8];
                wrow(wsc + k * 128, w8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(w8[j], pre[k][j], acc[j]);
            }
        for (int k = PRE; k < nu; ++k) {
            const f32x8 pv = reinterpret_cast<const f32x8*>(rec_of(ub0, v_lo + k) + 256)[oc * 64 + ln];
            float w8[8];
            wrow(wsc + k * 128, w8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w8[j], pv[j], acc[j]);
        }
        emit(0, acc, true);
    }
    {   // clip ub0 + 1: only the unit that contains a clip edge
        const bool live = ub0 + 1 <= ub1;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (live) {
            const int b = ub0 + 1;
            const int u_lo = (b * T) / UT, u_hi = (min((b + 1) * T, M) - 1) / UT, nu = u_hi - u_lo + 1;
            for (int k0 = 0; k0 < nu; k0 += 8) {
                f32x8 pb[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pb[k] = reinterpret_cast<const f32x8*>(rec_of(b, min(u_lo + k0 + k, u_hi)) + 256)[oc * 64 + ln];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k0 + k < nu) {
                        float w8[8];
                        wrow(wsc + (NU + k0 + k) * 128, w8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(w8[j], pb[k][j], acc[j]);
                    }
            }
        }
        emit(1, acc, live);
    }
}

}  // namespace dc
using namespace dc;

// LDS-DMA of one 1-KiB fragment: lane i's 16 bytes at gsrc land at lds_dst + 16*i.  Issued through inline asm
// on purpose: for the builtin form hipcc inserts `s_waitcnt vmcnt(0)` before the next LDS read of ANY address
// (it assumes the DMA may alias), which would turn every "one stage ahead" prefetch into a synchronous copy.
// All waits for these copies are explicit (stage_sync, the FiLM ring).  The statements overwrite M0 (the DMA's LDS base) and do NOT
// restore it: hipcc treats M0 as reserved (a clobber is refused with a warning) and itself touches it only for indirect register
// indexing, LDS-direct and message instructions, none of which these translation units contain -
// tests/test_host_logic.py::test_production_layer_kernel_has_no_register_spills checks the ISA for it.
// Round 5 (instruction diet, profiles/r05_census.md): the LDS address is the low half of the generic pointer (a generic LDS pointer is
// aperture << 32 | offset; the addrspace cast the old form used costs a null check of 5 scalar instructions per copy), M0 is no longer
// saved and restored around every copy, and the *_s forms take a wave-uniform base in SGPRs plus ONE per-lane byte offset, so that a
// stage image costs scalar adds instead of a 64-bit vector add per fragment.
DEV unsigned lds_addr(const char* p) { return (unsigned)(size_t)p; }
DEV void lds_dma16(const void* gsrc /*per-lane*/, const char* lds_dst /*wave-uniform*/) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(lds_dst));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(dst) : "memory");
}
template <int OFF = 0, bool NT = false>
DEV void lds_dma16_s(const void* sbase /*wave-uniform*/, unsigned voff /*per-lane byte offset*/, unsigned lds_dst /*wave-uniform LDS address*/) {
    // OFF: the instruction's immediate offset - it is added to the global address AND to the LDS address
    static_assert(OFF >= 0 && OFF < 4096, "immediate offset field");
    if constexpr (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3 nt" : : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(OFF) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(OFF) : "memory");
}

template <int NW>
DEV void stage_frags(const void* __restrict__ src /*wave-uniform*/, char* dst, int nfrags, int wave /*wave-uniform*/, int lane) {
    const char* s = reinterpret_cast<const char*>(src) + (size_t)wave * 1024;
    unsigned d = __builtin_amdgcn_readfirstlane(lds_addr(dst)) + wave * 1024;
    const unsigned voff = lane * 16;
    for (int f = wave; f < nfrags; f += NW) {
        lds_dma16_s(s, voff, d);
        s += NW * 1024;
        d += NW * 1024;
    }
}
// Progress priority: a wave lowers its issue priority as it advances through a stage (s_setprio 3 at the stage's head ... 0 near its
// closer), so that of the two waves of a SIMD the one that is BEHIND wins arbitration and both reach the closer together - under the
// hardware's oldest-first rule the older wave runs ahead and then idles at every closer while its partner finishes alone at
// single-wave efficiency.  Same-box A/B (profiles/r04_ab_stage_prio.txt): k_layer -1.2 ... -1.9 % per launch, loop -0.7 % on average
// over three boxes (-1.3 / +-0 / -0.9); the inverse mapping (control) +0.3 %.  -DDC_STAGE_PRIO=0 builds without it, =1 without the
// record tail's checkpoints.
#ifndef DC_STAGE_PRIO
#define DC_STAGE_PRIO 2
#endif
template <int P, int LEVEL = 1>
DEV void sprio() {
#if DC_STAGE_PRIO
    if constexpr (DC_STAGE_PRIO >= LEVEL) {
#ifdef DC_STAGE_PRIO_INV      // control: the wave that is AHEAD wins
        __builtin_amdgcn_s_setprio(3 - P);
#else
        __builtin_amdgcn_s_setprio(P);
#endif
    }
#endif
}
DEV void stage_sync() {
    __builtin_amdgcn_sched_barrier(0);           // stages do not interleave: keeps each stage's live set separate
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef DC_DIAG_NO_STAGE_BARRIER                 // diagnostic build (timing only, results invalid): the waves of a workgroup free-run
    __syncthreads();                             // through the stage closers - the bound on what any point-to-point hand-off can gain
#endif
    sprio<3>();
    __builtin_amdgcn_sched_barrier(0);
}

template <class T16, bool SPLIT>
DEV void attn_apply_tile(f32x16& y, const v8<T16>* __restrict__ afrag, int oc, const XFrag<T16, SPLIT>& q, int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const v8<T16> a = afrag[(oc * 2 + s) * 64 + lane];
        y = mfma(a, q.hi[s], y);
        if constexpr (SPLIT) {
            y = mfma(a, q.lo[s], y);
            const v8<T16> al = afrag[(8 + oc * 2 + s) * 64 + lane];
            y = mfma(al, q.hi[s], y);
        }
    }
}

// The attention / FFN output y that feeds a StylizationBlock: its LayerNorm statistics are taken from the exact
// fp32 accumulators as the tiles are produced; the tiles themselves are then held as packed f16 in the
// non-split modes (32 instead of 64 VGPRs - y only ever passes through LayerNorm -> FiLM -> SiLU -> f16 operand).
template <bool SPLIT> struct YT { using tile = f16x16; };
template <> struct YT<true> { using tile = f32x16; };
template <bool SPLIT> using ytile = typename YT<SPLIT>::tile;

struct RowStats {
    f32x2 s = {0.f, 0.f}, q = {0.f, 0.f};          // packed-fp32 partial sums (even / odd registers)
    DEV void add(const f32x16& x) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x2 v = {x[2 * r], x[2 * r + 1]};
            s += v;
            q = fma2(v, v, q);
        }
    }
    // LayerNorm(128) of a StylizationBlock input, in the log2(e) scaling of styl_tile: log2(e) nhat = x*rstd + shift
    DEV void finish(float& rstd, float& shift) {
        const float ss = xhalf_sum(s.x + s.y), qq = xhalf_sum(q.x + q.y);
        const float mean = ss * (1.f / 128.f);
        const float var = fmaxf(fmaf(-mean, mean, qq * (1.f / 128.f)), 0.f);
        rstd = __builtin_amdgcn_rsqf(var + 1e-5f) * 1.4426950408889634f;
        shift = -mean * rstd;
    }
};
template <bool SPLIT>
DEV void put_y(ytile<SPLIT>& dst, const f32x16& x) {
    if constexpr (SPLIT) {
        dst = x;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = (_Float16)x[r];
    }
}

// q = softmax_heads(Wq LN(h) + bq);  y = q . A per head  (weights image `w` in LDS, bias block behind it)
template <class T16, bool SPLIT>
DEV void query_attend(ytile<SPLIT> (&y)[4], float& y_rstd, float& y_shift, const f32x16 (&h)[4], const float* bq,
                      const v8<T16>* w, const v8<T16>* a0, const v8<T16>* a1, const GroupCtx& cx) {
    f32x16 q[4];
    {
        XFrag<T16, SPLIT> nf[4];
        ln_frags<T16, SPLIT>(nf, h);
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = ld_ft(bq, t, cx.hh);
        gemm_wa<4, 4, T16, SPLIT>(q, w, nf, cx.lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    sprio<2>();
    softmax_heads_ft(q);
    __builtin_amdgcn_sched_barrier(0);
    sprio<1>();
    RowStats st;
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
        if (oc == 2) sprio<0>();
        f32x16 acc = splat(0.f);
        XFrag<T16, SPLIT> qf;
        make_frag<T16, SPLIT>(q[oc], qf);
        if (!cx.straddle) {
            attn_apply_tile<T16, SPLIT>(acc, a0, oc, qf, cx.lane);
        } else {   // the group spans two clips: apply each clip's matrix to its own tokens (lanes)
            XFrag<T16, SPLIT> qm = qf;
            mask_frag<T16, SPLIT>(qm, cx.lane_in_b0);
            attn_apply_tile<T16, SPLIT>(acc, a0, oc, qm, cx.lane);
            mask_frag<T16, SPLIT>(qf, !cx.lane_in_b0);
            attn_apply_tile<T16, SPLIT>(acc, a1, oc, qf, cx.lane);
        }
        st.add(acc);
        put_y<SPLIT>(y[oc], acc);
    }
    st.finish(y_rstd, y_shift);
}

// v_fma_mix_f32: d = a * b + c with a read as the low / high fp16 half of a packed register - no separate conversion.
// (hipcc does not form it from `fmaf((float)half, ...)`: the FiLM tiles and the packed y tiles cost 3 cvt per element.)
template <int HI>
DEV float fma_mix_h(uint32_t h2, float b, float c) {
    float d;
    if constexpr (HI)
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(b), "v"(c));
    else
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(b), "v"(c));
    return d;
}
template <int HI>
DEV float add_mix_h(uint32_t h2, float c) {        // (float)half + c
    float d;
    if constexpr (HI)
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(c));
    else
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(c));
    return d;
}

// d = ga * n + hb with ga and hb read as fp16 halves of two packed registers
template <int HI>
DEV float fma_mix_hh(uint32_t g2, float n, uint32_t h2) {
    float d;
    if constexpr (HI)
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(g2), "v"(n), "v"(h2));
    else
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(g2), "v"(n), "v"(h2));
    return d;
}
// n-hat G' + H' from a tile pair's packed halves.  G1 (plain-operand kernels of a loop with a precise tail): the scale tile holds G' itself - one FMA; else it
// holds G' - 1 (fp16's 11 bits on the small part): (G' - 1) n + n, then + H'.  The FiLM GEMM is launched with the matching constants
// (dc_api.hip, enqueue_step).
template <int HI, bool G1>
DEV float film_affine(uint32_t g2, float n, uint32_t h2) {
#ifdef DC_NO_FILM_G1          // (A/B build: G' - 1 tiles everywhere, round 4's form)
    constexpr bool g1 = false;
#else
    constexpr bool g1 = G1;
#endif
    if constexpr (g1)
        return fma_mix_hh<HI>(g2, n, h2);
    else
        return add_mix_h<HI>(h2, fma_mix_h<HI>(g2, n, n));
}

// ---- packed-fp16 form of the stylization's elementwise chain (fp16 precision, plain-operand evaluations of a loop with a precise
// tail: G1).  The chain's result is rounded to fp16 for the MFMA anyway; computing it in fp16 from the affine on saves one vector
// instruction per element (DC_STYL_PK16 = 1: n-hat through v_fma_mix{lo,hi}_f16 - fp32 arithmetic on the fp32 statistics, ONE rounding -,
// then v_pk_fma_f16 / v_pk_add_f16 / v_pk_mul_f16 with v_exp_f16 / v_rcp_f16 per half: 4.5 instead of 5.5 per element, -192 of
// 4 537 per wave and layer) or 1.5 (DC_STYL_PK16 = 2: n-hat as v_pk_fma_f16 on fp16 copies of rstd / shift as well: 4.0).  hipcc
// extracts the high half of a pair through SDWA and re-packs with v_pack_b32_f16; the SDWA forms below write the high half in place
// (tools/probe_pk16.hip checks them on the box).
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// The high half is written in place by a second SDWA instruction (dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE).  Two gfx940+ hazards apply
// that the compiler's hazard recognizer cannot see inside asm statements: a transcendental's result, and any result written through a
// destination select (SDWA dst_sel, the high-half write of v_fma_mixhi_f16), is not forwarded to an instruction in the next slot - back
// to back, a quarter of the values came out wrong on the box (tools/probe_pk16.hip).  So a tile's eight low halves run first, the eight
// high halves after them, and a phase that ends in such writes closes with s_nop 1 before the compiler's own instructions consume them
// (whatever order the scheduler then picks); scheduling barriers keep the phases apart.
DEV uint32_t exp2neg_lo16(uint32_t u) {                       // {2^-u.lo, 0}
    uint32_t e;
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(e) : "v"(u));
    return e;
}
DEV void exp2neg_hi16(uint32_t& e, uint32_t u) {             // e.hi = 2^-u.hi, e.lo kept
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(e) : "v"(u));
}
DEV uint32_t rcp_lo16(uint32_t d) {
    uint32_t r;
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(d));
    return r;
}
DEV void rcp_hi16(uint32_t& r, uint32_t d) {
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r) : "v"(d));
}
DEV uint32_t nhat_lo16(uint32_t y2, float rstd, float shift) {       // {fp16(y.lo * rstd + shift), -}: fp32 arithmetic, one rounding
    uint32_t n;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(n) : "v"(y2), "v"(rstd), "v"(shift));
    return n;
}
DEV void nhat_hi16(uint32_t& n, uint32_t y2, float rstd, float shift) {
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(n) : "v"(y2), "v"(rstd), "v"(shift));
}
DEV void phase_end16() {
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}
// u / (1 + 2^-u) on the eight packed pairs of a tile (in place)
DEV void silu_l2_tile16(uint32_t (&u)[8]) {
    uint32_t e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = exp2neg_lo16(u[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) exp2neg_hi16(e[k], u[k]);
    phase_end16();
    const h16x2 one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = __builtin_bit_cast(uint32_t, (h16x2)(__builtin_bit_cast(h16x2, e[k]) + one));
    __builtin_amdgcn_sched_barrier(0);
    uint32_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = rcp_lo16(e[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) rcp_hi16(r[k], e[k]);
    phase_end16();
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = __builtin_bit_cast(uint32_t, (h16x2)(__builtin_bit_cast(h16x2, u[k]) * __builtin_bit_cast(h16x2, r[k])));
}

// one k-tile of the FiLM-modulated, SiLU'ed operand: z = SiLU(nhat*G' + H'), gp = G' - 1 (split operands) or G'; everything in the log2(e)
// scaling of silu_l2_pair: rstd / shift arrive multiplied by log2(e), hp = log2(e) H', z = log2(e) SiLU(.)
template <class T16, bool SPLIT, class YTile, bool G1 = false>
DEV void styl_tile(XFrag<T16, SPLIT>& zf, const YTile& y, float rstd, float shift, const f16x16& gp, const f16x16& hp) {
#if defined(DC_STYL_PK16) && DC_STYL_PK16
    if constexpr (std::is_same<YTile, f16x16>::value && std::is_same<T16, _Float16>::value && !SPLIT && G1) {
        const u32x8 yw = __builtin_bit_cast(u32x8, y), gw = __builtin_bit_cast(u32x8, gp), hw = __builtin_bit_cast(u32x8, hp);
        // (every vector element goes through a scalar copy first: __builtin_bit_cast of an ext-vector ELEMENT expression reads the vector's
        // first element whatever the index - hipcc 7.2 - which cost round 6 two GPU sessions)
        uint32_t zw[8];
#if DC_STYL_PK16 == 2
        const h16x2 rs = {(_Float16)rstd, (_Float16)rstd}, sh = {(_Float16)shift, (_Float16)shift};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t yk = yw[k];
            zw[k] = __builtin_bit_cast(uint32_t, (h16x2)__builtin_elementwise_fma(__builtin_bit_cast(h16x2, yk), rs, sh));
        }
#else
#pragma unroll
        for (int k = 0; k < 8; ++k) zw[k] = nhat_lo16(yw[k], rstd, shift);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) nhat_hi16(zw[k], yw[k], rstd, shift);
        phase_end16();
#endif
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t gk = gw[k], hk = hw[k], nk = zw[k];
            zw[k] = __builtin_bit_cast(uint32_t, (h16x2)__builtin_elementwise_fma(__builtin_bit_cast(h16x2, gk), __builtin_bit_cast(h16x2, nk),
                                                                                  __builtin_bit_cast(h16x2, hk)));
        }
        __builtin_amdgcn_sched_barrier(0);
        silu_l2_tile16(zw);
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
        zf.hi[0] = __builtin_bit_cast(v8<T16>, (u32x4_){zw[0], zw[1], zw[2], zw[3]});
        zf.hi[1] = __builtin_bit_cast(v8<T16>, (u32x4_){zw[4], zw[5], zw[6], zw[7]});
        return;
    }
#endif
    f32x16 z;
    if constexpr (std::is_same<YTile, f16x16>::value) {      // packed y (non-split formats): three mixed-precision FMAs per element
        const u32x8 yw = __builtin_bit_cast(u32x8, y), gw = __builtin_bit_cast(u32x8, gp), hw = __builtin_bit_cast(u32x8, hp);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float n0 = fma_mix_h<0>(yw[k], rstd, shift), n1 = fma_mix_h<1>(yw[k], rstd, shift);
            const f32x2 zz = silu_l2_pair(film_affine<0, G1 && !SPLIT>(gw[k], n0, hw[k]), film_affine<1, G1 && !SPLIT>(gw[k], n1, hw[k]));
            z[2 * k] = zz.x;
            z[2 * k + 1] = zz.y;
        }
    } else {             // fp32 y (split formats): the FiLM tiles still enter through mixed-precision FMAs (no v_cvt_f32_f16)
        const u32x8 gw = __builtin_bit_cast(u32x8, gp), hw = __builtin_bit_cast(u32x8, hp);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float n0 = fmaf((float)y[2 * k], rstd, shift), n1 = fmaf((float)y[2 * k + 1], rstd, shift);
            const f32x2 zz = silu_l2_pair(film_affine<0, G1 && !SPLIT>(gw[k], n0, hw[k]), film_affine<1, G1 && !SPLIT>(gw[k], n1, hw[k]));
            z[2 * k] = zz.x;
            z[2 * k + 1] = zz.y;
        }
    }
    make_frag<T16, SPLIT>(z, zf);
}

// StylizationBlock (transformer.py:68-81) accumulated straight into the residual stream:
//   h += W_o * SiLU( LN(y) * (1 + scale) + shift ) + b_o          (weights image `w` in LDS, b_o behind it)
// with LN(y)*(1+scale)+shift = nhat*G' + H', nhat = (y-mean)*rstd; the FiLM GEMM delivers G'-1 and H' tiles.
// E tiles come straight from global memory (Eg: this block's 8 tiles for this group).
template <class T16, bool SPLIT, bool G1 = false>
DEV void styl_accumulate(f32x16 (&h)[4], const ytile<SPLIT> (&y)[4], float rstd, float shift, const f16x8* __restrict__ Eg,
                         const float* bo, const v8<T16>* w, int lane, int hh) {
    XFrag<T16, SPLIT> zf[4];
    {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f16x16 gp, hp;
            gp = load_etile(Eg + kt * 128 + lane);
            hp = load_etile(Eg + (4 + kt) * 128 + lane);
            styl_tile<T16, SPLIT, ytile<SPLIT>, G1>(zf[kt], y[kt], rstd, shift, gp, hp);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x16 bb = ld_ft(bo, t, hh);
#pragma unroll
        for (int r = 0; r < 8; ++r) {                                  // v_pk_add_f32
            const f32x2 v = (f32x2){h[t][2 * r], h[t][2 * r + 1]} + (f32x2){bb[2 * r], bb[2 * r + 1]};
            h[t][2 * r] = v.x;
            h[t][2 * r + 1] = v.y;
        }
    }
    gemm_wa<4, 4, T16, SPLIT>(h, w, zf, lane);
    __builtin_amdgcn_sched_barrier(0);
}

DEV f16x16 join16(const f16x8& lo8, const f16x8& hi8) {
    f16x16 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = lo8[i];
        v[8 + i] = hi8[i];
    }
    return v;
}
// The same block for the split formats (no LDS left for FiLM rings: two 65-KiB weight images).  The FiLM tiles travel through
// registers two k-tiles ahead: those of k-tiles 0 and 1 are requested at the END of the preceding stage (epf_fetch in front of a
// closer that leaves exactly these 8 loads in flight, stage_sync_keep8) and land behind the barrier and the next image's DMA
// issue; those of k-tile kt + 2 as soon as those of k-tile kt have been consumed; the out-projection's MFMAs of a k-tile follow
// its SiLU directly - two tile pairs (32 registers) in flight instead of all operand fragments of the block (64) plus loads at
// the point of use.  Same products in the same order per accumulator as styl_accumulate (k-tile-major): identical results.
struct EPf {
    f16x8 g0[2], g1[2], h0[2], h1[2];            // [slot]: the 16-byte halves of the (G' - 1, H') tiles of k-tiles kt, kt + 1
};
DEV void epf_fetch(EPf& e, const f16x8* __restrict__ Eg, int kt, int slot, int lane) {
    const f16x8* pg = Eg + kt * 128 + lane;
    const f16x8* ph = Eg + (4 + kt) * 128 + lane;
    e.g0[slot] = __builtin_nontemporal_load(pg);
    e.g1[slot] = __builtin_nontemporal_load(pg + 64);
    e.h0[slot] = __builtin_nontemporal_load(ph);
    e.h1[slot] = __builtin_nontemporal_load(ph + 64);
}
// stage closer that leaves the wave's 8 youngest vector-memory operations (the FiLM tiles just requested) in flight: vmcnt counts
// in issue order, so everything older - this wave's share of the next weight image - has landed
DEV void stage_sync_keep8() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#ifndef DC_DIAG_NO_STAGE_BARRIER
    __syncthreads();
#endif
    sprio<3>();
    __builtin_amdgcn_sched_barrier(0);
}
template <class T16, bool SPLIT, bool PRE /* k-tiles 0, 1 were requested by the preceding stage */>
DEV void styl_accumulate_pf(f32x16 (&h)[4], const ytile<SPLIT> (&y)[4], float rstd, float shift, const f16x8* __restrict__ Eg, EPf& e,
                            const float* bo, const v8<T16>* w, int lane, int hh) {
    if constexpr (!PRE) {
        epf_fetch(e, Eg, 0, 0, lane);
        epf_fetch(e, Eg, 1, 1, lane);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x16 bb = ld_ft(bo, t, hh);
#pragma unroll
        for (int r = 0; r < 8; ++r) {                                  // v_pk_add_f32
            const f32x2 v = (f32x2){h[t][2 * r], h[t][2 * r + 1]} + (f32x2){bb[2 * r], bb[2 * r + 1]};
            h[t][2 * r] = v.x;
            h[t][2 * r + 1] = v.y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        XFrag<T16, SPLIT> zf;
        styl_tile<T16, SPLIT, ytile<SPLIT>>(zf, y[kt], rstd, shift, join16(e.g0[kt & 1], e.g1[kt & 1]), join16(e.h0[kt & 1], e.h1[kt & 1]));
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < 4) epf_fetch(e, Eg, kt + 2, kt & 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_kt<4, 4, T16, SPLIT>(h, w, kt, zf, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (kt == 0) sprio<2>();
        if (kt == 1) sprio<1>();
        if (kt == 2) sprio<0>();
    }
}

// Per-wave FiLM tile ring in LDS: two 4-KiB slots, each holding one k-tile's (G'-1, H') tile pair.
// FiLM tiles are read exactly once: non-temporal, so that 88 MB per layer do not flush the weights, attention fragments and
// records the workgroups of an XCD share through L2 (same-box A/B: -4 % k_layer, -3.5 % loop)
DEV void ering_issue(const f16x8* __restrict__ Eg /*wave-uniform: the block's 8 tiles for this wave's group*/, int kt, char* slot, int lane) {
#ifdef DC_DIAG_NO_ELOAD
    return;      // diagnostic build (timing only, results invalid): what the FiLM tile reads cost the layer kernel
#endif
    const unsigned voff = lane * 16, d = __builtin_amdgcn_readfirstlane(lds_addr(slot));
    const f16x8* gsrc = Eg + kt * 128;
    const f16x8* hsrc = Eg + (4 + kt) * 128;
#ifdef DC_E_PLAIN
    constexpr bool nt = false;
#else
    constexpr bool nt = true;
#endif
    lds_dma16_s<0, nt>(gsrc, voff, d);            // (the immediate offset moves the global AND the LDS address: the second half of a
    lds_dma16_s<1024, nt>(gsrc, voff, d);         // tile lands 1 KiB behind the first without a second LDS base)
    lds_dma16_s<0, nt>(hsrc, voff, d + 2048);
    lds_dma16_s<1024, nt>(hsrc, voff, d + 2048);
}
// same StylizationBlock with the FiLM tiles of k-tiles 0,1 arriving through the ring (issued a stage ago) and those of
// k-tiles 2,3 prefetched into registers at the start of the preceding stage (EPre; they landed with that stage's closing
// vmcnt(0)): nothing in this stage waits on HBM.  `prefetch_next` (the next stage's weight image and the next block's
// ring tiles) is issued as soon as both ring slots have been read.
struct EPre {
    f16x8 glo[2], ghi[2], hlo[2], hhi[2];       // the 16-byte halves exactly as loaded
};
// Compiler-tracked non-temporal loads (the tiles are read exactly once).  Until round 4 the production kernel issued them through
// inline asm the compiler could not see (no wait at the use: they are complete at the preceding stage's closing vmcnt(0)) - sound
// only while the register allocator never copied or spilled the targets; the tracked form measured equal on the same box
// (profiles/r04_ab_epre_tracked.txt) and replaced it.
DEV void epre_load(EPre& e, const f16x8* __restrict__ Eg, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f16x8* pg = Eg + (2 + i) * 128 + lane;
        const f16x8* ph = Eg + (4 + 2 + i) * 128 + lane;
#if defined(DC_DIAG_NO_ELOAD)
        // (values that are not compile-time constants, so that the stylization arithmetic stays)
        const f16x8 z = __builtin_bit_cast(f16x8, (f32x4){(float)lane, (float)i, 1.f, 2.f});
        e.glo[i] = e.ghi[i] = e.hlo[i] = e.hhi[i] = z;
        (void)pg; (void)ph;
#elif defined(DC_E_PLAIN)
        e.glo[i] = pg[0];
        e.ghi[i] = pg[64];
        e.hlo[i] = ph[0];
        e.hhi[i] = ph[64];
#else
        e.glo[i] = __builtin_nontemporal_load(pg);
        e.ghi[i] = __builtin_nontemporal_load(pg + 64);
        e.hlo[i] = __builtin_nontemporal_load(ph);
        e.hhi[i] = __builtin_nontemporal_load(ph + 64);
#endif
    }
}
// The tracked prefetch without its price: the compiler guards the FIRST use of a load it knows about with a vmcnt wait, and
// placed in the middle of the consuming stage that wait would also cover the LDS-DMAs issued just before it.  A use at the very
// start of the consuming stage - behind the previous stage's closing vmcnt(0), before any new vector-memory operation - puts the
// compiler's wait where the queue is empty.
DEV void epre_landed(EPre& e) {
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(e.glo[i]), "+v"(e.ghi[i]), "+v"(e.hlo[i]), "+v"(e.hhi[i]));
}
template <class T16, bool SPLIT, bool G1 = false, class F>
DEV void styl_accumulate_ring(f32x16 (&h)[4], const ytile<SPLIT> (&y)[4], float rstd, float shift, const EPre& ep,
                              char* ring, const float* bo, const v8<T16>* w, int lane, int hh, F&& prefetch_next) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x16 bb = ld_ft(bo, t, hh);
#pragma unroll
        for (int r = 0; r < 8; ++r) {                                  // v_pk_add_f32
            const f32x2 v = (f32x2){h[t][2 * r], h[t][2 * r + 1]} + (f32x2){bb[2 * r], bb[2 * r + 1]};
            h[t][2 * r] = v.x;
            h[t][2 * r + 1] = v.y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        f16x16 gp, hp;
        if (kt < 2) {
            const f16x8* sp = reinterpret_cast<const f16x8*>(ring + kt * 4096) + lane;
            gp = load_etile(sp);
            hp = load_etile(sp + 128);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kt == 1) prefetch_next();                                   // both slots have been read
            __builtin_amdgcn_sched_barrier(0);
        } else {
            gp = join16(ep.glo[kt - 2], ep.ghi[kt - 2]);
            hp = join16(ep.hlo[kt - 2], ep.hhi[kt - 2]);
        }
        XFrag<T16, SPLIT> zf;
        styl_tile<T16, SPLIT, ytile<SPLIT>, G1>(zf, y[kt], rstd, shift, gp, hp);
        mma_kt<4, 4, T16, SPLIT>(h, w, kt, zf, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (kt == 0) sprio<2>();
        if (kt == 1) sprio<1>();
        if (kt == 2) sprio<0>();
    }
}

// ------------------------------------------------------------------------------------
// The DDIM update of one element (gaussian_diffusion.py:503-521 p_mean_variance's pred_xstart, :812-830 ddim_sample), fp32 with
// the reference's own fp32 scalars c[0..4] (DC_COEF):  mo = the denoiser's output, xt = x_t, z = this iteration's noise draw.
//   pred = mo  |  sqrt(1/abar) x_t - sqrt(1/abar - 1) mo  (EPSILON);   clamp(-1, 1) when clip_denoised;
//   eps  = (sqrt(1/abar) x_t - pred) / sqrt(1/abar - 1);   x_{t-1} = sqrt(abar_prev) pred + sqrt(1 - abar_prev - sigma^2) eps + sigma z
// (sigma is 0 at t = 0 - abar_prev = 1 there - which is the reference's nonzero_mask.)  Returns x_{t-1}; `bad` collects non-finite pred.
// ------------------------------------------------------------------------------------
DEV float ddim_update(float mo, float xt, const float* __restrict__ c, int flags, bool noisy, float z, bool& bad) {
    const float sr = c[0], srm1 = c[1], cx0 = c[2], ceps = c[3];
    float pred = mo;
    if (flags & DC_UPD_EPS) pred = sr * xt - srm1 * mo;
    bad = bad || !(fabsf(pred) <= 3.0e38f);
    if (flags & DC_UPD_CLIP) pred = fminf(fmaxf(pred, -1.f), 1.f);
    const float eps = (sr * xt - pred) / srm1;
    float xn = pred * cx0 + ceps * eps;
    if (noisy) xn += c[4] * z;
    return xn;
}

