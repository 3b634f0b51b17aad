// dc_layer16.hip - the decoder layer with SIXTEEN tokens per wave on v_mfma_f32_16x16x32 (non-split formats, workgroup
// records, production build only).
//
// Why: k_layer (32 tokens per wave on 32x32x16) needs 238 VGPRs, i.e. two waves per SIMD, and at two waves per SIMD only
// ~65 % of the issue slots are used.  With 16 tokens per wave every per-token quantity takes half the
// registers: 16 waves per workgroup (the same 256-token unit), FOUR waves per SIMD.
//
// Layout.  Accumulator tile of v_mfma_f32_16x16x32: lane l holds column n = l & 15 (the TOKEN) and rows 4 (l >> 4) + i,
// i < 4, of a 16-row block.  An activation of 128 features is 8 such blocks: x[rb][i] = feature 16 rb + 4 q4 + i, q4 = l >> 4.
// Two consecutive blocks convert in registers into the B operand of a 32-deep k-step: element j of lane (n, q4) is
//   feature 32 m + 16 (j >> 2) + 4 q4 + (j & 3)            ("chained" k order; the host packs the weights to match),
// and the same registers are the A operand of the transposed products (X^T W) of the record tail.
// Everything in HBM keeps the layouts of the 32-token kernels (residual stream, FiLM tiles, unit records, attention
// fragments of the cross-attention pre-pass), addressed at 16- or 8-byte granularity - so k_embed_front, the FiLM GEMM and
// the conditioning pre-pass are shared, and the two layer kernels are interchangeable launch by launch.
#include "dc_dev.h"
#include "dc_launch.h"

namespace {

constexpr int L16_WSZ = 33 * 1024;
constexpr int L16_OFF_AF = 2 * L16_WSZ;              // attention fragments: [2 clips][8 heads][64 lanes] (16 KiB); tail: column maxima
constexpr int L16_OFF_ER = L16_OFF_AF + 16384;       // per-wave FiLM tile rings: 16 x 4 KiB; tail: K^T V staging
constexpr int L16_OFF_SS = L16_OFF_ER + 16 * 4096;   // tail: column sums
constexpr int L16_LDS = L16_OFF_SS + 17 * 512;

struct C16 {
    int g, half, lane, n, q4;
    int tok;            // this lane's token in the FT form (n on the lane)
    int first;          // first token of the wave
    int b0, b1;         // first / last clip touched by the wave
    int boundary;       // first token of clip b1 when straddling
    bool straddle, lane_in_b0;
};
DEV C16 make_c16(int g, int half, int lane, int M, int T) {
    C16 c;
    c.g = g;
    c.half = half;
    c.lane = lane;
    c.n = lane & 15;
    c.q4 = lane >> 4;
    c.first = 32 * g + 16 * half;
    c.tok = c.first + c.n;
    const int last = min(c.first + 15, M - 1);
    c.b0 = min(c.first, M - 1) / T;
    c.b1 = last / T;
    c.straddle = c.b1 != c.b0;
    c.boundary = (c.b0 + 1) * T;
    c.lane_in_b0 = c.tok < c.boundary;
    return c;
}

// reductions over the four lane groups (l, l ^ 16, l ^ 32, l ^ 48)
DEV float xq_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
DEV float xq_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// operand fragment of one 32-deep k-step from two consecutive 16-row blocks
template <class T16>
DEV v8<T16> frag2(const f32x4& a, const f32x4& b) {
    v8<T16> f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = (T16)a[j];
        f[4 + j] = (T16)b[j];
    }
    return f;
}
// ... with only the first block (the other 16 k-slots are padding)
template <class T16>
DEV v8<T16> frag1(const f32x4& a) {
    v8<T16> f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = (T16)a[j];
        f[4 + j] = (T16)0.f;
    }
    return f;
}
template <class T16>
DEV v8<T16> zero_frag() {
    v8<T16> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (T16)0.f;
    return f;
}

// residual stream in the 32-token kernels' image: [g][tile t][quarter q][64 lanes][4]; block rb of lane (n, q4) of half-wave
// `half` is the 16-byte piece (t = rb >> 1, q = 2 (rb & 1) + (q4 >> 1), lane n + 16 half + 32 (q4 & 1))
DEV const f32x4* h_piece(const float* hbuf, const C16& c, int rb) {
    return reinterpret_cast<const f32x4*>(hbuf) + (size_t)c.g * 1024 + ((rb >> 1) * 4 + 2 * (rb & 1) + (c.q4 >> 1)) * 64 + c.n + 16 * c.half +
           32 * (c.q4 & 1);
}
DEV void load_h16(f32x4 (&h)[8], const float* __restrict__ hbuf, const C16& c) {
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) h[rb] = *h_piece(hbuf, c, rb);
}
DEV void store_h16(const f32x4 (&h)[8], float* __restrict__ hbuf, const C16& c) {
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) *const_cast<f32x4*>(h_piece(hbuf, c, rb)) = h[rb];
}

// nn.LayerNorm(128) statistics of an activation (8 blocks)
DEV void ln16_stats(const f32x4 (&x)[8], float& mean, float& rstd) {
    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x2 v = {x[rb][2 * p], x[rb][2 * p + 1]};
            s2 += v;
            q2 = __builtin_elementwise_fma(v, v, q2);
        }
    const float s = xq_sum(s2.x + s2.y), q = xq_sum(q2.x + q2.y);
    mean = s * (1.f / 128.f);
    const float var = fmaxf(fmaf(-mean, mean, q * (1.f / 128.f)), 0.f);
    rstd = rsqrtf(var + 1e-5f);
}
template <class T16>
DEV void ln16_frags(v8<T16> (&nb)[4], const f32x4 (&x)[8]) {
    float mean, rstd;
    ln16_stats(x, mean, rstd);
    const float shift = -mean * rstd;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = fmaf(x[2 * m][j], rstd, shift);
            b[j] = fmaf(x[2 * m + 1][j], rstd, shift);
        }
        nb[m] = frag2<T16>(a, b);
    }
}

// acc[rb] += W[rb][:] x   over KM k-steps; weight image [m][rb] fragments in LDS
template <int RB, int KM, class T16>
DEV void gemm16(f32x4 (&acc)[RB], const v8<T16>* __restrict__ w, const v8<T16> (&xb)[KM], int lane) {
#pragma unroll
    for (int m = 0; m < KM; ++m) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16(w[(m * RB + rb) * 64 + lane], xb[m], acc[rb]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct Stats16 {
    f32x2 s = {0.f, 0.f}, q = {0.f, 0.f};
    DEV void add(const f32x4& x) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x2 v = {x[2 * p], x[2 * p + 1]};
            s += v;
            q = __builtin_elementwise_fma(v, v, q);
        }
    }
    // LayerNorm(128) of a StylizationBlock input in the log2(e) scaling of the SiLU: log2(e) nhat = x * rstd + shift
    DEV void finish(float& rstd, float& shift) {
        const float ss = xq_sum(s.x + s.y), qq = xq_sum(q.x + q.y);
        const float mean = ss * (1.f / 128.f);
        const float var = fmaxf(fmaf(-mean, mean, qq * (1.f / 128.f)), 0.f);
        rstd = rsqrtf(var + 1e-5f) * 1.4426950408889634f;
        shift = -mean * rstd;
    }
};
typedef __attribute__((ext_vector_type(2))) _Float16 hh2;
struct Y16 {            // one block of the attention / FFN output as packed f16 (2 registers)
    uint32_t p[2];
};
DEV Y16 pack_y(const f32x4& x) {
    Y16 y;
    const hh2 a = {(_Float16)x[0], (_Float16)x[1]}, b = {(_Float16)x[2], (_Float16)x[3]};
    y.p[0] = __builtin_bit_cast(uint32_t, a);
    y.p[1] = __builtin_bit_cast(uint32_t, b);
    return y;
}

// q = softmax_heads(Wq LN(h) + bq) ; y = q . A per head.  A head = one 16-row block, spread over the four lane groups.
// af: [8 heads][64 lanes] fragments of the wave's clip(s): rows = the head's value features, k-slots of the head PAIR
// (the other head's slots are zero).
template <class T16>
DEV void query_attend16(Y16 (&y)[8], float& y_rstd, float& y_shift, const f32x4 (&h)[8], const float* bq, const v8<T16>* w,
                        const v8<T16>* a0, const v8<T16>* a1, const C16& c) {
    f32x4 q[8];
    {
        v8<T16> nb[4];
        ln16_frags<T16>(nb, h);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) q[rb] = *reinterpret_cast<const f32x4*>(bq + 16 * rb + 4 * c.q4);
        gemm16<8, 4, T16>(q, w, nb, c.lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        const float m = xq_max(fmaxf(fmaxf(q[rb][0], q[rb][1]), fmaxf(q[rb][2], q[rb][3])));
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q[rb][i] = exp2f_fast(q[rb][i] - m);          // q carries log2(e): folded into Wq, bq
            s += q[rb][i];
        }
        const float inv = fast_rcp(xq_sum(s));
#pragma unroll
        for (int i = 0; i < 4; ++i) q[rb][i] *= inv;
    }
    __builtin_amdgcn_sched_barrier(0);
    Stats16 st;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const v8<T16> qb = frag2<T16>(q[2 * m], q[2 * m + 1]);
        f32x4 ya = z4, yb = z4;
        if (!c.straddle) {
            ya = mfma16(a0[(2 * m) * 64 + c.lane], qb, ya);
            yb = mfma16(a0[(2 * m + 1) * 64 + c.lane], qb, yb);
        } else {   // the wave spans two clips: each clip's matrix applies to its own tokens (lanes)
            const v8<T16> zf = zero_frag<T16>();
            const v8<T16> q0 = c.lane_in_b0 ? qb : zf, q1 = c.lane_in_b0 ? zf : qb;
            ya = mfma16(a0[(2 * m) * 64 + c.lane], q0, ya);
            yb = mfma16(a0[(2 * m + 1) * 64 + c.lane], q0, yb);
            ya = mfma16(a1[(2 * m) * 64 + c.lane], q1, ya);
            yb = mfma16(a1[(2 * m + 1) * 64 + c.lane], q1, yb);
        }
        st.add(ya);
        st.add(yb);
        y[2 * m] = pack_y(ya);
        y[2 * m + 1] = pack_y(yb);
    }
    st.finish(y_rstd, y_shift);
}

// FiLM tiles of one 32-feature k-tile for this half-wave: G' - 1 and H' blocks (fb = 0, 1), 4 f16 each
struct E16 {
    uint32_t g[2][2], h[2][2];        // [fb][pair]
};
// global image: tile = [2 parts][64 lanes][8 f16]; lane (n, q4) of half-wave `half` reads the 8-byte half (q4 >> 1) of the
// 16-byte piece of old lane n + 16 half + 32 (q4 & 1), part fb
DEV const uint32_t* e_piece(const f16x8* tile, const C16& c, int fb) {
    return reinterpret_cast<const uint32_t*>(tile + fb * 64 + c.n + 16 * c.half + 32 * (c.q4 & 1)) + 2 * (c.q4 >> 1);
}
DEV void e16_load(E16& e, const f16x8* __restrict__ Eg /* block's 8 tiles */, int kt, const C16& c) {
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        const uint32_t* pg = e_piece(Eg + kt * 128, c, fb);
        const uint32_t* ph = e_piece(Eg + (4 + kt) * 128, c, fb);
        e.g[fb][0] = __builtin_nontemporal_load(pg);
        e.g[fb][1] = __builtin_nontemporal_load(pg + 1);
        e.h[fb][0] = __builtin_nontemporal_load(ph);
        e.h[fb][1] = __builtin_nontemporal_load(ph + 1);
    }
}
// one k-tile of the FiLM-modulated, SiLU'ed operand (see styl_tile): blocks 2 kt, 2 kt + 1
template <class T16>
DEV v8<T16> styl16(const Y16& ya, const Y16& yb, float rstd, float shift, const E16& e) {
    f32x4 z[2];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        const Y16& y = fb ? yb : ya;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float n0 = fma_mix_h<0>(y.p[p], rstd, shift), n1 = fma_mix_h<1>(y.p[p], rstd, shift);
            const f32x2 zz = silu_l2_pair(add_mix_h<0>(e.h[fb][p], fma_mix_h<0>(e.g[fb][p], n0, n0)),
                                          add_mix_h<1>(e.h[fb][p], fma_mix_h<1>(e.g[fb][p], n1, n1)));
            z[fb][2 * p] = zz.x;
            z[fb][2 * p + 1] = zz.y;
        }
    }
    return frag2<T16>(z[0], z[1]);
}
// StylizationBlock accumulated into the residual stream: h += W_o SiLU(nhat G' + H') + b_o
template <class T16>
DEV void styl_accumulate16(f32x4 (&h)[8], const Y16 (&y)[8], float rstd, float shift, const f16x8* __restrict__ Eg, const float* bo,
                           const v8<T16>* w, const C16& c) {
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) h[rb] += *reinterpret_cast<const f32x4*>(bo + 16 * rb + 4 * c.q4);
    E16 e[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) e16_load(e[kt], Eg, kt, c);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const v8<T16> zb = styl16<T16>(y[2 * kt], y[2 * kt + 1], rstd, shift, e[kt]);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) h[rb] = mfma16(w[(kt * 8 + rb) * 64 + c.lane], zb, h[rb]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The workgroup's own combine (all 1024 threads; see wg_combine_attn): unit records of the previous kernel -> attention
// operand fragments of clips ub0, ub0+1 in the 16-token kernel's form: af [2 clips][8 heads][64 lanes], lane (l, q4) of head hd
// = rows l (value feature), k-slots j: (j >> 2) == (hd & 1) ? A[d = 4 q4 + (j & 3)][l] : 0.
// Thread (ci, oc, ln, piece) sums one 16-byte piece of the records' K^T V image: old lane ln = (cc, hh) of tile oc, kept values
// 4 piece + i  <->  head 2 oc + (cc >> 4), d = 8 piece + 4 hh + i, l = cc & 15.
template <class T16>
DEV void wg_combine_attn16(const float* __restrict__ recs, v8<T16>* af, float* scratch, int ub0, int M, int T, int tid, int wg) {
    constexpr int NU = 17, PRE = 9;
    float* wsc = scratch;
    float* zsc = scratch + 2 * NU * 128;
    const int ci = tid >> 9, piece = tid & 1, ln = (tid >> 1) & 63, oc = (tid >> 7) & 3;
    const int cc = ln & 31, hh = ln >> 5;
    const int b = ub0 + ci;
    const int ub1 = (min((wg + 1) * 256, M) - 1) / T;
    const bool live = b <= ub1;
    const int bv = live ? b : ub0;
    const int v_lo = (bv * T) / 256, v_hi = (min((bv + 1) * T, M) - 1) / 256;
    const int nu = live ? v_hi - v_lo + 1 : 0;
    auto rec_of = [&](int clip, int u) { return recs + ((size_t)u * 2 + ((u * 256 >= clip * T) ? 0 : 1)) * DC_REC_FLOATS; };
    f32x4 pre[PRE];
    if (live) {
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            pre[k] = reinterpret_cast<const f32x4*>(rec_of(bv, min(v_lo + k, v_hi)) + 256)[(oc * 64 + ln) * 2 + piece];
    }
    if (tid < 256) {       // per (clip, feature): the units' maxima / sums -> rescale weights and the normaliser
        const int ca = tid >> 7, f = tid & 127, ba = ub0 + ca;
        const bool la = ba <= ub1;
        const int bav = la ? ba : ub0;
        const int a_lo = (bav * T) / 256, a_hi = (min((bav + 1) * T, M) - 1) / 256;
        const int na = la ? a_hi - a_lo + 1 : 0;
        float mr[PRE], sr[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const float* R = rec_of(bav, min(a_lo + k, a_hi));
            mr[k] = R[f];
            sr[k] = k < na ? R[128 + f] : 0.f;
        }
        float mstar = -INFINITY;
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            if (sr[k] > 0.f) mstar = fmaxf(mstar, mr[k]);
        for (int k = PRE; k < na; ++k) {
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
    const int rowb = 32 * oc + 16 * (cc >> 4) + 8 * piece + 4 * hh;       // value i <-> K feature (row) rowb + i
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PRE; ++k)
        if (k < nu) acc += *reinterpret_cast<const f32x4*>(wsc + (ci * NU + k) * 128 + rowb) * pre[k];
    for (int k = PRE; k < nu; ++k)
        acc += *reinterpret_cast<const f32x4*>(wsc + (ci * NU + k) * 128 + rowb) *
               reinterpret_cast<const f32x4*>(rec_of(b, v_lo + k) + 256)[(oc * 64 + ln) * 2 + piece];
    const f32x4 z4 = *reinterpret_cast<const f32x4*>(zsc + ci * 128 + rowb);
    const int hd = 2 * oc + (cc >> 4), e = hd & 1, l = cc & 15;
    v4<T16> val, zero;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        val[i] = (T16)(z4[i] > 0.f ? acc[i] * fast_rcp(z4[i]) : 0.f);
        zero[i] = (T16)0.f;
    }
    v4<T16>* dst = reinterpret_cast<v4<T16>*>(af + (size_t)(ci * 8 + hd) * 64 + l + 16 * (2 * piece + hh));      // q4 = 2 piece + hh
    dst[e] = val;
    dst[e ^ 1] = zero;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// k_layer16: one decoder layer for one 256-token unit, 16 waves x 16 tokens.  Same stages, staging and barriers as k_layer.
// stop_after (test hook, wave-uniform): 1 / 2 / 3 = leave after the self-attention / cross-attention / FFN block.
// ------------------------------------------------------------------------------------------------------------------
template <class T16, bool STAMP>
__global__ __launch_bounds__(1024, 1)
void k_layer16(const DcModel* __restrict__ dm, int l, float* __restrict__ hbuf, const f16x16* __restrict__ E, int NT,
               const v8<T16>* __restrict__ a_ca /*[L][B][16][64] 32-token form*/, float* __restrict__ recs, const int* __restrict__ length,
               const float* __restrict__ xin, float* __restrict__ xout, int out_mode, const float* __restrict__ coef_cur,
               const int* __restrict__ snap_cur, float* __restrict__ snaps, int M, int T, int G, int B, int stop_after,
               size_t rec_stride, const int* __restrict__ iter_base, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using W = v8<T16>;
    constexpr int NW = 16;
    // diagnostic build: 100 MHz timestamps per stage for the 16 waves of workgroup 3 of layer 3 (tools/stage_stamps16.py)
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            if (stamps && blockIdx.x == 3 && (threadIdx.x & 63) == 0 && l == 3)
                stamps[264 + (threadIdx.x >> 6) * 32 + k] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wg = wg_index();
    int g = wg * 8 + (wave >> 1);
    const bool active = g < G;                   // idle waves still take part in the staging and barriers
    if (!active) g = G - 1;
    const C16 c = make_c16(g, wave & 1, lane, M, T);
    char* buf0 = lds;
    char* buf1 = lds + L16_WSZ;
    const W* w0 = reinterpret_cast<const W*>(buf0);
    const W* w1 = reinterpret_cast<const W*>(buf1);
    const float* c0 = reinterpret_cast<const float*>(buf0 + 32 * 1024);      // constants block behind the 32 fragments
    const float* c1 = reinterpret_cast<const float*>(buf1 + 32 * 1024);
    const int ub0 = (wg * 256) / T;
    const W* af = reinterpret_cast<const W*>(lds + L16_OFF_AF);
    const DcLayer16& L = dm->l16[l];
    const int nl = dm->num_layers;
    const bool last = l + 1 >= nl;
    const f16x8* Eg = reinterpret_cast<const f16x8*>(E) + ((size_t)g * NT + (size_t)l * 24) * 128;   // 3 blocks x 8 tiles
    const float* recs_in = recs + (size_t)(l & 1) * rec_stride;
    float* recs_out = recs + (size_t)((l + 1) & 1) * rec_stride;

    f32x4 h[8];
    stamp(0);
    load_h16(h, hbuf, c);
    stage_frags<NW>(L.sa_q, buf0, 33, wave, lane);
    wg_combine_attn16<T16>(recs_in, reinterpret_cast<W*>(lds + L16_OFF_AF), reinterpret_cast<float*>(buf1), ub0, M, T, tid, wg);
    stage_sync();
    stamp(1);

    // ---- self-attention
    stage_frags<NW>(L.sa_o, buf1, 33, wave, lane);
    Y16 y[8];
    float y_rstd, y_shift;
    query_attend16<T16>(y, y_rstd, y_shift, h, c0, w0, af + (size_t)(c.b0 - ub0) * 8 * 64, af + (size_t)(c.b1 - ub0) * 8 * 64, c);
    stamp(2);
    stage_sync();
    stamp(3);
    {   // cross-attention query image + the cross-attention fragments of clips ub0, ub0+1 (built from the 32-token form by
        // k_cond_af16 into dm-independent storage: a_ca here IS that 16-token form: [L][B][8 heads][64 lanes])
        stage_frags<NW>(L.ca_q, buf0, 33, wave, lane);
        const W* acl = a_ca + ((size_t)l * B) * 8 * 64;
        const int c1i = min(ub0 + 1, B - 1);
        stage_frags<NW>(acl + (size_t)ub0 * 8 * 64, lds + L16_OFF_AF, 8, wave, lane);
        stage_frags<NW>(acl + (size_t)c1i * 8 * 64, lds + L16_OFF_AF + 8192, 8, wave, lane);
    }
    styl_accumulate16<T16>(h, y, y_rstd, y_shift, Eg, c1, w1, c);
    stamp(4);
    if (stop_after == 1) {
        if (active) store_h16(h, hbuf, c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    stage_sync();
    stamp(5);
    // ---- cross-attention
    stage_frags<NW>(L.ca_o, buf1, 33, wave, lane);
    query_attend16<T16>(y, y_rstd, y_shift, h, c0, w0, af + (size_t)(c.b0 - ub0) * 8 * 64, af + (size_t)(c.b1 - ub0) * 8 * 64, c);
    stamp(6);
    stage_sync();
    stamp(7);
    stage_frags<NW>(L.ffn_w, buf0, 33, wave, lane);          // W1 (16 fragments) | W2 (16) | b1[64], b2[128]
    styl_accumulate16<T16>(h, y, y_rstd, y_shift, Eg + 8 * 128, c1, w1, c);
    stamp(8);
    if (stop_after == 2) {
        if (active) store_h16(h, hbuf, c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    stage_sync();
    stamp(9);
    // ---- FFN
    stage_frags<NW>(L.ffn_o, buf1, 33, wave, lane);
    {
        f32x4 u[4];
        {
            W hb[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) hb[m] = frag2<T16>(h[2 * m], h[2 * m + 1]);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) u[rb] = *reinterpret_cast<const f32x4*>(c0 + 16 * rb + 4 * c.q4);            // b1
            gemm16<4, 4, T16>(u, w0, hb, lane);
        }
        W ub[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 ga, gb;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x2 a = gelu_erf_pair(u[2 * m][2 * p], u[2 * m][2 * p + 1]), b = gelu_erf_pair(u[2 * m + 1][2 * p], u[2 * m + 1][2 * p + 1]);
                ga[2 * p] = a.x;
                ga[2 * p + 1] = a.y;
                gb[2 * p] = b.x;
                gb[2 * p + 1] = b.y;
            }
            ub[m] = frag2<T16>(ga, gb);
        }
        f32x4 yf[8];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) yf[rb] = *reinterpret_cast<const f32x4*>(c0 + 64 + 16 * rb + 4 * c.q4);         // b2
        gemm16<8, 2, T16>(yf, w0 + 16 * 64, ub, lane);
        Stats16 st;
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            st.add(yf[rb]);
            y[rb] = pack_y(yf[rb]);
        }
        st.finish(y_rstd, y_shift);
    }
    stamp(10);
    stage_sync();
    stamp(11);
    if (!last)
        stage_frags<NW>(dm->l16[l + 1].sa_k, buf0, 33, wave, lane);
    else
        stage_frags<NW>(dm->out16, buf0, 17, wave, lane);        // 8 hi + 8 lo fragments + bias: always runs split
    styl_accumulate16<T16>(h, y, y_rstd, y_shift, Eg + 16 * 128, c1, w1, c);
    stamp(12);
    if (stop_after == 3) {
        if (active) store_h16(h, hbuf, c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    stage_sync();
    stamp(13);

    if (!last) {
        // ---- next layer's self-attention front half: K [buf0], V [buf1] in TF form (token on the ROW: lane = feature), unit record
        stage_frags<NW>(dm->l16[l + 1].sa_v, buf1, 33, wave, lane);
        if (active) store_h16(h, hbuf, c);
        W nb[4];
        ln16_frags<T16>(nb, h);
        float* mx = reinterpret_cast<float*>(lds + L16_OFF_AF);               // [128 features][2 slots][16 waves] = 16 KiB
        char* pst = lds + L16_OFF_ER;                                          // [16 waves][4 blocks][64 lanes] f32x4 = 64 KiB per round
        float* ss = reinterpret_cast<float*>(lds + L16_OFF_SS);                // column sums [16 waves + the straddler's second slot][128]
        float* scw = reinterpret_cast<float*>(buf0) + wave * 256;              // after the barrier: this wave's rescale factors [2 slots][128]
        f32x4* xp = reinterpret_cast<f32x4*>(buf0 + 16384);                    // second slot of the straddling wave: [4 blocks][64]
        const int f = lane & 15;                                               // TF form: this lane's feature inside a block
        // validity of this lane's four token rows (4 q4 + i) for the two clip slots
        const int s0 = c.b0 - ub0;
        const bool strad = active && c.straddle;
        bool okA[4], okB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tk = c.first + 4 * c.q4 + i;
            const int bb = min(tk, M - 1) / T, nn = tk - bb * T;
            const bool ok = active && tk < M && nn < min(length[bb], T);
            okA[i] = ok && bb == c.b0;
            okB[i] = ok && bb != c.b0;
        }
        // exp2(K - wave maximum) per clip slot as the first halves of A-operand fragments; maxima and sums of slot A in registers,
        // of slot B (only the wave that straddles two clips has one) in LDS
        v4<T16> efA[8], efB[8];
        float ssA[8], mA[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const float bk = c0[16 * cb + f];
            f32x4 K = {bk, bk, bk, bk};
#pragma unroll
            for (int m = 0; m < 4; ++m) K = mfma16(nb[m], w0[(m * 8 + cb) * 64 + lane], K);
            auto keys = [&](const bool (&ok)[4], v4<T16>& ef, float& ssum, float& mcol) {
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i) m = ok[i] ? fmaxf(m, K[i]) : m;
                m = xq_max(m);
                mcol = m;
                const float mz = m == -INFINITY ? 0.f : m;
                float sacc = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = ok[i] ? exp2f_fast(K[i] - mz) : 0.f;
                    sacc += e;
                    ef[i] = (T16)e;
                }
                ssum = xq_sum(sacc);
            };
            keys(okA, efA[cb], ssA[cb], mA[cb]);
            float mB = -INFINITY;
            if (strad) {
                float sB;
                keys(okB, efB[cb], sB, mB);
                if (c.q4 == 0) ss[16 * 128 + 16 * cb + f] = sB;
            }
            if (c.q4 == 0) {
                mx[((16 * cb + f) * 2 + s0) * 16 + wave] = mA[cb];
                mx[((16 * cb + f) * 2 + (s0 ^ 1)) * 16 + wave] = s0 ? -INFINITY : mB;
            }
            if (cb & 1) __builtin_amdgcn_sched_barrier(0);
        }
        stamp(14);
        __builtin_amdgcn_sched_barrier(0);
        if (active)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // all but the 8 stores of h: the value image has landed
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        stamp(15);
        auto unit_max = [&](int feat, int sl) {
            const f32x4* p = reinterpret_cast<const f32x4*>(mx + (feat * 2 + sl) * 16);
            const f32x4 a = p[0], b = p[1], cc = p[2], d = p[3];
            const float m = fmaxf(fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))),
                                  fmaxf(fmaxf(fmaxf(cc[0], cc[1]), fmaxf(cc[2], cc[3])), fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]))));
            return m == -INFINITY ? 0.f : m;
        };
        // (buf0's key image is consumed: scw / xp live there now)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const float fa = mA[cb] == -INFINITY ? 0.f : exp2f_fast(mA[cb] - unit_max(16 * cb + f, s0));
            if (c.q4 == 0) {
                scw[16 * cb + f] = fa;
                ss[wave * 128 + 16 * cb + f] = ssA[cb] * fa;
            }
            if (strad) {
                const float mB = mx[((16 * cb + f) * 2 + 1) * 16 + wave];
                const float fb = mB == -INFINITY ? 0.f : exp2f_fast(mB - unit_max(16 * cb + f, 1));
                if (c.q4 == 0) {
                    scw[128 + 16 * cb + f] = fb;
                    ss[16 * 128 + 16 * cb + f] *= fb;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(16);
        // two rounds of four heads: every wave stages its K^T V blocks, the waves 0..7 sum block (round * 4 + (w & 3)) of slot w >> 2
        // over the 16 contributors in wave order and write the unit record in the 32-token kernels' format
        f32x4* mine = reinterpret_cast<f32x4*>(pst) + (size_t)wave * 4 * 64 + lane;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const v4<T16> zh = {(T16)0.f, (T16)0.f, (T16)0.f, (T16)0.f};
        auto widen = [&](const v4<T16>& a) {
            W r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = a[j];
                r[4 + j] = zh[j];
            }
            return r;
        };
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (rnd) __syncthreads();             // the sums of round 0 are done
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cb = 4 * rnd + k;
                const float bv = c1[16 * cb + f];
                f32x4 V = {bv, bv, bv, bv};
#pragma unroll
                for (int m = 0; m < 4; ++m) V = mfma16(nb[m], w1[(m * 8 + cb) * 64 + lane], V);
                f32x4 va, vb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    va[i] = okA[i] ? V[i] : 0.f;
                    vb[i] = okB[i] ? V[i] : 0.f;
                }
                // P[d][l] = sum over the wave's tokens of exp(K - m)[tok][d] V[tok][l]; then rescale row d to the unit maximum
                f32x4 PA = mfma16(widen(efA[cb]), frag1<T16>(va), z4);
                PA *= *reinterpret_cast<const f32x4*>(scw + 16 * cb + 4 * c.q4);
                mine[k * 64] = PA;
                if (strad) {
                    f32x4 PB = mfma16(widen(efB[cb]), frag1<T16>(vb), z4);
                    PB *= *reinterpret_cast<const f32x4*>(scw + 128 + 16 * cb + 4 * c.q4);
                    xp[k * 64 + lane] = PB;
                }
                if (k & 1) __builtin_amdgcn_sched_barrier(0);
            }
            stamp(17 + 3 * rnd);
            __syncthreads();
            stamp(18 + 3 * rnd);
            if (wave < 8) {
                const int k = wave & 3, cb = 4 * rnd + k, sl = wave >> 2;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                float ssum = 0.f;
                const int edge = (ub0 + 1) * T;                      // first token of slot 1's clip
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int fv = 32 * (wg * 8 + (v >> 1)) + 16 * (v & 1);     // first token of contributor v
                    if (wg * 8 + (v >> 1) >= G) continue;
                    const int s0v = fv >= edge ? 1 : 0;
                    const bool sv = !s0v && min(fv + 15, M - 1) >= edge;
                    if (s0v == sl) {
                        acc += reinterpret_cast<const f32x4*>(pst)[(size_t)(v * 4 + k) * 64 + lane];
                        ssum += ss[v * 128 + 16 * cb + f];
                    }
                    if (sv && sl == 1) {
                        acc += xp[k * 64 + lane];
                        ssum += ss[16 * 128 + 16 * cb + f];
                    }
                }
                float* R = recs_out + ((size_t)wg * 2 + sl) * DC_REC_FLOATS;
                if (c.q4 == 0) {
                    R[16 * cb + f] = unit_max(16 * cb + f, sl);
                    R[128 + 16 * cb + f] = ssum;
                }
                // lane (l = lane & 15, q4) holds P[d = 4 q4 + i][l]: the 32-token record keeps it in tile oc = cb >> 1, old lane
                // 16 (cb & 1) + l + 32 (q4 & 1), values 4 (q4 >> 1) + i
                reinterpret_cast<f32x4*>(R + 256 + ((cb >> 1) * 64 + 16 * (cb & 1) + f + 32 * (c.q4 & 1)) * 8)[c.q4 >> 1] = acc;
            }
            stamp(19 + 3 * rnd);
        }
        return;
    }
    // ---- output projection [buf0: 8 hi + 8 lo fragments, bias behind them] + DDIM update
    f32x4 x0[2];
    {
        const float* ob = reinterpret_cast<const float*>(buf0 + 16 * 1024);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) x0[rb] = *reinterpret_cast<const f32x4*>(ob + 16 * rb + 4 * c.q4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 la, lb;
            W hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const T16 a = (T16)h[2 * m][j], b = (T16)h[2 * m + 1][j];
                hi[j] = a;
                hi[4 + j] = b;
                la[j] = h[2 * m][j] - (float)a;
                lb[j] = h[2 * m + 1][j] - (float)b;
            }
            lo = frag2<T16>(la, lb);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const W wh = w0[(m * 2 + rb) * 64 + lane], wl = w0[(8 + m * 2 + rb) * 64 + lane];
                x0[rb] = mfma16(wh, hi, x0[rb]);
                x0[rb] = mfma16(wh, lo, x0[rb]);
                x0[rb] = mfma16(wl, hi, x0[rb]);
            }
        }
    }
    if (!active || c.tok >= M) return;
    const int P = dm->input_feats;
    float sr = 0.f, srm1 = 1.f, cx0 = 0.f, ceps = 0.f;
    int snap = -1;
    if (out_mode != 0) {
        const int ib = iter_base ? *iter_base : 0;
        coef_cur += 4 * ib;
        sr = coef_cur[0];
        srm1 = coef_cur[1];
        cx0 = coef_cur[2];
        ceps = coef_cur[3];
        snap = snap_cur[ib];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ft = 16 * rb + 4 * c.q4 + i;
            if (ft < P) {
                const size_t o = (size_t)c.tok * P + ft;
                if (out_mode == 0) {
                    xout[o] = x0[rb][i];
                } else {
                    const float xt = xin[o];
                    const float eps = (sr * xt - x0[rb][i]) / srm1;
                    const float xn = x0[rb][i] * cx0 + ceps * eps;
                    xout[o] = xn;
                    if (snap >= 0) snaps[(size_t)snap * M * P + o] = xn;
                }
            }
        }
}

// cross-attention fragments of the pre-pass, 32-token form [L*B][16 frags (8 hi + 8 lo)][64][8] -> 16-token form [L*B][8 heads][64][8]:
// source fragment (oc, s), lane (c, hh), element j = A[d = 8 (j >> 2) + 4 hh + (j & 3)][l = c & 15] of head 2 oc + s for c >> 4 == s
template <class T16>
__global__ void k_cond_af16(const v8<T16>* __restrict__ src, v8<T16>* __restrict__ dst, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (matrix, head, lane)
    if (idx >= n * 8 * 64) return;
    const int lane = idx & 63, hd = (idx >> 6) & 7, mtx = idx >> 9;
    const int l = lane & 15, q4 = lane >> 4, oc = hd >> 1, s = hd & 1;
    // d = 4 q4 + i  ->  source: hh = q4 & 1, j = 4 (q4 >> 1) + i, lane c = 16 s + l
    const v8<T16> a = src[((size_t)mtx * 16 + oc * 2 + s) * 64 + 16 * s + l + 32 * (q4 & 1)];
    v8<T16> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T16)0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[4 * s + i] = a[4 * (q4 >> 1) + i];
    dst[((size_t)mtx * 8 + hd) * 64 + lane] = o;
}

static hipError_t l16_optin(const void* fn) {
    static unsigned long long done[3] = {0, 0, 0};
    static const void* fns[3] = {nullptr, nullptr, nullptr};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int slot = 0;
    while (slot < 2 && fns[slot] != fn && fns[slot] != nullptr) ++slot;
    fns[slot] = fn;
    if (dev < 64 && ((done[slot] >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS);
    if (e == hipSuccess && dev < 64) done[slot] |= 1ull << dev;
    return e;
}

hipError_t dc_launch_layer16(hipStream_t st, int fmt, const DcModel* dm, int l, float* hbuf, const void* E, int NT, const void* a_ca16,
                             float* recs, const int* length, const float* xin, float* xout, int out_mode, const float* coef_cur,
                             const int* snap_cur, float* snaps, int M, int T, int G, int B, int stop_after, size_t rec_stride,
                             const int* iter_base, unsigned long long* stamps) {
    const dim3 grid((G + 7) / 8), block(1024);
    if (stamps) {      // diagnostic build (DC_STAMPS=1)
        if (fmt != 1) return hipErrorInvalidValue;
        if (hipError_t e = l16_optin((const void*)k_layer16<_Float16, true>)) return e;
        k_layer16<_Float16, true><<<grid, block, L16_LDS, st>>>(dm, l, hbuf, (const f16x16*)E, NT, (const f16x8*)a_ca16, recs, length, xin, xout,
                                                                out_mode, coef_cur, snap_cur, snaps, M, T, G, B, stop_after, rec_stride, iter_base,
                                                                stamps);
        return hipGetLastError();
    }
    if (fmt == 1) {
        if (hipError_t e = l16_optin((const void*)k_layer16<_Float16, false>)) return e;
        k_layer16<_Float16, false><<<grid, block, L16_LDS, st>>>(dm, l, hbuf, (const f16x16*)E, NT, (const f16x8*)a_ca16, recs, length, xin, xout,
                                                          out_mode, coef_cur, snap_cur, snaps, M, T, G, B, stop_after, rec_stride, iter_base, nullptr);
    } else {
        if (hipError_t e = l16_optin((const void*)k_layer16<__bf16, false>)) return e;
        k_layer16<__bf16, false><<<grid, block, L16_LDS, st>>>(dm, l, hbuf, (const f16x16*)E, NT, (const bf16x8*)a_ca16, recs, length, xin, xout,
                                                        out_mode, coef_cur, snap_cur, snaps, M, T, G, B, stop_after, rec_stride, iter_base, nullptr);
    }
    return hipGetLastError();
}
hipError_t dc_launch_cond_af16(hipStream_t st, int fmt, const void* a_ca, void* a_ca16, int n_matrices) {
    const int n = n_matrices * 8 * 64;
    if (fmt == 1)
        k_cond_af16<_Float16><<<dim3((n + 255) / 256), dim3(256), 0, st>>>((const f16x8*)a_ca, (f16x8*)a_ca16, n_matrices);
    else
        k_cond_af16<__bf16><<<dim3((n + 255) / 256), dim3(256), 0, st>>>((const bf16x8*)a_ca, (bf16x8*)a_ca16, n_matrices);
    return hipGetLastError();
}
