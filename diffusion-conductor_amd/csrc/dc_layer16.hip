// dc_layer16.hip - the decoder layer for SMALL batches on SIXTEEN tokens per wave (v_mfma_f32_16x16x32), non-split formats.
//
// Why.  With one clip per call - the reference's own call pattern (trainers/ddpm_trainer.py:184, tools/eval_new.py:113-121) - the
// layer kernel is not short of issue slots (94 % of the CUs are idle) but of chain LENGTH: a 32-token wave that has its SIMD to
// itself still walks LN -> 32 MFMAs -> softmax -> ... -> record tail one link after the other, 34 us per layer.  A 16-token wave
// carries half the per-lane vector work and half the matrix-pipe time per link (16x16x32: 16 cycles per MFMA instead of 32), so the
// same chain is about half as long; the batch is small enough for every 64-token unit (4 waves, ONE wave per SIMD) to get a CU.
// (Round 2 built this formulation for the FULL chip - 16 waves per 256-token unit, four waves per SIMD at 128 VGPRs - where issue
// slots are the limit and it lost: tools/negative_results.  Here it runs at one wave per SIMD with the whole register file.)
//
// Units.  Clip-aligned 64-token units: clip b owns upc = ceil(T / 64) workgroups (T = clip stride, a multiple of 32); wave w of
// unit u holds tokens 64 u + 16 w .. + 15 of the clip.  No wave and no unit spans two clips, so a clip's result does not depend on
// the batch around it.  One unit record per workgroup (slot 0 of the 32-token kernels' record format, so that layer 0 combines the
// 128-token unit records k_embed_front's narrow form writes, and later layers the 64-token ones written here).
//
// Combine (round 5).  The attention operand of a layer is the clip-wide sum of the previous layer's unit records.  Every workgroup
// reduces one slice of it, publishes the slice as tagged 8-byte granules and gathers the clip's operand from its neighbours inside the
// launch, behind LayerNorm, query projection and softmax of the self-attention stage (Slice16, gather16_*): 8 KiB read per workgroup
// instead of 261, -8 % per layer launch with one clip per call.  DC_L16_OWN_COMBINE=1 keeps round 4's form (wg_combine_attn16).
//
// Layout.  Accumulator tile of v_mfma_f32_16x16x32: lane l holds column n = l & 15 (the TOKEN) and rows 4 (l >> 4) + i, i < 4, of a
// 16-row block.  An activation of 128 features is 8 such blocks: x[rb][i] = feature 16 rb + 4 q4 + i, q4 = l >> 4.  Two consecutive
// blocks convert in registers into the B operand of a 32-deep k-step: element j of lane (n, q4) is
//   feature 32 m + 16 (j >> 2) + 4 q4 + (j & 3)            ("chained" k order; the host packs the weights to match, DcLayer16),
// and the same registers are the A operand of the transposed products (X^T W) of the record tail.  Everything in HBM keeps the
// layouts of the 32-token kernels (residual stream, FiLM tiles, unit records), addressed at 16- or 8-byte granularity - so
// k_embed_front, the FiLM GEMM and the conditioning pre-pass are shared.
#include "dc_dev.h"
#include "dc_launch.h"

namespace {

constexpr int L16_NW = 4;
constexpr int L16_WSZ = 33 * 1024;
constexpr int L16_OFF_AF = 2 * L16_WSZ;              // attention fragments of the clip: [8 heads][64 lanes] (8 KiB); tail: column maxima [128][4]
constexpr int L16_OFF_PST = L16_OFF_AF + 8192;       // tail: K^T V staging [4 waves][8 blocks][64 lanes] f32x4 (32 KiB)
constexpr int L16_OFF_SS = L16_OFF_PST + 32768;      // tail: column sums [4 waves][128] (2 KiB)
constexpr int L16_OFF_SCW = L16_OFF_SS + 2048;       // tail: per-wave rescale factors [4 waves][128] (2 KiB)
constexpr int L16_LDS = L16_OFF_SCW + 2048;
constexpr int L16_MAXU = 32;                         // unit records per clip the combine holds (T <= 2048 at 64 tokens per unit)
static_assert(L16_LDS - L16_OFF_PST >= 2 * 128 * (L16_MAXU + 4) * 4, "the combine's transposed scalars overlay the tail's staging");

struct C16 {
    int g, half, lane, n, q4;
    int tok;            // this lane's token in the FT form (n on the lane), flat in the token space (clip stride T)
    int first;          // first token of the wave (flat)
    int b, n0;          // clip, first token of the wave inside the clip
};

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
    for (int rb = 0; rb < 8; ++rb) __builtin_nontemporal_store(h[rb], const_cast<f32x4*>(h_piece(hbuf, c, rb)));
}

// nn.LayerNorm(128) statistics of an activation (8 blocks); the affine is folded into the projection that follows
DEV void ln16_stats(const f32x4 (&x)[8], float& mean, float& rstd) {
    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x2 v = {x[rb][2 * p], x[rb][2 * p + 1]};
            s2 += v;
            q2 = fma2(v, v, q2);
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

// acc[rb] += W[rb][:] x   over KM k-steps; weight image [m][rb] fragments in LDS (RB independent accumulator chains per k-step)
// At one wave per SIMD nobody covers an LDS read's latency: the RB weight fragments of k-step m + 1 are requested before the MFMAs of
// k-step m issue (hipcc on its own emits read - wait - MFMA per fragment, ~120 exposed cycles per MFMA).
template <int RB, class T16>
DEV void wfrags(v8<T16> (&f)[RB], const v8<T16>* __restrict__ w, int first, int lane) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) f[rb] = w[(first + rb) * 64 + lane];
}
template <int RB, int KM, class T16>
DEV void gemm16(f32x4 (&acc)[RB], const v8<T16>* __restrict__ w, const v8<T16> (&xb)[KM], int lane) {
    v8<T16> cur[RB], nxt[RB];
    wfrags<RB, T16>(cur, w, 0, lane);
#pragma unroll
    for (int m = 0; m < KM; ++m) {
        if (m + 1 < KM) wfrags<RB, T16>(nxt, w, (m + 1) * RB, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma16(cur[rb], xb[m], acc[rb]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) cur[rb] = nxt[rb];
    }
}

struct Stats16 {
    f32x2 s = {0.f, 0.f}, q = {0.f, 0.f};
    DEV void add(const f32x4& x) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x2 v = {x[2 * p], x[2 * p + 1]};
            s += v;
            q = fma2(v, v, q);
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
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
struct Y16 {            // one block of the attention / FFN output as packed f16 (2 registers), as the 32-token kernels keep it
    uint32_t p[2];
};
DEV Y16 pack_y(const f32x4& x) {
    Y16 y;
    const hh2 a = {(_Float16)x[0], (_Float16)x[1]}, b = {(_Float16)x[2], (_Float16)x[3]};
    y.p[0] = __builtin_bit_cast(uint32_t, a);
    y.p[1] = __builtin_bit_cast(uint32_t, b);
    return y;
}

// q = softmax_heads(Wq LN(h) + bq) ; y = q . A per head (transformer.py:104,109,119 / 147,150,156).  A head = one 16-row block,
// spread over the four lane groups.  af: [8 heads][64 lanes] fragments of the wave's clip: rows = the head's value features,
// k-slots of the head PAIR (the other head's slots are zero).
struct NoMid {
    DEV void operator()() const {}
};
template <class T16, class Mid = NoMid>
DEV void query_attend16(Y16 (&y)[8], float& y_rstd, float& y_shift, const f32x4 (&h)[8], const float* bq, const v8<T16>* w,
                        const v8<T16>* af, const C16& c, Mid&& mid = Mid() /* runs between the softmax and the first read of af */) {
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
    mid();
    __builtin_amdgcn_sched_barrier(0);
    Stats16 st;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    v8<T16> afr[8];
    wfrags<8, T16>(afr, af, 0, c.lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const v8<T16> qb = frag2<T16>(q[2 * m], q[2 * m + 1]);
        const f32x4 ya = mfma16(afr[2 * m], qb, z4);
        const f32x4 yb = mfma16(afr[2 * m + 1], qb, z4);
        st.add(ya);
        st.add(yb);
        y[2 * m] = pack_y(ya);
        y[2 * m + 1] = pack_y(yb);
    }
    st.finish(y_rstd, y_shift);
}

// FiLM tiles of one 32-feature k-tile for this half-wave: G' - 1 and H' blocks (fb = 0, 1), 4 f16 each
struct E16 {
    u32x2 g[2], h[2];        // [fb]
};
// global image: tile = [2 parts][64 lanes][8 f16]; lane (n, q4) of half-wave `half` reads the 8-byte half (q4 >> 1) of the
// 16-byte piece of old lane n + 16 half + 32 (q4 & 1), part fb
DEV const u32x2* e_piece(const f16x8* tile, const C16& c, int fb) {
    return reinterpret_cast<const u32x2*>(tile + fb * 64 + c.n + 16 * c.half + 32 * (c.q4 & 1)) + (c.q4 >> 1);
}
DEV void e16_load(E16 (&e)[4], const f16x8* __restrict__ Eg /* block's 8 tiles */, const C16& c) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            e[kt].g[fb] = __builtin_nontemporal_load(e_piece(Eg + kt * 128, c, fb));
            e[kt].h[fb] = __builtin_nontemporal_load(e_piece(Eg + (4 + kt) * 128, c, fb));
        }
}
// one k-tile of the FiLM-modulated, SiLU'ed operand (see styl_tile in dc_dev.h): blocks 2 kt, 2 kt + 1
template <class T16, bool G1 = false>
DEV v8<T16> styl16(const Y16& ya, const Y16& yb, float rstd, float shift, const E16& e) {
    f32x4 z[2];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        const Y16& y = fb ? yb : ya;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float n0 = fma_mix_h<0>(y.p[p], rstd, shift), n1 = fma_mix_h<1>(y.p[p], rstd, shift);
            const f32x2 zz = silu_l2_pair(film_affine<0, G1>(e.g[fb][p], n0, e.h[fb][p]), film_affine<1, G1>(e.g[fb][p], n1, e.h[fb][p]));
            z[fb][2 * p] = zz.x;
            z[fb][2 * p + 1] = zz.y;
        }
    }
    return frag2<T16>(z[0], z[1]);
}
// StylizationBlock (transformer.py:68-81) accumulated into the residual stream: h += W_o SiLU(nhat G' + H') + b_o
template <class T16, bool G1 = false>
DEV void styl_accumulate16(f32x4 (&h)[8], const Y16 (&y)[8], float rstd, float shift, const E16 (&e)[4], const float* bo,
                           const v8<T16>* w, const C16& c) {
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) h[rb] += *reinterpret_cast<const f32x4*>(bo + 16 * rb + 4 * c.q4);
    v8<T16> cur[8], nxt[8];
    wfrags<8, T16>(cur, w, 0, c.lane);
    v8<T16> zb = styl16<T16, G1>(y[0], y[1], rstd, shift, e[0]);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt + 1 < 4) wfrags<8, T16>(nxt, w, (kt + 1) * 8, c.lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) h[rb] = mfma16(cur[rb], zb, h[rb]);
        if (kt + 1 < 4) zb = styl16<T16, G1>(y[2 * kt + 2], y[2 * kt + 3], rstd, shift, e[kt + 1]);     // (vector work beside the matrix pipe)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) cur[rb] = nxt[rb];
    }
}

// The workgroup's own combine (256 threads; see wg_combine_attn_narrow in dc_dev.h): the `nu` unit records of clip b written by the
// previous kernel (unit k of the clip at recs + (b * nu + k) * stride floats) -> the clip's attention operand fragments in this
// kernel's form: af [8 heads][64 lanes], lane (l, q4) of head hd = rows l (value feature), k-slots j:
// (j >> 2) == (hd & 1) ? A[d = 4 q4 + (j & 3)][l] : 0.   Thread (oc, ln) sums both 16-byte pieces of its 32 bytes of the records'
// K^T V image: old lane ln = (cc, hh) of tile oc, kept values 4 piece + i  <->  head 2 oc + (cc >> 4), d = 8 piece + 4 hh + i,
// l = cc & 15.  Fixed summation order (unit by unit).  scratch (LDS): w [L16_MAXU][128] + z [128] floats (16.5 KiB); scratch2: the
// units' scalars transposed, [2][128][L16_MAXU + 4] floats (36 KiB).
template <class T16>
DEV void wg_combine_attn16(const float* __restrict__ recs, size_t stride, int nu, int b, v8<T16>* af, float* scratch, float* scratch2, int tid) {
    constexpr int PRE = 8;                           // K^T V blocks per batch (two batches in flight: 128 registers)
    static_assert(L16_MAXU == 4 * PRE, "four batches");
    constexpr int FS = L16_MAXU + 4;                 // floats per feature row of the transposed scalar arrays (36: conflict-free b128 rows)
    float* wsc = scratch;                            // [L16_MAXU][128]: the units' rescale weights
    float* zsc = scratch + L16_MAXU * 128;           // [128]
    float* tsc = scratch2;                           // [2 parts][128 features][FS]: the units' column maxima (part 0) / sums (part 1)
    const float* R0 = recs + (size_t)b * nu * stride;
    const unsigned st32 = (unsigned)stride, last32 = (unsigned)(nu - 1) * st32;      // (a clip's records span < 2^31 floats)
    const int oc = tid >> 6, ln = tid & 63, cc = ln & 31, hh = ln >> 5;
    // every load of the combine is issued up front, branch-free (indices clamped to a valid unit), so that ONE memory round trip
    // covers them: the scalars of all units (thread (part, f): maxima for part 0, sums for part 1), then the first K^T V blocks
    {
        const int part = tid >> 7, f = tid & 127;
        f32x4 v[L16_MAXU / 4];
#pragma unroll
        for (int k = 0; k < L16_MAXU; ++k) v[k >> 2][k & 3] = R0[min((unsigned)k * st32, last32) + 128 * part + f];
        f32x4* dst = reinterpret_cast<f32x4*>(tsc + (part * 128 + f) * FS);
#pragma unroll
        for (int k = 0; k < L16_MAXU / 4; ++k) dst[k] = v[k];
    }
    auto blk = [&](int k) { return reinterpret_cast<const f32x8*>(R0 + min((unsigned)k * st32, last32) + 256)[oc * 64 + ln]; };
    f32x8 bA[PRE], bB[PRE];                          // units 0..7 and 8..15 are requested up front, 16..23 / 24..31 as these are consumed
#pragma unroll
    for (int k = 0; k < PRE; ++k) bA[k] = blk(k);
#pragma unroll
    for (int k = 0; k < PRE; ++k) bB[k] = blk(PRE + k);
    __syncthreads();
    if (tid < 128) {       // per feature: the units' maxima / sums -> rescale weights and the normaliser
        const int f = tid;
        f32x4 mv[L16_MAXU / 4], sv[L16_MAXU / 4];
#pragma unroll
        for (int k = 0; k < L16_MAXU / 4; ++k) {
            mv[k] = reinterpret_cast<const f32x4*>(tsc + f * FS)[k];
            sv[k] = reinterpret_cast<const f32x4*>(tsc + (128 + f) * FS)[k];
        }
        float mstar = -INFINITY;
#pragma unroll
        for (int k = 0; k < L16_MAXU; ++k)
            if (k < nu && sv[k >> 2][k & 3] > 0.f) mstar = fmaxf(mstar, mv[k >> 2][k & 3]);
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < L16_MAXU; ++k) {
            const float su = k < nu ? sv[k >> 2][k & 3] : 0.f;
            const float ww = su > 0.f ? exp2f_fast(mv[k >> 2][k & 3] - mstar) : 0.f;
            wsc[k * 128 + f] = ww;
            z += ww * su;
        }
        zsc[f] = z;
    }
    __syncthreads();
    const int rowb = 32 * oc + 16 * (cc >> 4) + 4 * hh;       // kept value 4 piece + i <-> K feature (row) rowb + 8 piece + i
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    auto wrow = [&](const float* base, float (&w8)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + rowb), c2 = *reinterpret_cast<const f32x4*>(base + rowb + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w8[j] = a[j];
            w8[4 + j] = c2[j];
        }
    };
    auto consume = [&](const f32x8 (&blkv)[PRE], int k0) {
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            if (k0 + k < nu) {
                float w8[8];
                wrow(wsc + (k0 + k) * 128, w8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(w8[j], blkv[k][j], acc[j]);
            }
    };
    consume(bA, 0);
    if (nu > 2 * PRE) {
#pragma unroll
        for (int k = 0; k < PRE; ++k) bA[k] = blk(2 * PRE + k);
    }
    consume(bB, PRE);
    if (nu > 3 * PRE) {
#pragma unroll
        for (int k = 0; k < PRE; ++k) bB[k] = blk(3 * PRE + k);
    }
    if (nu > 2 * PRE) consume(bA, 2 * PRE);
    if (nu > 3 * PRE) consume(bB, 3 * PRE);
    float z8[8];
    wrow(zsc, z8);
    const int hd = 2 * oc + (cc >> 4), e = hd & 1, l = cc & 15;
#pragma unroll
    for (int piece = 0; piece < 2; ++piece) {
        v4<T16> val, zero;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            val[i] = (T16)(z8[4 * piece + i] > 0.f ? acc[4 * piece + i] * fast_rcp(z8[4 * piece + i]) : 0.f);
            zero[i] = (T16)0.f;
        }
        v4<T16>* dst = reinterpret_cast<v4<T16>*>(af + (size_t)hd * 64 + l + 16 * (2 * piece + hh));      // q4 = 2 piece + hh
        dst[e] = val;
        dst[e ^ 1] = zero;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// The same combine, shared between the clip's workgroups inside the launch (round 5).  With one clip per call the prologue above is
// bound by ONE CU's read rate for data another CU has just written: 29 records x 9 KiB = 261 KiB at 62 - 70 GB/s is 4 of the
// prologue's 6.1 us, and every one of the clip's 29 workgroups pays it to build the same 8-KiB operand.  Here every workgroup reduces
// a 1/32 SLICE of the operand over the clip's records (8 KiB read instead of 261), publishes it as 8-byte {two f16 values, tag}
// granules with sc1 (write-through) stores, and gathers the clip's 1 024 granules with sc1 loads, re-polling until every tag is
// this launch's - the data-tagged hand-off of MI355X_MICROARCH.md (persistent kernels, granule / allgather rows): no flag, no
// counter, no fence, nothing to reset.  A granule that still carries another tag is, within one (B, T) geometry, the previous
// launch's (every launch overwrites all of them; the buffer is cleared whenever the geometry changes), so tags only have to differ
// between consecutive launches.  The clip's workgroups are co-resident by construction (B * upc <= CUs is the launch condition of
// this kernel, one workgroup per CU by its LDS); should the GPU be shared so that they are not, the poll is bounded and raises
// DC_STATUS_SYNC_TIMEOUT instead of hanging.
// Slice s (0..31) = the outputs of threads 8 s .. 8 s + 7 of wg_combine_attn16 (oc = s >> 3, ln = 8 (s & 7) + vt): item (vt, jp) =
// values 2 jp, 2 jp + 1 of virtual thread vt; same products per value as above, summed over the units in a fixed tree (4 per lane in
// unit order, then an xor tree over 8 lanes): deterministic, a clip's operand does not depend on the batch around it.
// Granule gi = (hd * 64 + lane64) * 2 + half: the two f16 values half of the non-zero 8 bytes of AF fragment (hd, lane64).
// ------------------------------------------------------------------------------------------------------------------
constexpr int L16_SLICES = 32;
constexpr unsigned L16_POLL_LIMIT = 1u << 18;

// One slice of the clip's combine: load() issues the slice's 12 loads per lane, publish() reduces and stores the granule.
template <class T16>
struct Slice16 {
    f32x2 mk[4], sk[4], pk[4];
    int kq, oc, cc, hh, jp, nu;
    DEV void load(const float* __restrict__ recs, size_t stride, int nu_, int b, int s, int wave, int lane) {
        // 32 items per slice = 8 per wave; an item's units are spread over 8 lanes (lane = 8 it + kq: units kq, kq + 8, kq + 16, kq + 24),
        // so that the whole wave loads and no lane holds more than four units; the lanes' partial results meet in a fixed xor tree
        nu = nu_;
        kq = lane & 7;
        const int it = lane >> 3, item = 4 * it + wave, vt = item >> 2;
        jp = item & 3;
        oc = s >> 3;
        const int ln = 8 * (s & 7) + vt;
        cc = ln & 31;
        hh = ln >> 5;
        const int rowb = 32 * oc + 16 * (cc >> 4) + 4 * hh;
        const int j = 2 * jp, f = rowb + (j & 3) + 8 * (j >> 2);          // features f, f + 1 (f even)
        const float* R0 = recs + (size_t)b * nu * stride;
        const float* pm = R0 + f;
        const float* pp = R0 + 256 + (size_t)(oc * 64 + ln) * 8 + j;
        const unsigned st32 = (unsigned)stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                       // every load up front, indices clamped (ONE memory round trip)
            const unsigned o = min((unsigned)(kq + 8 * q), (unsigned)(nu - 1)) * st32;
            mk[q] = *reinterpret_cast<const f32x2*>(pm + o);
            sk[q] = *reinterpret_cast<const f32x2*>(pm + 128 + o);
            pk[q] = *reinterpret_cast<const f32x2*>(pp + o);
        }
    }
    DEV void publish(unsigned long long* __restrict__ gran, int b, unsigned tag) {
        auto xmax = [](float v) {
            v = fmaxf(v, __shfl_xor(v, 1));
            v = fmaxf(v, __shfl_xor(v, 2));
            return fmaxf(v, __shfl_xor(v, 4));
        };
        auto xsum = [](float v) {
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            return v + __shfl_xor(v, 4);
        };
        float mstar0 = -INFINITY, mstar1 = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (kq + 8 * q >= nu) sk[q] = (f32x2){0.f, 0.f};
            mstar0 = sk[q].x > 0.f ? fmaxf(mstar0, mk[q].x) : mstar0;
            mstar1 = sk[q].y > 0.f ? fmaxf(mstar1, mk[q].y) : mstar1;
        }
        mstar0 = xmax(mstar0);
        mstar1 = xmax(mstar1);
        float z0 = 0.f, z1 = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float w0 = sk[q].x > 0.f ? exp2f_fast(mk[q].x - mstar0) : 0.f;
            const float w1 = sk[q].y > 0.f ? exp2f_fast(mk[q].y - mstar1) : 0.f;
            z0 = fmaf(w0, sk[q].x, z0);
            z1 = fmaf(w1, sk[q].y, z1);
            a0 = fmaf(w0, pk[q].x, a0);
            a1 = fmaf(w1, pk[q].y, a1);
        }
        z0 = xsum(z0), z1 = xsum(z1), a0 = xsum(a0), a1 = xsum(a1);
        if (kq == 0) {
            typedef __attribute__((ext_vector_type(2))) T16 t2;
            const t2 v = {(T16)(z0 > 0.f ? a0 * fast_rcp(z0) : 0.f), (T16)(z1 > 0.f ? a1 * fast_rcp(z1) : 0.f)};
            const int hd = 2 * oc + (cc >> 4), l = cc & 15, piece = jp >> 1;
            const int gi = ((hd * 64 + l + 16 * (2 * piece + hh)) << 1) + (jp & 1);
            const unsigned long long g = (unsigned long long)__builtin_bit_cast(unsigned, v) | ((unsigned long long)tag << 32);
            __hip_atomic_store(gran + (size_t)b * 1024 + gi, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store_dwordx2 ... sc1: one granule, one store
        }
    }
};
// every thread: granules 4 tid .. 4 tid + 3 = AF fragments 2 tid, 2 tid + 1 of the clip (8 heads x 64 lanes); AF in LDS as above.
// The four loads are issued right behind the publication (gather16_issue) and looked at only where the operand is needed - behind
// LayerNorm, the query projection and the softmax of the self-attention stage (gather16_finish): by then every workgroup of the clip
// has published (they started within a microsecond of each other), and a granule that was read too early is polled again.
struct Gran16 {
    unsigned long long g[4];
};
DEV Gran16 gather16_issue(const unsigned long long* __restrict__ gran, int b, int tid) {
    const unsigned long long* G = gran + (size_t)b * 1024 + 4 * tid;
    Gran16 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = __hip_atomic_load(G + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 ... sc1
    return r;
}
template <class T16>
DEV void gather16_finish(Gran16 r, const unsigned long long* __restrict__ gran, int b, unsigned tag, v8<T16>* af, int tid, int* __restrict__ status) {
    const unsigned long long* G = gran + (size_t)b * 1024 + 4 * tid;
    unsigned spins = 0;
    for (;;) {
        const bool ok = (unsigned)(r.g[0] >> 32) == tag && (unsigned)(r.g[1] >> 32) == tag && (unsigned)(r.g[2] >> 32) == tag &&
                        (unsigned)(r.g[3] >> 32) == tag;
        if (ok || ++spins > L16_POLL_LIMIT) break;
        // another workgroup of this loop has already given up: the loop's results are void whatever arrives now, so the launches
        // that follow must not each wait out the limit again (one timeout per loop, not one per layer launch)
        if ((spins & 63u) == 0u && status && (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & DC_STATUS_SYNC_TIMEOUT)) {
            spins = L16_POLL_LIMIT + 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.g[i] = __hip_atomic_load(G + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (spins > L16_POLL_LIMIT && status) atomicOr(status, DC_STATUS_SYNC_TIMEOUT);
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int fi = 2 * tid + q, hd = fi >> 6, e = hd & 1;
        u4 w = {0u, 0u, 0u, 0u};
        w[2 * e] = (unsigned)r.g[2 * q];
        w[2 * e + 1] = (unsigned)r.g[2 * q + 1];
        reinterpret_cast<u4*>(af)[fi] = w;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// k_layer16: one decoder layer for one 64-token unit of one clip: 4 waves x 16 tokens, one wave per SIMD.  Same stages and
// staging as k_layer (dc_kernels.hip): SA back half -> CA -> FFN -> next layer's SA front half + unit record, or the output
// projection fused with the DDIM update (gaussian_diffusion.py:812-830).  The FiLM tiles of a StylizationBlock are requested at
// the head of the stage in front of it (registers are plentiful at one wave per SIMD).
// ------------------------------------------------------------------------------------------------------------------
template <class T16, bool G1 = false /* FiLM scale tiles hold G' (film_affine, dc_dev.h) */>
__global__ __launch_bounds__(256, 2)      // (one workgroup per CU by its LDS; "2" keeps the compiler off the AGPR half: 250 VGPRs, no accvgpr moves)
void k_layer16(const DcModel* __restrict__ dm, int l, float* __restrict__ hbuf, const f16x16* __restrict__ E, int NT,
               const v8<T16>* __restrict__ a_ca /*[L][B][8 heads][64] 16-token form*/, float* __restrict__ recs, const int* __restrict__ length,
               const float* __restrict__ xin, float* __restrict__ xout, int out_mode, const float* __restrict__ coef_cur,
               const int* __restrict__ snap_cur, float* __restrict__ snaps, int M, int T, int B, int upc,
               size_t rec_stride /* floats between the two alternating record buffers */,
               int nu_in, size_t stride_in /* unit records per clip / floats per unit of the records this layer combines */,
               const int* __restrict__ iter_base, int Tx, const DcUpdate upd,
               unsigned long long* __restrict__ gran /* [B][1024] granules of the shared combine, or nullptr: every workgroup combines alone */,
               unsigned tag_base /* + 16 * (*iter_base) + l + 1 = this launch's tag */) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using W = v8<T16>;
    constexpr int NW = L16_NW;
#ifdef DC_L16_STAMPS      // diagnostic build (tools/stage_stamps16.py): s_memrealtime of workgroup 3's waves at the stage boundaries
#define LSTAMP(k)                                                                                                          \
    do {                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if (upd.stamps && blockIdx.x == 3 && (threadIdx.x & 63) == 0) upd.stamps[(threadIdx.x >> 6) * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
#else
#define LSTAMP(k) do {} while (0)
#endif
    // (k_layer's one-round-trip kernel-argument trick measured 1 % SLOWER here - hipcc splits the batch in two around an SGPR reuse - and is not used)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wg = wg_index();
    C16 c;
    c.b = wg / upc;
    const int u = wg - c.b * upc;
    c.n0 = 64 * u + 16 * wave;
    const bool active = c.n0 < T;                // idle waves (the clip's last unit) still take part in the staging and barriers
    if (!active) c.n0 = T - 16;
    c.first = c.b * T + c.n0;
    c.g = c.first >> 5;
    c.half = (c.n0 >> 4) & 1;
    c.lane = lane;
    c.n = lane & 15;
    c.q4 = lane >> 4;
    c.tok = c.first + c.n;
    const int nact = min(NW, (T - 64 * u + 15) >> 4);       // active waves of the workgroup (a prefix)
    char* buf0 = lds;
    char* buf1 = lds + L16_WSZ;
    const W* w0 = reinterpret_cast<const W*>(buf0);
    const W* w1 = reinterpret_cast<const W*>(buf1);
    const float* c0 = reinterpret_cast<const float*>(buf0 + 32 * 1024);      // constants block behind the 32 fragments
    const float* c1 = reinterpret_cast<const float*>(buf1 + 32 * 1024);
    const W* af = reinterpret_cast<const W*>(lds + L16_OFF_AF);
    const DcLayer16& L = dm->l16[l];
    const int nl = dm->num_layers;
    const bool last = l + 1 >= nl;
    const f16x8* Eg = reinterpret_cast<const f16x8*>(E) + ((size_t)c.g * NT + (size_t)l * 24) * 128;   // 3 blocks x 8 tiles
    const float* recs_in = recs + (size_t)(l & 1) * rec_stride;
    float* recs_out = recs + (size_t)((l + 1) & 1) * rec_stride;

    f32x4 h[8];
    LSTAMP(0);
    const unsigned tag = tag_base + (iter_base ? 16u * (unsigned)*iter_base : 0u) + (unsigned)l + 1u;
    load_h16(h, hbuf, c);
    stage_frags<NW>(L.sa_q, buf0, 33, wave, lane);
    E16 e[4];
    Gran16 gr{};
    if (gran) {       // (wave-uniform) this workgroup's slices of the clip's combine, published now, gathered where the operand is needed (stage 1)
        // (the slice's loads in FRONT of the residual-stream loads measured no better: profiles/r05_small_batch_shared_combine.txt)
        Slice16<T16> sl;
        const bool drop = (upd.flags & DC_UPD_TEST_DROP_SLICE) && wg == 0;      // (test hook: the timeout path)
        for (int s2 = u; s2 < L16_SLICES && !drop; s2 += upc) {      // (one slice; a second for the first few workgroups at T = 1800; more for short clips)
            sl.load(recs_in, stride_in, nu_in, c.b, s2, wave, lane);
            sl.publish(gran, c.b, tag);
        }
        LSTAMP(20);
        gr = gather16_issue(gran, c.b, tid);
    } else
        wg_combine_attn16<T16>(recs_in, stride_in, nu_in, c.b, reinterpret_cast<W*>(lds + L16_OFF_AF), reinterpret_cast<float*>(buf1),
                               reinterpret_cast<float*>(lds + L16_OFF_PST) /* the tail's 36 KiB: free until then */, tid);
    e16_load(e, Eg, c);                          // (behind the combine, whose record batches need the registers; first used in stage 2)
    LSTAMP(1);
    // closer that leaves the 16 FiLM-tile loads just issued in flight (they are the wave's youngest vector-memory operations; the
    // compiler waits for them where stage 2 first uses them): everything older - h, the records, the query image - has landed
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    LSTAMP(2);

    // ---- self-attention
    stage_frags<NW>(L.sa_o, buf1, 33, wave, lane);
    Y16 y[8];
    float y_rstd, y_shift;
    query_attend16<T16>(y, y_rstd, y_shift, h, c0, w0, af, c, [&]() {
        if (gran) {
            LSTAMP(21);
            gather16_finish<T16>(gr, gran, c.b, tag, reinterpret_cast<W*>(lds + L16_OFF_AF), tid, upd.status);
            __syncthreads();          // the operand is written by all four waves
            LSTAMP(22);
        }
    });
    LSTAMP(3);
    stage_sync();
    LSTAMP(4);
    {   // cross-attention query image + the clip's cross-attention fragments (16-token form, k_cond_af16)
        stage_frags<NW>(L.ca_q, buf0, 33, wave, lane);
        stage_frags<NW>(a_ca + ((size_t)l * B + c.b) * 8 * 64, lds + L16_OFF_AF, 8, wave, lane);
    }
    styl_accumulate16<T16, G1>(h, y, y_rstd, y_shift, e, c1, w1, c);
    LSTAMP(5);
    stage_sync();
    LSTAMP(6);
    // ---- cross-attention
    stage_frags<NW>(L.ca_o, buf1, 33, wave, lane);
    e16_load(e, Eg + 8 * 128, c);
    query_attend16<T16>(y, y_rstd, y_shift, h, c0, w0, af, c);
    LSTAMP(7);
    stage_sync();
    LSTAMP(8);
    stage_frags<NW>(L.ffn_w, buf0, 33, wave, lane);          // W1 (16 fragments) | W2 (16) | b1[64], b2[128]
    styl_accumulate16<T16, G1>(h, y, y_rstd, y_shift, e, c1, w1, c);
    LSTAMP(9);
    stage_sync();
    LSTAMP(10);
    // ---- FFN
    stage_frags<NW>(L.ffn_o, buf1, 33, wave, lane);
    e16_load(e, Eg + 16 * 128, c);
    {
        f32x4 uu[4];
        {
            W hb[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) hb[m] = frag2<T16>(h[2 * m], h[2 * m + 1]);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) uu[rb] = *reinterpret_cast<const f32x4*>(c0 + 16 * rb + 4 * c.q4);            // b1
            gemm16<4, 4, T16>(uu, w0, hb, lane);
        }
        W ub[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 ga, gb;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x2 a = gelu_erf_pair(uu[2 * m][2 * p], uu[2 * m][2 * p + 1]), b2 = gelu_erf_pair(uu[2 * m + 1][2 * p], uu[2 * m + 1][2 * p + 1]);
                ga[2 * p] = a.x;
                ga[2 * p + 1] = a.y;
                gb[2 * p] = b2.x;
                gb[2 * p + 1] = b2.y;
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
    LSTAMP(11);
    stage_sync();
    LSTAMP(12);
    if (!last)
        stage_frags<NW>(dm->l16[l + 1].sa_k, buf0, 33, wave, lane);
    else
        stage_frags<NW>(dm->out16, buf0, 17, wave, lane);        // 8 hi + 8 lo fragments + bias: always runs split
    styl_accumulate16<T16, G1>(h, y, y_rstd, y_shift, e, c1, w1, c);
    LSTAMP(13);
    stage_sync();
    LSTAMP(14);

    if (!last) {
        // ---- next layer's self-attention front half (transformer.py:104-117): K [buf0], V [buf1] in TF form (token on the ROW:
        // lane = feature), unit record: column maxima over the unit first, so that every wave exponentiates against the unit's
        // maximum side by side and the K^T V blocks are summed over the waves in wave order (deterministic)
        stage_frags<NW>(dm->l16[l + 1].sa_v, buf1, 33, wave, lane);
        if (active) store_h16(h, hbuf, c);
        W nb[4];
        ln16_frags<T16>(nb, h);
        float* mx = reinterpret_cast<float*>(lds + L16_OFF_AF);               // [128 features][4 waves]
        f32x4* pst = reinterpret_cast<f32x4*>(lds + L16_OFF_PST);              // [4 waves][8 blocks][64 lanes]
        float* ss = reinterpret_cast<float*>(lds + L16_OFF_SS);                // [4 waves][128]
        float* scw = reinterpret_cast<float*>(lds + L16_OFF_SCW) + wave * 128; // this wave's rescale factors [128]
        const int f = lane & 15;                                               // TF form: this lane's feature inside a block
        const int len = min(length[c.b], T);
        bool ok[4];                                                            // validity of this lane's four token rows 4 q4 + i
#pragma unroll
        for (int i = 0; i < 4; ++i) ok[i] = active && c.n0 + 4 * c.q4 + i < len;
        v4<T16> ef[8];
        float ssw[8], mw[8];
        auto col_frags = [&](W (&fr)[4], const W* w, int cb) {      // the four k-steps of output block cb
#pragma unroll
            for (int m = 0; m < 4; ++m) fr[m] = w[(m * 8 + cb) * 64 + lane];
        };
        W kc[4], kn[4];
        col_frags(kc, w0, 0);
        // software pipeline (round 5, bit-identical, -1.3 % per loop at one clip): block cb + 1's MFMA chain is issued in front of block
        // cb's vector work (maximum, exponentials, sums) - at one wave per SIMD nobody else fills the matrix pipe's latency
        auto kchain = [&](int cb, const W (&fr)[4]) {
            const float bk = c0[16 * cb + f];
            f32x4 K = {bk, bk, bk, bk};
#pragma unroll
            for (int m = 0; m < 4; ++m) K = mfma16(nb[m], fr[m], K);
            return K;
        };
        f32x4 Kc = kchain(0, kc), Kn = Kc;
        col_frags(kn, w0, 1);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            __builtin_amdgcn_sched_barrier(0);
            if (cb + 1 < 8) Kn = kchain(cb + 1, kn);
            if (cb + 2 < 8) col_frags(kn, w0, cb + 2);
            const f32x4 K = Kc;
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i) m = ok[i] ? fmaxf(m, K[i]) : m;
            m = xq_max(m);
            mw[cb] = m;
            const float mz = m == -INFINITY ? 0.f : m;
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ex = ok[i] ? exp2f_fast(K[i] - mz) : 0.f;
                sacc += ex;
                ef[cb][i] = (T16)ex;
            }
            ssw[cb] = xq_sum(sacc);
            if (c.q4 == 0) mx[(16 * cb + f) * 4 + wave] = m;
            Kc = Kn;
        }
        __builtin_amdgcn_sched_barrier(0);
        LSTAMP(15);
        if (active)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // all but the 8 stores of h: the value image has landed
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        LSTAMP(16);
        auto unit_max = [&](int feat) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(mx + feat * 4);
            const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
            return m == -INFINITY ? 0.f : m;
        };
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const float fa = mw[cb] == -INFINITY ? 0.f : exp2f_fast(mw[cb] - unit_max(16 * cb + f));
            if (c.q4 == 0) {
                scw[16 * cb + f] = fa;
                ss[wave * 128 + 16 * cb + f] = ssw[cb] * fa;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        col_frags(kc, w1, 0);
        auto vchain = [&](int cb, const W (&fr)[4]) {
            const float bv = c1[16 * cb + f];
            f32x4 V = {bv, bv, bv, bv};
#pragma unroll
            for (int m = 0; m < 4; ++m) V = mfma16(nb[m], fr[m], V);
            return V;
        };
        f32x4 Vc = vchain(0, kc), Vn = Vc;
        col_frags(kn, w1, 1);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            __builtin_amdgcn_sched_barrier(0);
            if (cb + 1 < 8) Vn = vchain(cb + 1, kn);
            if (cb + 2 < 8) col_frags(kn, w1, cb + 2);
            f32x4 va;
#pragma unroll
            for (int i = 0; i < 4; ++i) va[i] = ok[i] ? Vc[i] : 0.f;
            W ea;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ea[j] = ef[cb][j];
                ea[4 + j] = (T16)0.f;
            }
            f32x4 PA = mfma16(ea, frag1<T16>(va), z4);
            PA *= *reinterpret_cast<const f32x4*>(scw + 16 * cb + 4 * c.q4);
            pst[(size_t)(wave * 8 + cb) * 64 + lane] = PA;
            Vc = Vn;
        }
        __builtin_amdgcn_sched_barrier(0);
        LSTAMP(17);
        __syncthreads();
        LSTAMP(18);
        // wave w sums blocks 2 w, 2 w + 1 over the unit's waves in wave order and writes them in the 32-token kernels' record format
        float* R = recs_out + (size_t)wg * DC_REC_FLOATS;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int cb = 2 * wave + k;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float ssum = 0.f;
#pragma unroll
            for (int v = 0; v < NW; ++v)
                if (v < nact) {
                    acc += pst[(size_t)(v * 8 + cb) * 64 + lane];
                    ssum += ss[v * 128 + 16 * cb + f];
                }
            if (c.q4 == 0) {
                R[16 * cb + f] = unit_max(16 * cb + f);
                R[128 + 16 * cb + f] = ssum;
            }
            // lane (l = lane & 15, q4) holds P[d = 4 q4 + i][l]: the 32-token record keeps it in tile oc = cb >> 1, old lane
            // 16 (cb & 1) + l + 32 (q4 & 1), values 4 (q4 >> 1) + i
            reinterpret_cast<f32x4*>(R + 256 + ((cb >> 1) * 64 + 16 * (cb & 1) + f + 32 * (c.q4 & 1)) * 8)[c.q4 >> 1] = acc;
        }
        LSTAMP(19);
        return;
    }
    // ---- output projection (transformer.py:496) [buf0: 8 hi + 8 lo fragments, bias behind them] + DDIM update
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
                const T16 a = (T16)h[2 * m][j], b2 = (T16)h[2 * m + 1][j];
                hi[j] = a;
                hi[4 + j] = b2;
                la[j] = h[2 * m][j] - (float)a;
                lb[j] = h[2 * m + 1][j] - (float)b2;
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
    const int xn = c.n0 + c.n;                                       // this lane's frame inside the clip
    if (!active || xn >= Tx) return;                                 // idle wave / padding frame
    const int P = dm->input_feats;
    const size_t xrow = (size_t)c.b * Tx + xn;                       // row of xin / xout / snaps
    bool bad = false;
    if (out_mode == 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ft = 16 * rb + 4 * c.q4 + i;
                if (ft < P) {
                    xout[xrow * P + ft] = x0[rb][i];
                    bad = bad || !(fabsf(x0[rb][i]) <= 3.0e38f);
                }
            }
    } else {
        // graph-captured loop: coef_cur / snap_cur point at this step's slot of the per-iteration tables and *iter_base is the
        // iteration at which the graph replay began; otherwise they are the scalars k_begin_step prepared
        const int ib = iter_base ? *iter_base : 0;
        coef_cur += DC_COEF * ib;
        const int snap = snap_cur[ib];
        const bool noisy = (upd.flags & DC_UPD_NOISY) != 0;
        const float* zrow = nullptr;
        if (noisy) zrow = *upd.zslot + ((upd.flags & DC_UPD_ZSTEP) ? (size_t)0 : (size_t)(iter_base ? upd.step + ib : snap_cur[1]) * B * Tx * P);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ft = 16 * rb + 4 * c.q4 + i;
                if (ft < P) {
                    const size_t o = xrow * P + ft;
                    const float xnew = ddim_update(x0[rb][i], xin[o], coef_cur, upd.flags, noisy, noisy ? zrow[o] : 0.f, bad);
                    xout[o] = xnew;
                    if (snap >= 0) snaps[(size_t)snap * B * Tx * P + o] = xnew;
                }
            }
    }
    if (bad && upd.status) atomicOr(upd.status, DC_STATUS_NONFINITE);
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

int dc_layer16_max_units(void) { return L16_MAXU; }

namespace {
template <class T16, bool G1>
hipError_t launch_layer16_t(hipStream_t st, const DcModel* dm, int l, float* hbuf, const void* E, int NT, const void* a_ca16,
                            float* recs, const int* length, const float* xin, float* xout, int out_mode, const float* coef_cur,
                            const int* snap_cur, float* snaps, int M, int T, int B, int upc, size_t rec_stride, int nu_in, size_t stride_in,
                            const int* iter_base, int Tx, const DcUpdate& upd, unsigned long long* gran, unsigned tag_base) {
    static unsigned long long done = 0;      // > 64 KiB of dynamic LDS needs the opt-in, per device
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev >= 64 || !((done >> dev) & 1ull)) {
        if (hipError_t e = hipFuncSetAttribute((const void*)k_layer16<T16, G1>, hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS)) return e;
        if (dev < 64) done |= 1ull << dev;
    }
    k_layer16<T16, G1><<<dim3(B * upc), dim3(256), L16_LDS, st>>>(dm, l, hbuf, (const f16x16*)E, NT, (const v8<T16>*)a_ca16, recs, length, xin, xout,
                                                                  out_mode, coef_cur, snap_cur, snaps, M, T, B, upc, rec_stride, nu_in, stride_in,
                                                                  iter_base, Tx, upd, gran, tag_base);
    return hipGetLastError();
}
}  // namespace

hipError_t dc_launch_layer16(hipStream_t st, int fmt, const DcModel* dm, int l, float* hbuf, const void* E, int NT, const void* a_ca16,
                             float* recs, const int* length, const float* xin, float* xout, int out_mode, const float* coef_cur,
                             const int* snap_cur, float* snaps, int M, int T, int B, int upc, size_t rec_stride, int nu_in, size_t stride_in,
                             const int* iter_base, int Tx, const DcUpdate& upd, unsigned long long* gran, unsigned tag_base, bool g1) {
    if (nu_in < 1 || nu_in > L16_MAXU || upc < 1 || (T & 31)) return hipErrorInvalidValue;
#define L16_ARGS st, dm, l, hbuf, E, NT, a_ca16, recs, length, xin, xout, out_mode, coef_cur, snap_cur, snaps, M, T, B, upc, rec_stride, nu_in, stride_in, \
                 iter_base, Tx, upd, gran, tag_base
    if (fmt == 1) return g1 ? launch_layer16_t<_Float16, true>(L16_ARGS) : launch_layer16_t<_Float16, false>(L16_ARGS);
    return g1 ? launch_layer16_t<__bf16, true>(L16_ARGS) : launch_layer16_t<__bf16, false>(L16_ARGS);
#undef L16_ARGS
}
hipError_t dc_launch_cond_af16(hipStream_t st, int fmt, const void* a_ca, void* a_ca16, int n_matrices) {
    const int n = n_matrices * 8 * 64;
    if (fmt == 1)
        k_cond_af16<_Float16><<<dim3((n + 255) / 256), dim3(256), 0, st>>>((const f16x8*)a_ca, (f16x8*)a_ca16, n_matrices);
    else
        k_cond_af16<__bf16><<<dim3((n + 255) / 256), dim3(256), 0, st>>>((const bf16x8*)a_ca, (bf16x8*)a_ca16, n_matrices);
    return hipGetLastError();
}
