// dc_fused.hip - the fused decoder-layer kernel (non-split operand formats, T >= 128): one launch per layer computes the
// layer's three FiLM blocks (StylizationBlock.emb_layers, transformer.py:57-60,74-78 - 74 % of the step's FLOPs) INSIDE the
// layer, so the 708 MB per step of FiLM tiles that the separate GEMM wrote and k_layer read back never exist.
//
// Shape of the kernel:
//   * workgroup = 4 waves x 32 tokens = one 128-token UNIT; 80 KiB of LDS and <= 256 VGPRs, so two workgroups share a CU and
//     the two waves of a SIMD belong to DIFFERENT workgroups: while one sits in a FiLM block (pure MFMA + LDS reads) the
//     other is in the attention / stylization chain (VALU heavy) - the matrix pipe and the VALU overlap without any
//     software pipelining inside a wave;
//   * every weight of the layer - 3 x 256 FiLM fragments and the eleven 128-wide projections - is ONE linear stream of
//     16-KiB chunks (16 MFMA fragments, in consumption order, packed by the host) that LDS-DMA moves L2 -> a 3-slot LDS ring;
//     one barrier per chunk; a chunk serves all four waves (each weight fragment read from LDS feeds one MFMA per wave:
//     half of the LDS's 256 B/clk at full MFMA rate);
//   * the FiLM operand S = SiLU(emb) of the wave's own 32 tokens comes straight from global memory (f16 fragment image,
//     59 MB per step, resident in the 256-MB Infinity Cache), two fragments per 32-deep k-tile, two k-tiles ahead;
//   * a FiLM block = 8 accumulator tiles (128 registers) over 16 k-tiles; the result is held as packed f16 (64 registers,
//     exactly the tiles the separate GEMM used to store) while the block's attention / FFN half runs;
//   * the linear attention's cross-token reduction works per unit: the tail of layer l writes one record per unit and clip
//     slot, the prologue of layer l+1 combines the <= 32 unit records of the workgroup's <= 2 clips (fixed order:
//     re-runs are bit-identical).
// vmcnt bookkeeping: per ring iteration a wave issues [extras][2 S loads][4 chunk DMAs]; chunk c's DMAs are the last
// operations of iteration c-2, so `s_waitcnt vmcnt(#operations of iteration c-1)` at the top of iteration c means "chunk c
// and everything older has landed".  Operations the compiler adds on its own only make that wait stricter.
#include "dc_dev.h"
#include "dc_launch.h"

namespace {

constexpr int FNW = 4;                      // waves per workgroup
constexpr int FUT = 32 * FNW;               // tokens per unit
constexpr int F_SLOT = 16384;               // ring slot = one chunk = 16 fragments
constexpr int F_OFF_AF = 3 * F_SLOT;        // attention fragments of the unit's <= 2 clips (16 KiB); tail: xp | rescale strips
constexpr int F_OFF_CONST = F_OFF_AF + 16384;   // the layer's constants (8 KiB, dc_common.h DCF_C_*)
constexpr int F_OFF_SCR = F_OFF_CONST + 8192;   // prologue: m* | z of the combine; tail: column maxima | column sums
constexpr int F_LDS = F_OFF_SCR + 8192;         // 80 KiB: two workgroups per CU
constexpr int F_NU = 32;                    // units per clip the combine's LDS weights hold (T <= ~4000)

DEV char* ring_slot(char* lds, int c) { return lds + (c % 3) * F_SLOT; }
DEV void ring_issue(const void* stream, int c, char* lds, int wave, int lane) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(stream) + (size_t)c * 16 * 64 + lane;
    char* dst = ring_slot(lds, c);
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_dma16(src + (size_t)(wave * 4 + i) * 64, dst + (wave * 4 + i) * 1024);
}
template <int N>
DEV void wait_vm() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
DEV void ring_bar() {
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
DEV void ring_wait() {
    wait_vm<N>();
    ring_bar();
}

// uniform base in SGPRs + per-lane byte offset in one VGPR: distinct fragments cost scalar adds, not VGPR address pairs.
// SAFE = false: no wait is generated at the use (the ring waits cover it) - the destination must stay live until then.
template <bool SAFE, class T16>
DEV v8<T16> s_load(const v8<T16>* sbase, unsigned voff) {
    if constexpr (SAFE) {
        return *reinterpret_cast<const v8<T16>*>(reinterpret_cast<const char*>(sbase) + voff);
    } else {
        v8<T16> v;
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
        return v;
    }
}

DEV float wg_colmax4(const float* mx, int oc, int sl, int c) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(mx + ((oc * 2 + sl) * 32 + c) * 4);
    const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
    return m == -INFINITY ? 0.f : m;
}

typedef __attribute__((ext_vector_type(2))) _Float16 h2v;
DEV f16x16 pack_tile(const f32x16& x) {
    u32x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const h2v p = {(_Float16)x[2 * k], (_Float16)x[2 * k + 1]};
        o[k] = __builtin_bit_cast(uint32_t, p);
    }
    return __builtin_bit_cast(f16x16, o);
}

// ------------------------------------------------------------------------------------------------------------------
// Combine of the unit records (prologue of a layer): attention operand fragments A[d][l] of clips ub0, ub0+1 from the
// records the previous kernel wrote -> af [2 clips][8 frags][64 lanes] in LDS.  256 threads.
//   A[d][l] = sum_u w_u[d] P_u[d][l] / sum_u w_u[d] s_u[d],   w_u = exp2(m_u - max_u m_u)     (fixed summation order)
// wsc (LDS): weights [2][F_NU][128]; zsc (LDS): normalisers [2][128].
// ------------------------------------------------------------------------------------------------------------------
template <class T16>
DEV void wg_combine4(const float* __restrict__ recs, v8<T16>* af, float* wsc, float* zsc, int ub0, int M, int T, int tid, int wg) {
    constexpr int PRE = 16;
    const int ub1 = (min((wg + 1) * FUT, M) - 1) / T;          // last clip this unit touches
    auto rec_of = [&](int clip, int u) { return recs + ((size_t)u * 2 + ((u * FUT >= clip * T) ? 0 : 1)) * DC_REC_FLOATS; };
    {   // phase A: thread = (clip ca, feature f): m*, weights, normaliser
        const int ca = tid >> 7, f = tid & 127, ba = ub0 + ca;
        const bool la = ba <= ub1;
        const int bav = la ? ba : ub0;
        const int a_lo = (bav * T) / FUT, a_hi = (min((bav + 1) * T, M) - 1) / FUT;
        const int na = la ? a_hi - a_lo + 1 : 0;
        float mr[PRE], sr[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) {                        // branch-free: indices clamped, results predicated
            const float* R = rec_of(bav, min(a_lo + k, a_hi));
            mr[k] = R[f];
            sr[k] = R[128 + f];
        }
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            if (k >= na) sr[k] = 0.f;
        float mstar = -INFINITY;
#pragma unroll
        for (int k = 0; k < PRE; ++k)
            if (sr[k] > 0.f) mstar = fmaxf(mstar, mr[k]);
        for (int k = PRE; k < na; ++k) {                       // clips longer than 16 units (T > 1920)
            const float* R = rec_of(ba, a_lo + k);
            if (R[128 + f] > 0.f) mstar = fmaxf(mstar, R[f]);
        }
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const float ww = sr[k] > 0.f ? exp2f_fast(mr[k] - mstar) : 0.f;
            if (k < na) wsc[(ca * F_NU + k) * 128 + f] = ww;
            z += ww * sr[k];
        }
        for (int k = PRE; k < na; ++k) {
            const float* R = rec_of(ba, a_lo + k);
            const float su = R[128 + f];
            const float ww = su > 0.f ? exp2f_fast(R[f] - mstar) : 0.f;
            wsc[(ca * F_NU + k) * 128 + f] = ww;
            z += ww * su;
        }
        zsc[ca * 128 + f] = z;
    }
    __syncthreads();
    // phase B: thread = (feature tile oc, lane ln) for both clips in turn; the K^T V blocks in batches of 8 units
    const int oc = tid >> 6, ln = tid & 63, c = ln & 31, hh = ln >> 5;
    const int rowb = 32 * oc + 16 * (c >> 4) + 4 * hh;        // kept value j <-> feature row rowb + (j&3) + 8*(j>>2)
    auto wrow = [&](const float* base, float (&w8)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + rowb), c2 = *reinterpret_cast<const f32x4*>(base + rowb + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w8[j] = a[j];
            w8[4 + j] = c2[j];
        }
    };
#pragma unroll 1
    for (int ci = 0; ci < 2; ++ci) {
        const int b = ub0 + ci;
        const bool live = b <= ub1;
        const int bv = live ? b : ub0;
        const int u_lo = (bv * T) / FUT, u_hi = (min((bv + 1) * T, M) - 1) / FUT;
        const int nu = live ? u_hi - u_lo + 1 : 0;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int k0 = 0; k0 < nu; k0 += 8) {
            f32x8 pre[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pre[k] = reinterpret_cast<const f32x8*>(rec_of(bv, min(u_lo + k0 + k, u_hi)) + 256)[oc * 64 + ln];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < nu) {
                    float w8[8];
                    wrow(wsc + (ci * F_NU + k0 + k) * 128, w8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(w8[j], pre[k][j], acc[j]);
                }
        }
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
    }
}

// ------------------------------------------------------------------------------------------------------------------
// "Front half" of LinearTemporalSelfAttention for the NEXT layer (transformer.py:104-117), shared by the layer kernel's
// tail and the step prologue: K = Wk n + bk, V = Wv n + bv (TF form) of the unit's 128 tokens, softmax over the sequence
// of K against the UNIT's column maxima, exp(K - m)^T V, and one record per unit and clip slot.
// Stream chunks cK, cK+1 = key image halves (k-tiles 0,1 | 2,3), cK+2, cK+3 = value image halves; on entry chunks cK and
// cK+1 are in flight and the wave has issued NPREV operations since cK's DMAs.  Ring slots (cK+1)%3 and (cK+2)%3 are free
// once every wave has finished the first value half: the four waves' K^T V blocks are staged there for the ordered sum.
// ------------------------------------------------------------------------------------------------------------------
template <class T16, int NPREV>
DEV void front_tail(const f32x16 (&h)[4], bool st_h, float* __restrict__ hbuf, char* lds, const void* stream, int cK, const float* bk,
                    const float* bv, const GroupCtx& cx, bool active, int wave, int lane, int wg, int ub0, int B, int M, int T, int G,
                    const int* __restrict__ length, float* __restrict__ recs_out) {
    using W = v8<T16>;
    float* mx = reinterpret_cast<float*>(lds + F_OFF_SCR);                  // [4 oc][2 slots][32 cols][4 waves]
    float* ss = reinterpret_cast<float*>(lds + F_OFF_SCR + 4096);           // [(4 + 1)][4 oc][32]
    f32x8* xp = reinterpret_cast<f32x8*>(lds + F_OFF_AF);                   // second slot of the straddling wave
    float* scw = reinterpret_cast<float*>(lds + F_OFF_AF + 8192) + wave * 2 * 4 * 32;   // this wave's rescale factors
    // ---- iteration kA
    ring_wait<NPREV>();
    // h goes out here (16 stores, ahead of this iteration's chunk DMA): they drain behind the K/V projections
    if (st_h) store_h(h, hbuf, cx.g, lane);
    __builtin_amdgcn_sched_barrier(0);
    ring_issue(stream, cK + 2, lds, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    XFrag<T16, false> nf[4];
    ln_frags<T16, false>(nf, h);
    const RowRange vr0 = valid_rows_clip(cx, ub0, B, M, T, length, active);
    const RowRange vr1 = valid_rows_clip(cx, ub0 + 1, B, M, T, length, active);
    const int s0 = cx.b0 - ub0;
    const RowRange vr_own = s0 ? vr1 : vr0;
    const bool strad = active && cx.straddle;
    XFrag<T16, false> efA[4], efB[4];
    float ssA[4], ssB[4], mA[4], mB[4];
    auto keys_of = [&](const f32x16& K, const RowRange& rr, XFrag<T16, false>& ef, float& ssum, float& mcol) {
        float m = -INFINITY;
        const bool full = __builtin_amdgcn_readfirstlane(rr.span) == 32u;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, K[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) m = row_ok(rr, r) ? fmaxf(m, K[r]) : m;
        }
        m = xhalf_max(m);
        mcol = m;                                               // -inf: no valid row of this slot in the wave
        const float mz = m == -INFINITY ? 0.f : m;
        f32x16 Ee;
        float sacc = 0.f;
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Ee[r] = exp2f_fast(K[r] - mz);
                sacc += Ee[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Ee[r] = row_ok(rr, r) ? exp2f_fast(K[r] - mz) : 0.f;
                sacc += Ee[r];
            }
        }
        ssum = xhalf_sum(sacc);
        make_frag<T16, false>(Ee, ef);
    };
    auto quad_half = [&](f32x16 (&P)[4], const W* w, int kt0) {          // acc[oc] += X^T W[oc] over k-tiles kt0, kt0+1 (one chunk)
#pragma unroll
        for (int kl = 0; kl < 2; ++kl)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int oc = 0; oc < 4; ++oc) P[oc] = mfma(nf[kt0 + kl].hi[s], w[((kl * 4 + oc) * 2 + s) * 64 + lane], P[oc]);
    };
    {
        f32x16 Kp[4] = {splat(bk[cx.c]), splat(bk[32 + cx.c]), splat(bk[64 + cx.c]), splat(bk[96 + cx.c])};
        quad_half(Kp, reinterpret_cast<const W*>(ring_slot(lds, cK)), 0);
        // ---- iteration kB (previous iteration: 4 DMAs + the 16 stores of h)
        if (st_h)
            wait_vm<20>();
        else
            wait_vm<4>();
        ring_bar();
        ring_issue(stream, cK + 3, lds, wave, lane);
        quad_half(Kp, reinterpret_cast<const W*>(ring_slot(lds, cK + 1)), 2);
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            keys_of(Kp[oc], vr_own, efA[oc], ssA[oc], mA[oc]);
            mB[oc] = -INFINITY;
            if (strad) keys_of(Kp[oc], vr1, efB[oc], ssB[oc], mB[oc]);
            if (cx.hh == 0) {
                mx[((oc * 2 + s0) * 32 + cx.c) * 4 + wave] = mA[oc];
                mx[((oc * 2 + (s0 ^ 1)) * 32 + cx.c) * 4 + wave] = s0 ? -INFINITY : mB[oc];
            }
        }
    }
    // ---- iteration vA: the maxima of all waves are visible behind its barrier
    ring_wait<4>();
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
        const float fa = mA[oc] == -INFINITY ? 0.f : exp2f_fast(mA[oc] - wg_colmax4(mx, oc, s0, cx.c));
        ssA[oc] *= fa;
        if (cx.hh == 0) scw[(0 * 4 + oc) * 32 + cx.c] = fa;
        if (strad) {
            const float fb = mB[oc] == -INFINITY ? 0.f : exp2f_fast(mB[oc] - wg_colmax4(mx, oc, 1, cx.c));
            ssB[oc] *= fb;
            if (cx.hh == 0) scw[(1 * 4 + oc) * 32 + cx.c] = fb;
        }
    }
    f32x16 Vp[4] = {splat(bv[cx.c]), splat(bv[32 + cx.c]), splat(bv[64 + cx.c]), splat(bv[96 + cx.c])};
    quad_half(Vp, reinterpret_cast<const W*>(ring_slot(lds, cK + 2)), 0);
    // ---- iteration vB: nothing was issued in vA; behind this barrier slots (cK+1)%3 and (cK+2)%3 are free
    ring_wait<0>();
    quad_half(Vp, reinterpret_cast<const W*>(ring_slot(lds, cK + 3)), 2);
    const int rowq = 16 * (cx.c >> 4) + 4 * cx.hh;            // kept value j <-> column (row of P) rowq + (j&3) + 8(j>>2)
    auto block_of = [&](const XFrag<T16, false>& ef, const f32x16& V, const RowRange& rr, const float* sc) {
        f32x16 Vm;
        if (__builtin_amdgcn_readfirstlane(rr.span) == 32u) {
            Vm = V;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) Vm[r] = row_ok(rr, r) ? V[r] : 0.f;
        }
        XFrag<T16, false> vf;
        make_frag<T16, false>(Vm, vf);
        f32x16 P = splat(0.f);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) P = mfma(ef.hi[s2], vf.hi[s2], P);
        f32x8 keep = keep_head_block(P, cx.c);
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(sc + rowq), f1 = *reinterpret_cast<const f32x4*>(sc + rowq + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            keep[i] *= f0[i];
            keep[4 + i] *= f1[i];
        }
        return keep;
    };
    auto pst_of = [&](int v) { return reinterpret_cast<f32x8*>(ring_slot(lds, cK + 1 + (v >> 1)) + (v & 1) * 8192); };   // [4 oc][64]
    {
        f32x8* pst = pst_of(wave);
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            pst[oc * 64 + lane] = block_of(efA[oc], Vp[oc], vr_own, scw + (0 * 4 + oc) * 32);
            if (cx.hh == 0) ss[(wave * 4 + oc) * 32 + cx.c] = ssA[oc];
            if (strad) {
                xp[oc * 64 + lane] = block_of(efB[oc], Vp[oc], vr1, scw + (1 * 4 + oc) * 32);
                if (cx.hh == 0) ss[(4 * 4 + oc) * 32 + cx.c] = ssB[oc];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // wave w sums feature tile oc = w of both slots over the four waves, in wave order
    {
        const int oc = wave, c = lane & 31;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            f32x8 acc;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            float ssum = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int gv = wg * 4 + v;
                if (gv >= G) continue;
                const int edge = (ub0 + 1) * T;                       // first token of slot 1's clip
                const int s0v = 32 * gv >= edge ? 1 : 0;              // the wave's primary slot
                const bool sv = !s0v && min(32 * gv + 31, M - 1) >= edge;
                if (s0v == sl) {
                    const f32x8 p = pst_of(v)[oc * 64 + lane];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += p[i];
                    ssum += ss[(v * 4 + oc) * 32 + c];
                }
                if (sv && sl == 1) {
                    const f32x8 p = xp[oc * 64 + lane];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += p[i];
                    ssum += ss[(4 * 4 + oc) * 32 + c];
                }
            }
            float* R = recs_out + ((size_t)wg * 2 + sl) * DC_REC_FLOATS;
            if (lane < 32) {
                R[32 * oc + c] = wg_colmax4(mx, oc, sl, c);
                R[128 + 32 * oc + c] = ssum;
            }
            reinterpret_cast<f32x8*>(R + 256)[oc * 64 + lane] = acc;
        }
    }
}

// One FiLM block: E = W_blk S + c for the wave's 32 tokens, all 8 feature tiles (G'_0, H'_0, G'_1, H'_1, ...) at once.
// C0 = the block's first chunk, J0 = its first FiLM iteration of the layer (S register ring slot = iteration % 3),
// NPREV0 = operations the wave issued in the iteration before the block.  FIRST: the block that opens the kernel (its
// chunk 0 has landed, chunks 1 and 2 are in flight, S(0) and S(1) have landed).
template <class T16, bool SAFE, int C0, int J0, int NPREV0, bool FIRST>
DEV void film_block(f16x16 (&E)[8], v8<T16> (&sreg)[3][2], const v8<T16>* Sg, unsigned soff, const float* fb, char* lds,
                    const void* stream, int wave, int lane, int hh, bool compute) {
#ifdef EXP_NO_FILM
    compute = false;
#endif
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = ld_ft(fb, t, hh);
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
        constexpr int dummy = 0;
        (void)dummy;
        if (kt == 0) {
            if constexpr (!FIRST) ring_wait<NPREV0>();
        } else if (kt <= 14) {
            ring_wait<6>();                                     // previous iteration: 2 S loads + 4 DMAs
        } else {
            ring_wait<4>();                                     // iteration 14 issued no S loads
        }
        if (kt + 2 < 16) {
#pragma unroll
            for (int s = 0; s < 2; ++s) sreg[(J0 + kt + 2) % 3][s] = s_load<SAFE, T16>(Sg + (size_t)(2 * (kt + 2) + s) * 64, soff);
        }
        if (!(FIRST && kt == 0)) ring_issue(stream, C0 + kt + 2, lds, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (compute) {
            const v8<T16>* w = reinterpret_cast<const v8<T16>*>(ring_slot(lds, C0 + kt)) + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const v8<T16> b = sreg[(J0 + kt) % 3][s];
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) {
#pragma unroll
                    for (int t = 4 * tq; t < 4 * tq + 4; ++t) acc[t] = mfma(w[(t * 2 + s) * 64], b, acc[t]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) E[t] = pack_tile(acc[t]);
}

// F.softmax over head_dim for the two heads of one 32-feature tile (see softmax_heads_ft)
DEV void softmax_heads_tile(f32x16& q) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float m = q[8 * p];
#pragma unroll
        for (int j = 1; j < 8; ++j) m = fmaxf(m, q[8 * p + j]);
        m = xhalf_max(m);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e = exp2f_fast(q[8 * p + j] - m);     // q carries log2(e): folded into Wq, bq
            q[8 * p + j] = e;
            s += e;
        }
        const float inv = fast_rcp(xhalf_sum(s));
#pragma unroll
        for (int j = 0; j < 8; ++j) q[8 * p + j] *= inv;
    }
}

// q = softmax_heads(Wq LN(h) + bq);  y = q . A per head.  The query image is packed by OUTPUT-tile pairs: chunk cQ holds
// output tiles 0,1 (all four k-tiles: fragment ((kt * 2 + ot2) * 2 + s)), chunk cQ+1 tiles 2,3 - a 32-feature tile holds
// whole heads, so each chunk's two tiles run projection -> softmax -> attention on their own and only 2 accumulator tiles
// are live next to the FiLM tiles (all four at once cost 270 spilled registers).
// `mid` is issued at the top of the second iteration, ahead of its chunk DMA (NS0 = operations s_issue0 adds to the first).
template <class T16, int NPREV, int NS0, class FS0, class FS1>
DEV void query_attend_ring(f16x16 (&y)[4], float& y_rstd, float& y_shift, const f32x16 (&h)[4], const float* bq, char* lds,
                           const void* stream, int cQ, const v8<T16>* a0, const v8<T16>* a1, const GroupCtx& cx, int wave,
                           FS0&& s_issue0, FS1&& s_issue1, bool compute) {
#ifdef EXP_NO_Q
    compute = false;
#endif
    XFrag<T16, false> nf[4];
    RowStats st;
    auto half = [&](int op) {
        f32x16 q[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) q[t] = ld_ft(bq, 2 * op + t, cx.hh);
        gemm_wa<2, 4, T16, false>(q, reinterpret_cast<const v8<T16>*>(ring_slot(lds, cQ + op)), nf, cx.lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int oc = 2 * op + t;
            softmax_heads_tile(q[t]);
            f32x16 acc = splat(0.f);
            XFrag<T16, false> qf;
            make_frag<T16, false>(q[t], qf);
            if (!cx.straddle) {
                attn_apply_tile<T16, false>(acc, a0, oc, qf, cx.lane);
            } else {   // the group spans two clips: apply each clip's matrix to its own tokens (lanes)
                XFrag<T16, false> qm = qf;
                mask_frag<T16, false>(qm, cx.lane_in_b0);
                attn_apply_tile<T16, false>(acc, a0, oc, qm, cx.lane);
                mask_frag<T16, false>(qf, !cx.lane_in_b0);
                attn_apply_tile<T16, false>(acc, a1, oc, qf, cx.lane);
            }
            st.add(acc);
            put_y<false>(y[oc], acc);
        }
    };
    ring_wait<NPREV>();
    s_issue0();
    ring_issue(stream, cQ + 2, lds, wave, cx.lane);
    if (compute) {
        ln_frags<T16, false>(nf, h);
        half(0);
    }
    ring_wait<4 + NS0>();
    s_issue1();
    ring_issue(stream, cQ + 3, lds, wave, cx.lane);
    if (compute) {
        half(1);
        st.finish(y_rstd, y_shift);
    }
}

// StylizationBlock accumulated into the residual stream with the FiLM tiles in registers:
//   h += W_o SiLU(nhat G' + H') + b_o.   Chunks cO, cO+1 = the out-projection image's k-tile halves.
template <class T16, int NPREV, int NS0, class FS0, class FS1>
DEV void styl_ring(f32x16 (&h)[4], const f16x16 (&y)[4], float rstd, float shift, const f16x16 (&E)[8], const float* bo, char* lds,
                   const void* stream, int cO, int lane, int hh, int wave, FS0&& s_issue0, FS1&& s_issue1, bool compute) {
#ifdef EXP_NO_STYL
    compute = false;
#endif
    ring_wait<NPREV>();
    s_issue0();
    ring_issue(stream, cO + 2, lds, wave, lane);
    if (compute) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x16 bb = ld_ft(bo, t, hh);
#pragma unroll
            for (int r = 0; r < 16; ++r) h[t][r] += bb[r];
        }
        const v8<T16>* w = reinterpret_cast<const v8<T16>*>(ring_slot(lds, cO));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            XFrag<T16, false> zf;
            styl_tile<T16, false, f16x16>(zf, y[kt], rstd, shift, E[2 * kt], E[2 * kt + 1]);
            mma_kt<4, 2, T16, false>(h, w, kt, zf, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    ring_wait<4 + NS0>();
    s_issue1();
    ring_issue(stream, cO + 3, lds, wave, lane);
    if (compute) {
        const v8<T16>* w = reinterpret_cast<const v8<T16>*>(ring_slot(lds, cO + 1));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            XFrag<T16, false> zf;
            styl_tile<T16, false, f16x16>(zf, y[2 + kt], rstd, shift, E[4 + 2 * kt], E[4 + 2 * kt + 1]);
            mma_kt<4, 2, T16, false>(h, w, kt, zf, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// k_flayer: one decoder layer for one 128-token unit (see the header).  Stream chunks (dc_common.h DCF_CH_*):
//   F0 0..15 | Q 16,17 | O 18,19 | F1 20..35 | CQ 36,37 | CO 38,39 | F2 40..55 | FFN 56,57 | FO 58,59 | K 60,61 | V 62,63
//   (last layer: OUT 60 instead of K, V).
// DBG: test-hook build - stop after block (dbg & 0xff) = 1, 2, 3; (dbg >> 16) & 3 leading blocks skipped; tracked loads.
// ------------------------------------------------------------------------------------------------------------------
template <class T16, bool DBG>
__global__ __launch_bounds__(256, 2)
void k_flayer(const DcModel* __restrict__ dm, int l, float* __restrict__ hbuf, const v8<T16>* __restrict__ S,
              const v8<T16>* __restrict__ a_ca /*[L][B][16][64]*/, float* __restrict__ recs, const int* __restrict__ length,
              const float* __restrict__ xin, float* __restrict__ xout, int out_mode, const float* __restrict__ coef_cur,
              const int* __restrict__ snap_cur, float* __restrict__ snaps, int M, int T, int G, int B, int dbg, size_t rec_stride,
              const int* __restrict__ iter_base) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using W = v8<T16>;
    constexpr bool SAFE = DBG;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wg = wg_index();
    int g = wg * FNW + wave;
    const bool active = g < G;                   // idle waves still take part in the staging and barriers
    if (!active) g = G - 1;
    g = __builtin_amdgcn_readfirstlane(g);
    const GroupCtx cx = make_ctx(g, lane, M, T);
    const int nl = dm->num_layers;
    const bool last = l + 1 >= nl;
    const void* stream = dm->fl_stream[l];
    const float* cst = reinterpret_cast<const float*>(lds + F_OFF_CONST);
    const int ub0 = (wg * FUT) / T;
    const W* af = reinterpret_cast<const W*>(lds + F_OFF_AF);
    const W* a0 = af + (size_t)(cx.b0 - ub0) * 8 * 64;
    const W* a1 = af + (size_t)(cx.b1 - ub0) * 8 * 64;
    const int skip_blocks = DBG ? (dbg >> 16) & 3 : 0;
    const int stop_after = DBG ? dbg & 0xff : 0;

    // ---- prologue: residual stream, constants, chunk 0, S(0), S(1) in flight; combine of the previous layer's records
    f32x16 h[4];
    load_h(h, hbuf, g, lane);
    {
        const bf16x8* csrc = reinterpret_cast<const bf16x8*>(dm->fl_consts[l]);
        lds_dma16(csrc + (size_t)(2 * wave) * 64 + lane, lds + F_OFF_CONST + (2 * wave) * 1024);
        lds_dma16(csrc + (size_t)(2 * wave + 1) * 64 + lane, lds + F_OFF_CONST + (2 * wave + 1) * 1024);
    }
    ring_issue(stream, 0, lds, wave, lane);
    const W* Sg = S + (size_t)g * 32 * 64;
    const unsigned soff = lane * 16;
    W sreg[3][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) sreg[j][s] = s_load<SAFE, T16>(Sg + (size_t)(2 * j + s) * 64, soff);
#ifndef EXP_NO_COMBINE
    wg_combine4<T16>(recs + (size_t)(l & 1) * rec_stride, reinterpret_cast<W*>(lds + F_OFF_AF), reinterpret_cast<float*>(lds + F_SLOT),
                     reinterpret_cast<float*>(lds + F_OFF_SCR), ub0, M, T, tid, wg);
#endif
    ring_wait<0>();                              // everything above has landed; the combine's LDS weights are dead
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(h[t]));      // the compiler's own wait for h falls here, where it costs nothing
    ring_issue(stream, 1, lds, wave, lane);
    ring_issue(stream, 2, lds, wave, lane);
    float* recs_out = recs + (size_t)((l + 1) & 1) * rec_stride;
    const W* acl = a_ca + (size_t)l * B * 16 * 64;
    auto no_s = [] {};
    auto stage_ca = [&] {                        // cross-attention fragments of clips ub0, ub0+1 -> AF (4 DMAs per wave)
        const int c1i = min(ub0 + 1, B - 1);
        const W* s0 = acl + (size_t)ub0 * 16 * 64 + lane;
        const W* s1 = acl + (size_t)c1i * 16 * 64 + lane;
        lds_dma16(s0 + (size_t)(2 * wave) * 64, lds + F_OFF_AF + (2 * wave) * 1024);
        lds_dma16(s0 + (size_t)(2 * wave + 1) * 64, lds + F_OFF_AF + (2 * wave + 1) * 1024);
        lds_dma16(s1 + (size_t)(2 * wave) * 64, lds + F_OFF_AF + 8192 + (2 * wave) * 1024);
        lds_dma16(s1 + (size_t)(2 * wave + 1) * 64, lds + F_OFF_AF + 8192 + (2 * wave + 1) * 1024);
    };
    f16x16 E[8], y[4];
    float y_rstd, y_shift;

    // ================= self-attention block =================
    film_block<T16, SAFE, 0, 0, 0, true>(E, sreg, Sg, soff, cst + DCF_C_FILM, lds, stream, wave, lane, cx.hh, skip_blocks < 1);
    // Q: iterations 16, 17 (previous iteration 15 issued 4 DMAs); no S prefetch here (F1 starts at iteration 20)
    query_attend_ring<T16, 4, 0>(y, y_rstd, y_shift, h, cst + DCF_C_BQ_SA, lds, stream, 16, a0, a1, cx, wave, no_s, no_s, skip_blocks < 1);
    // O: iterations 18, 19 issue S(16), S(17) = k-tiles 0, 1 of FiLM block 1 (register slots 16 % 3, 17 % 3); iteration 19
    // also brings the cross-attention fragments (every wave is past the self-attention apply behind iteration 18's barrier)
    styl_ring<T16, 4, 2>(h, y, y_rstd, y_shift, E, cst + DCF_C_BO_SA, lds, stream, 18, lane, cx.hh, wave,
                         [&] {
#pragma unroll
                             for (int s = 0; s < 2; ++s) sreg[16 % 3][s] = s_load<SAFE, T16>(Sg + (size_t)(0 + s) * 64, soff);
                         },
                         [&] {
                             stage_ca();
#pragma unroll
                             for (int s = 0; s < 2; ++s) sreg[17 % 3][s] = s_load<SAFE, T16>(Sg + (size_t)(2 + s) * 64, soff);
                         },
                         skip_blocks < 1);
    if constexpr (DBG)
        if (stop_after == 1) {
            if (active) store_h(h, hbuf, g, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    // ================= cross-attention block =================
    // F1: iteration 19 issued 4 (fragments) + 2 (S) + 4 (chunk) operations
    film_block<T16, SAFE, 20, 16, 10, false>(E, sreg, Sg, soff, cst + DCF_C_FILM + 256, lds, stream, wave, lane, cx.hh, skip_blocks < 2);
    query_attend_ring<T16, 4, 0>(y, y_rstd, y_shift, h, cst + DCF_C_BQ_CA, lds, stream, 36, a0, a1, cx, wave, no_s, no_s, skip_blocks < 2);
    styl_ring<T16, 4, 2>(h, y, y_rstd, y_shift, E, cst + DCF_C_BO_CA, lds, stream, 38, lane, cx.hh, wave,
                         [&] {
#pragma unroll
                             for (int s = 0; s < 2; ++s) sreg[32 % 3][s] = s_load<SAFE, T16>(Sg + (size_t)(0 + s) * 64, soff);
                         },
                         [&] {
#pragma unroll
                             for (int s = 0; s < 2; ++s) sreg[33 % 3][s] = s_load<SAFE, T16>(Sg + (size_t)(2 + s) * 64, soff);
                         },
                         skip_blocks < 2);
    if constexpr (DBG)
        if (stop_after == 2) {
            if (active) store_h(h, hbuf, g, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    // ================= FFN block =================
    film_block<T16, SAFE, 40, 32, 6, false>(E, sreg, Sg, soff, cst + DCF_C_FILM + 512, lds, stream, wave, lane, cx.hh, true);
    {
        // FFN: iteration 56 = W1 (16 fragments), 57 = W2
        ring_wait<4>();
        ring_issue(stream, 58, lds, wave, lane);
        f32x16 u[2];
        {
            XFrag<T16, false> hf[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) make_frag<T16, false>(h[kt], hf[kt]);
#pragma unroll
            for (int t = 0; t < 2; ++t) u[t] = ld_ft(cst + DCF_C_B1, t, cx.hh);
            gemm_wa<2, 4, T16, false>(u, reinterpret_cast<const W*>(ring_slot(lds, 56)), hf, lane);
        }
        XFrag<T16, false> uf[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) u[kt][r] = gelu_erf(u[kt][r]);
            make_frag<T16, false>(u[kt], uf[kt]);
        }
        ring_wait<4>();
        ring_issue(stream, 59, lds, wave, lane);
        RowStats st;
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                              // one output tile at a time
            f32x16 yf = ld_ft(cst + DCF_C_B2, t, cx.hh);
            mma_ot<4, 2, T16, false>(yf, reinterpret_cast<const W*>(ring_slot(lds, 57)), t, uf, lane);
            st.add(yf);
            put_y<false>(y[t], yf);
        }
        st.finish(y_rstd, y_shift);
    }
    // FO: iterations 58, 59 issue chunks 60 and (not in the last layer) 61
    {
        ring_wait<4>();
        ring_issue(stream, 60, lds, wave, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x16 bb = ld_ft(cst + DCF_C_BO_FFN, t, cx.hh);
#pragma unroll
            for (int r = 0; r < 16; ++r) h[t][r] += bb[r];
        }
        const W* w = reinterpret_cast<const W*>(ring_slot(lds, 58));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            XFrag<T16, false> zf;
            styl_tile<T16, false, f16x16>(zf, y[kt], y_rstd, y_shift, E[2 * kt], E[2 * kt + 1]);
            mma_kt<4, 2, T16, false>(h, w, kt, zf, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        ring_wait<4>();
        if (!last) ring_issue(stream, 61, lds, wave, lane);
        const W* w2 = reinterpret_cast<const W*>(ring_slot(lds, 59));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            XFrag<T16, false> zf;
            styl_tile<T16, false, f16x16>(zf, y[2 + kt], y_rstd, y_shift, E[4 + 2 * kt], E[4 + 2 * kt + 1]);
            mma_kt<4, 2, T16, false>(h, w2, kt, zf, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (DBG)
        if (stop_after == 3) {
            if (active) store_h(h, hbuf, g, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
#ifdef EXP_NO_TAIL
    if (!last) { if (active) store_h(h, hbuf, g, lane); return; }
#endif
    if (!last) {
        front_tail<T16, 4>(h, active, hbuf, lds, stream, 60, cst + DCF_C_BK, cst + DCF_C_BV, cx, active, wave, lane, wg, ub0, B, M, T, G,
                           length, recs_out);
        return;
    }
#ifdef EXP_NO_OUT
    return;
#endif
    // ---- output projection (always split: 8 hi + 8 lo fragments = chunk 60) + DDIM update
    ring_wait<0>();
    f32x16 x0[1];
    {
        XFrag<T16, true> hf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) make_frag<T16, true>(h[kt], hf[kt]);
        x0[0] = ld_ft(cst + DCF_C_BK, 0, cx.hh);
        gemm_wa<1, 4, T16, true>(x0, reinterpret_cast<const W*>(ring_slot(lds, 60)), hf, lane);
    }
    if (!active || cx.tok >= M) return;
    const int P = dm->input_feats;
    if (out_mode == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            if (f < P) xout[(size_t)cx.tok * P + f] = x0[0][r];
        }
    } else {
        const int ib = iter_base ? *iter_base : 0;
        coef_cur += 4 * ib;
        const float sr = coef_cur[0], srm1 = coef_cur[1], cx0 = coef_cur[2], ceps = coef_cur[3];
        const int snap = snap_cur[ib];
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

// ------------------------------------------------------------------------------------------------------------------
// k_fembed: step prologue for the fused path: h = joint_embed(x) + sequence_embedding[:T] (transformer.py:488-490), then
// layer 0's front half with unit records.  Stream (dm->fe_stream): chunk 0 = joint_embed image (8 hi + 8 lo fragments),
// 1, 2 = layer 0's key image, 3, 4 = its value image; constants (dm->fe_consts): joint_embed bias | bk | bv.
// ------------------------------------------------------------------------------------------------------------------
template <class T16>
__global__ __launch_bounds__(256, 2)
void k_fembed(const DcModel* __restrict__ dm, const float* __restrict__ x /*[M][P]*/, float* __restrict__ hbuf, float* __restrict__ recs,
              const int* __restrict__ length, int M, int T, int G, int B) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using W = v8<T16>;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wg = wg_index();
    int g = wg * FNW + wave;
    const bool active = g < G;
    if (!active) g = G - 1;
    g = __builtin_amdgcn_readfirstlane(g);
    const GroupCtx cx = make_ctx(g, lane, M, T);
    const int P = dm->input_feats;
    const bool live = cx.tok < M;
    const int n = live ? cx.tok % T : 0;
    const void* stream = dm->fe_stream;
    const float* cst = reinterpret_cast<const float*>(lds + F_OFF_CONST);
    const int ub0 = (wg * FUT) / T;
    {
        const bf16x8* csrc = reinterpret_cast<const bf16x8*>(dm->fe_consts);
        if (wave < 2) lds_dma16(csrc + (size_t)wave * 64 + lane, lds + F_OFF_CONST + wave * 1024);
    }
    ring_issue(stream, 0, lds, wave, lane);
    ring_issue(stream, 1, lds, wave, lane);
    // sequence_embedding rows: issued first, consumed after the embedding GEMM
    const float* se = dm->seq_emb + (size_t)n * DC_D;
    f32x4 sev[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) sev[4 * t + q] = *reinterpret_cast<const f32x4*>(se + 32 * t + 8 * q + 4 * cx.hh);
    f32x16 h[4];
    {
        XFrag<T16, true> xf[1];
        f32x16 xv;
        // A group's 32 x P floats are contiguous: 16-byte loads (<= 4 per lane) turned through a wave-private LDS patch
        const bool staged = active && 32 * g + 32 <= M;                   // wave-uniform
        if (staged) {
            float* xs = reinterpret_cast<float*>(lds + F_OFF_AF + wave * 3584);
            const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)g * 32 * P);
            f32x4 ch[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i * 64 + lane < 8 * P) ch[i] = src[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i * 64 + lane < 8 * P) reinterpret_cast<f32x4*>(xs)[i * 64 + lane] = ch[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = tile_row(r, cx.hh);
                xv[r] = f < P ? xs[cx.c * P + f] : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = tile_row(r, cx.hh);
                xv[r] = (live && f < P) ? x[(size_t)cx.tok * P + f] : 0.f;
            }
        }
        make_frag<T16, true>(xv, xf[0]);
        ring_wait<0>();                                   // chunks 0, 1 and the constants have landed
        ring_issue(stream, 2, lds, wave, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = ld_ft(cst, t, cx.hh);
        gemm_wa<4, 1, T16, true>(h, reinterpret_cast<const W*>(ring_slot(lds, 0)), xf, lane);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) h[t][4 * q + i] += sev[4 * t + q][i];
    front_tail<T16, 4>(h, active, hbuf, lds, stream, 1, cst + 128, cst + 256, cx, active, wave, lane, wg, ub0, B, M, T, G, length, recs);
}

// ---- launchers --------------------------------------------------------------------------------------------------------
static hipError_t f_optin(const void* fn) {
    static unsigned long long done[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    static const void* fns[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int dev = 0, slot = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    for (int i = 0; i < 8; ++i)
        if (fns[i] == fn || fns[i] == nullptr) {
            slot = i;
            fns[i] = fn;
            break;
        }
    if (slot >= 0 && dev < 64 && ((done[slot] >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
    if (e == hipSuccess && slot >= 0 && dev < 64) done[slot] |= 1ull << dev;
    return e;
}

hipError_t dc_launch_fembed(hipStream_t st, int fmt, const DcModel* dm, const float* x, float* hbuf, float* recs, const int* length,
                            int M, int T, int G, int B) {
    const dim3 grid((G + FNW - 1) / FNW), block(FNW * 64);
    if (fmt == 1) {
        if (hipError_t e = f_optin((const void*)k_fembed<_Float16>)) return e;
        k_fembed<_Float16><<<grid, block, F_LDS, st>>>(dm, x, hbuf, recs, length, M, T, G, B);
    } else {
        if (hipError_t e = f_optin((const void*)k_fembed<__bf16>)) return e;
        k_fembed<__bf16><<<grid, block, F_LDS, st>>>(dm, x, hbuf, recs, length, M, T, G, B);
    }
    return hipGetLastError();
}

template <class T16, bool DBG>
static hipError_t launch_flayer_t(hipStream_t st, const DcModel* dm, int l, float* hbuf, const void* S, const void* a_ca, float* recs,
                                  const int* length, const float* xin, float* xout, int out_mode, const float* coef_cur,
                                  const int* snap_cur, float* snaps, int M, int T, int G, int B, int dbg, size_t rec_stride,
                                  const int* iter_base) {
    if (hipError_t e = f_optin((const void*)k_flayer<T16, DBG>)) return e;
    k_flayer<T16, DBG><<<dim3((G + FNW - 1) / FNW), dim3(FNW * 64), F_LDS, st>>>(dm, l, hbuf, (const v8<T16>*)S, (const v8<T16>*)a_ca, recs,
                                                                                length, xin, xout, out_mode, coef_cur, snap_cur, snaps,
                                                                                M, T, G, B, dbg, rec_stride, iter_base);
    return hipGetLastError();
}
hipError_t dc_launch_flayer(hipStream_t st, int fmt, const DcModel* dm, int l, float* hbuf, const void* S, const void* a_ca, float* recs,
                            const int* length, const float* xin, float* xout, int out_mode, const float* coef_cur,
                            const int* snap_cur, float* snaps, int M, int T, int G, int B, int dbg, size_t rec_stride,
                            const int* iter_base) {
#define FL_ARGS st, dm, l, hbuf, S, a_ca, recs, length, xin, xout, out_mode, coef_cur, snap_cur, snaps, M, T, G, B, dbg, rec_stride, iter_base
#ifdef DC_FUSED_ONLY_F16
    return launch_flayer_t<_Float16, false>(FL_ARGS);
#else
    if (dbg != 0) return fmt == 1 ? launch_flayer_t<_Float16, true>(FL_ARGS) : launch_flayer_t<__bf16, true>(FL_ARGS);
    return fmt == 1 ? launch_flayer_t<_Float16, false>(FL_ARGS) : launch_flayer_t<__bf16, false>(FL_ARGS);
#endif
#undef FL_ARGS
}
int dc_fused_max_units_per_clip(void) { return F_NU; }
int dc_fused_unit_tokens(void) { return FUT; }
