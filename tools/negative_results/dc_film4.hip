// dc_film4.hip - round 3's measured dead end for the FiLM GEMM: the 256-accumulator-register wave tile at ONE wave per SIMD with
// hand-placed loads.  NOT part of the library.  It was built into dc_kernels.hip behind DC_FILM4=1
// (kernel below, launcher branch at the end), verified bit-identical to k_film_gemm3 (tools/check_film4.py at the time: B=3 x 300
// and B=32 x 1800, DDIM-50) and timed against it on the same box (gpurun_out -> profiles/r03_ab_film4.txt):
//     weight ring 2 k-steps deep:  18.5 ms per 50 launches vs 15.0  (+23 %)
//     weight ring 4 k-steps deep:  19.6 - 19.9 ms vs 15.2 - 15.3    (+29 %)
// The main loop compiles as intended (8 MFMA | 1 global load | 1 LDS read, no spills, no waits on fresh loads), so what is
// missing is the second wave: the slab fill, the 256-register epilogue (convert, lane swaps, 128 stores) and every residual
// wait run with the matrix pipe idle.  To build it again: paste the kernel before k_embed_front in dc_kernels.hip and the
// launcher branch in front of `int nwg = ...` in launch_film3_t.

// ------------------------------------------------------------------------------------
// Round-3 experiment (DC_FILM4=1; measured in DESIGN.md section 4, not the default): the same S-stationary GEMM with a
// 256-ACCUMULATOR-REGISTER wave tile at ONE wave per SIMD - 4 waves per workgroup, each sweeping TWO tile pairs (128 features x
// 128 tokens): per 32-deep k-step 8 weight fragments (L2 -> registers) and 8 slab fragments (LDS) feed 64 MFMAs, i.e. every slab
// fragment read feeds 8 MFMAs instead of 4 (half the LDS reads per FLOP of k_film_gemm3).  With no second wave on the SIMD to
// cover latencies the loads are placed by hand between the MFMAs (sched_group_barrier: one LDS read + one global load per 8
// MFMAs, two k-steps ahead).  Static ownership: wave w owns pairs 2w, 2w+1 of every round.  Results are bit-identical to
// k_film_gemm3 (same products in the same order per accumulator).
// ------------------------------------------------------------------------------------
template <class T16>
__global__ __launch_bounds__(256, 1) void k_film_gemm4(const v8<T16>* __restrict__ W, const float* __restrict__ bias16,
                                                       f16x16* __restrict__ E, int G, int NT, int nround,
                                                       const float* __restrict__ pp, const float* __restrict__ temb,
                                                       const int* __restrict__ t_clip, int T, int B, const int* __restrict__ iter_base) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using OP = v8<T16>;
    constexpr int KS = DC_E / 32;          // 16 k-steps
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const OP* slab = reinterpret_cast<const OP*>(lds);
    const int nblk = (G + 3) / 4;
    const long long nunit = (long long)nblk * nround;
    const int u0 = (int)(nunit * blockIdx.x / gridDim.x), u1 = (int)(nunit * (blockIdx.x + 1) / gridDim.x);
    auto wpair = [&](int p) { return W + (size_t)(2 * p) * 2 * KS * 64 + lane; };
    auto wfrag = [&](const OP* w, int ks, int i) { return w[((size_t)(i >> 1) * 2 * KS + ks * 2 + (i & 1)) * 64]; };
    int tb_cur = -1;
    for (int u = u0; u < u1; ++u) {
        const int tb = u / nround, r = u % nround;
        const int g0 = tb * 4;
        if (tb != tb_cur) {        // slab fill (as k_film_gemm3, 4 waves: 32 fragments each, in four batches of 8)
            tb_cur = tb;
            v8<T16>* slab_w = reinterpret_cast<v8<T16>*>(lds);
            const float* trow[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gg = min(g0 + (i >> 1), G - 1);
                const int b = min((gg * 32 + 16 * (i & 1) + (lane & 15)) / T, B - 1);
                trow[i] = temb + (size_t)t_clip[iter_base ? *iter_base : b] * 512 + 8 * (lane >> 4);
            }
            __syncthreads();                                  // everyone is done with the previous slab
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x8 pv[8], tv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = wave + 4 * (8 * q + i);                    // fragment (g, tb16, ks32) = (f >> 5, (f >> 4) & 1, f & 15)
                    const int gi = f >> 5, t16 = (f >> 4) & 1, ks = f & 15;
                    const int gg = min(g0 + gi, G - 1);
                    pv[i] = ld_pp(pp, (size_t)gg * DC_KS_E + 2 * ks + (lane >> 5), 32 * ((lane >> 4) & 1) + 16 * t16 + (lane & 15));
                    tv[i] = *reinterpret_cast<const f32x8*>(trow[2 * gi + t16] + 32 * ks);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = wave + 4 * (8 * q + i);
                    v8<T16> hi;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 z = silu_pair(pv[i][2 * j] + tv[i][2 * j], pv[i][2 * j + 1] + tv[i][2 * j + 1]);
                        hi[2 * j] = (T16)z.x;
                        hi[2 * j + 1] = (T16)z.y;
                    }
                    slab_w[f * 64 + lane] = hi;
                }
            }
            __syncthreads();
        }
        const int p0 = r * 8 + 2 * wave;                      // this wave's pairs p0, p0 + 1
        const OP* w0 = wpair(p0);
        const OP* w1 = wpair(p0 + 1);
        f32x4 acc[2][2][2][4][2];                             // [pair][tile][fb][g][t16]
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(bias16 + (((size_t)(2 * (p0 + pi) + ti) * 2 + fb) * 4 + (lane >> 4)) * 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[pi][ti][fb][g][0] = acc[pi][ti][fb][g][1] = c;
                }
        auto sfrag = [&](int i, int ks) { return slab[((size_t)(i >> 1) * 2 + (i & 1)) * KS * 64 + (size_t)ks * 64 + lane]; };   // i = 2 g + t16
#ifndef DC_FILM4_PF
#define DC_FILM4_PF 4
#endif
        constexpr int PF = DC_FILM4_PF;                       // weight ring depth in k-steps (global loads: L2 latency to cover)
        OP a[PF][8], bb[2][8];                                // weight fragments (pair, tile, fb) PF k-steps deep; slab fragments 2 deep
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[q][i] = wfrag(i < 4 ? w0 : w1, q, i & 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) bb[0][i] = sfrag(i, 0);
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS; ks0 += PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int ks = ks0 + q;
                const int ksn = (ks + PF) & (KS - 1);         // refill this slot for k-step ks + PF (wraps harmlessly at the end)
#pragma unroll
                for (int i = 0; i < 8; ++i) bb[(q + 1) & 1][i] = sfrag(i, (ks + 1) & (KS - 1));
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        acc[k >> 2][(k >> 1) & 1][k & 1][i >> 1][i & 1] =
                            mfma16(a[q][k], bb[q & 1][i], acc[k >> 2][(k >> 1) & 1][k & 1][i >> 1][i & 1]);
                    a[q][k] = wfrag(k < 4 ? w0 : w1, ksn, k & 3);        // weight fragment k of this slot is consumed: refill it
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 slab read (for the next k-step)
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // 8 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 global load (PF k-steps ahead)
                }
            }
        }
        // epilogue: as k_film_gemm3, for both pairs
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int p = p0 + pi;
            const int blk = p >> 2, t = p & 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g0 + g >= G) continue;
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    u32x8 o;
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const f32x4 x = acc[pi][ti][fb][g][0], y = acc[pi][ti][fb][g][1];
                            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                            const h2v xp = {(_Float16)x[2 * h2], (_Float16)x[2 * h2 + 1]}, yp = {(_Float16)y[2 * h2], (_Float16)y[2 * h2 + 1]};
                            const auto rr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, xp), __builtin_bit_cast(uint32_t, yp), false, false);
                            o[4 * fb + h2] = rr[0];
                            o[4 * fb + 2 + h2] = rr[1];
                        }
                    store_etile(E, (size_t)(g0 + g) * NT + blk * 8 + 4 * ti + t, lane, __builtin_bit_cast(f16x16, o));
                }
            }
        }
    }
}


// ---- launcher branch (inside launch_film3_t) ----
#if 0
    if (getenv("DC_FILM4") && !(ea && ea->x) && round0 == 0) {     // round-3 experiment: 256-accumulator tile, one wave per SIMD
        static unsigned long long optin4 = 0;
        if (hipError_t e = lds_optin((const void*)k_film_gemm4<T16>, 4 * DC_KS_E * 1024, optin4)) return e;
        const int n4 = (int)(nunit < ncu ? nunit : ncu);
        k_film_gemm4<T16><<<dim3(n4), dim3(256), 4 * DC_KS_E * 1024, st>>>((const v8<T16>*)W16, bias16, (f16x16*)E, G, NT, nround, pp, temb,
                                                                             t_clip, T, B, iter_base);
        return hipGetLastError();
    }
#endif
