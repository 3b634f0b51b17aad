// probe_fused.hip - feasibility probe for fusing the FiLM GEMM into the decoder-layer kernel ("fused layer").
//
// Skeleton of the fused layer kernel's data movement, without the layer's arithmetic details:
//   * a workgroup = 4 waves x 32 tokens; two workgroups per CU (<= 80 KiB LDS, <= 256 VGPRs), so the two waves of a SIMD
//     belong to different workgroups and drift freely against each other (one in a FiLM phase, the other in a VALU phase);
//   * ALL weights of a layer (FiLM blocks and the 128-wide projections) arrive as one linear stream of 16-KiB chunks
//     (16 MFMA fragments) through an LDS ring filled by LDS-DMA, one barrier per chunk;
//   * the FiLM operand S = SiLU(emb) of the wave's own 32 tokens comes straight from global memory (f16 fragment image,
//     2 fragments per 32-deep k-tile), prefetched 3 k-tiles ahead into registers;
//   * per FiLM block a wave accumulates all 8 output tiles (128 accumulator registers) over the 16 k-tiles;
//   * between FiLM blocks: `chain_slots` chunks of 128-wide-GEMM-like work (4 accumulator chains) + `valu_iters` rounds of
//     transcendental VALU work on 64 registers stand in for the attention / stylization chain.
// Prints the time per launch and the MFMA rate it implies.   Build: hipcc --offload-arch=gfx950 -O3 -o probe_fused probe_fused.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DEV __device__ __forceinline__
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

DEV void lds_dma16(const void* gsrc, const char* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
DEV f16x8 ld16_nowait(const f16x8* p) {
    f16x8 v;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// uniform base in SGPRs + per-lane byte offset in one VGPR: distinct fragments cost scalar adds, not VGPR address pairs
#ifndef S_MODE
#define S_MODE 1       // 1 = compiler-tracked loads, 2 = saddr asm without waits, 3 = 64-bit vaddr asm without waits
#endif
DEV f16x8 ld16_nowait_s(const void* sbase, unsigned voff) {
    f16x8 v;
#if S_MODE == 1
    v = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(sbase) + voff);
#elif S_MODE == 2
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
#else
    const char* p = reinterpret_cast<const char*>(sbase) + voff;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
#endif
    return v;
}
DEV f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

#ifndef NSLOT_
#define NSLOT_ 4
#endif
#ifndef NWAVES
#define NWAVES 4
#endif
#ifndef S_MODE
#define S_MODE 1
#endif
constexpr int NSLOT = NSLOT_;        // ring slots of 16 KiB
constexpr int FPW = 16 / NWAVES;     // fragments each wave copies per chunk
constexpr int SLOT_BYTES = 16384;

// wait until only the DMAs of the newest NSLOT-2 iterations (+ their S operations in the untracked modes) are outstanding
#if S_MODE == 1
#define VM_N ((NSLOT - 2) * FPW)
#else
#define VM_N ((NSLOT - 2) * (FPW + 2) + FPW)
#endif
#ifdef NO_DMA
#define RING_WAITVM()
#else
#define RING_WAITVM() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_N) : "memory")
#endif
#ifdef NO_BAR
#define RING_SYNC() do { RING_WAITVM(); } while (0)
#else
#define RING_SYNC() do { RING_WAITVM(); __syncthreads(); } while (0)
#endif
// one launch = `nblk` FiLM blocks, each followed by `chain_slots` chunks of chain work
template <int WITH_S>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 4 ? 2 : 1) void k_probe(const f16x8* __restrict__ W, const f16x8* __restrict__ S, float* __restrict__ out,
                                                  int G, int nblk, int chain_slots, int valu_iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g = min((int)blockIdx.x * NWAVES + wave, G - 1);
    const int slots_per_blk = 16 + chain_slots;
    const int total = nblk * slots_per_blk;
    const f16x8* Sg = S + (size_t)g * 32 * 64;                   // this wave's 32 fragments (wave-uniform base)
    const unsigned soff = lane * 16;
    auto issue = [&](int c) {                                   // chunk c of the stream -> slot c % NSLOT; 4 fragments per wave
#ifdef NO_DMA
        return;
#endif
        if (c < total) {
            const f16x8* src = W + (size_t)c * 16 * 64 + lane;
            char* dst = lds + (c % NSLOT) * SLOT_BYTES;
#pragma unroll
            for (int i = 0; i < FPW; ++i) lds_dma16(src + (size_t)(wave * FPW + i) * 64, dst + (wave * FPW + i) * 1024);
        } else {                                                // keep the vmcnt arithmetic uniform: dummy loads
#pragma unroll
            for (int i = 0; i < FPW; ++i) lds_dma16(W + lane, lds + NSLOT * SLOT_BYTES + wave * 1024);
        }
    };
    f32x16 h[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[t][r] = 0.01f * (float)(lane + r + t);
    // prologue: S fragments of k-tiles 0, 1 and chunks 0 .. NSLOT-2.  Per iteration a wave issues 2 S loads, then 4 DMAs; at the
    // top of iteration c it needs S(c) (issued first in iteration c-2) and chunk c (iteration c-3): vmcnt(10) = "all but the 4
    // DMAs of iteration c-2 and the 6 operations of iteration c-1".
    f16x8 sreg[4][2];        // ring of 4 k-tiles: 16 k-tiles per block keep the slot sequence continuous across blocks
#pragma unroll
    for (int c = 0; c < NSLOT - 1; ++c) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (c < 2)
                sreg[c][s2] = ld16_nowait_s(Sg + (size_t)(2 * c + s2) * 64, soff);
#if S_MODE != 1
            else
                lds_dma16(W + lane, lds + NSLOT * SLOT_BYTES + wave * 1024);
#endif
        }
        issue(c);
    }
#ifdef NO_LDS
    f16x8 wreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wreg[i] = W[lane + 64 * i];
#endif
    int c = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#ifdef TWO_B
        f32x16 acc2[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#endif
#pragma unroll
        for (int kt = 0; kt < 16; ++kt, ++c) {
            RING_SYNC();                                          // chunk c landed for every wave; everyone left slot (c-1) % NSLOT
            // S fragments 2 k-tiles ahead (wraps into the next block's first k-tiles: same tokens, same image).  A load without
            // a wait must have a destination the compiler keeps alive until it has landed: none behind the last block.
            if (kt < 14 || blk + 1 < nblk) {
                const int kn = (kt + 2) & 15;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    sreg[(kt + 2) & 3][s2] = ld16_nowait_s(Sg + (size_t)(WITH_S ? 2 * kn + s2 : 0) * 64, soff);
            } else {
#if S_MODE != 1
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) lds_dma16(W + lane, lds + NSLOT * SLOT_BYTES + wave * 1024);
#endif
            }
            issue(c + NSLOT - 1);
            const f16x8* w = reinterpret_cast<const f16x8*>(lds + (c % NSLOT) * SLOT_BYTES) + lane;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f16x8 b = sreg[kt & 3][s2];
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) {
#pragma unroll
#if defined(NO_LDS) && defined(LDS_DUMMY)
                    for (int t = 4 * tq; t < 4 * tq + 4; ++t) {      // the LDS reads are issued and waited for, but feed nothing
                        const f16x8 tmp = w[(t * 2 + s2) * 64];
                        asm volatile("" ::"v"(tmp));
                        acc[t] = mfma(wreg[t & 3], b, acc[t]);
                    }
#elif defined(NO_LDS)
                    for (int t = 4 * tq; t < 4 * tq + 4; ++t) acc[t] = mfma(wreg[t & 3], b, acc[t]);      // operands never leave the registers
#elif defined(TWO_B)
                    for (int t = 4 * tq; t < 4 * tq + 4; ++t) {      // each LDS fragment feeds two MFMAs (a second token group)
                        const f16x8 a = w[(t * 2 + s2) * 64];
                        acc[t] = mfma(a, b, acc[t]);
                        acc2[t] = mfma(a, sreg[(kt + 1) & 3][s2], acc2[t]);
                    }
#else
                    for (int t = 4 * tq; t < 4 * tq + 4; ++t) acc[t] = mfma(w[(t * 2 + s2) * 64], b, acc[t]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // E tiles as packed f16 (what the stylization consumes): folded into h so that nothing is dead
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h[t][r] += (float)(_Float16)acc[2 * t][r] * (float)(_Float16)acc[2 * t + 1][r];
#ifdef TWO_B
                h[t][r] += (float)(_Float16)acc2[2 * t][r] * (float)(_Float16)acc2[2 * t + 1][r];
#endif
            }
        // chain stand-in: chunks of 128-wide GEMM work (4 chains x 4 k-tiles x ... = 16 MFMAs per chunk) + VALU rounds
        for (int cs = 0; cs < chain_slots; ++cs, ++c) {
            RING_SYNC();
            // two placeholder operations keep the vmcnt arithmetic uniform (LDS-DMA into a scratch KiB: a register-destination
            // load without a wait must never target a register the compiler considers dead - it lands later)
#if S_MODE != 1
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) lds_dma16(W + lane, lds + NSLOT * SLOT_BYTES + wave * 1024);
#endif
            issue(c + NSLOT - 1);
            const f16x8* w = reinterpret_cast<const f16x8*>(lds + (c % NSLOT) * SLOT_BYTES) + lane;
            // 16 MFMAs in two chains of 8, operands converted from h one tile at a time; then VALU rounds on the two results
            f32x16 y[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) y[t] = h[t];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f16x8 xf[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[s2][j] = (_Float16)h[kt][8 * s2 + j];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int t = 0; t < 2; ++t) y[t] = mfma(w[((kt * 2 + t) * 2 + s2) * 64], xf[s2], y[t]);
            }
            for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __builtin_amdgcn_exp2f(-y[t][r]);
                        y[t][r] = y[t][r] * __builtin_amdgcn_rcpf(1.f + e) + 0.25f;
                    }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = 0.5f * h[t][r] + 1e-3f * y[t & 1][r];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += h[t][r];
    if (blockIdx.x * NWAVES + wave < G) out[(size_t)(blockIdx.x * NWAVES + wave) * 64 + lane] = s;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("S_MODE %d NWAVES %d NSLOT %d%s%s\n", S_MODE, NWAVES, NSLOT,
#ifdef NO_DMA
           " NO_DMA",
#else
           "",
#endif
#ifdef NO_BAR
           " NO_BAR"
#else
           ""
#endif
    );
    const int G = argc > 1 ? atoi(argv[1]) : 1800;            // 32 clips x 1800 frames / 32
    const int reps = 20;
    const size_t wbytes = (size_t)8 * 64 * SLOT_BYTES;        // 8 layers x 64 chunks
    const size_t sbytes = (size_t)G * 32 * 1024;
    f16x8 *W, *S;
    float* out;
    CHECK(hipMalloc((void**)&W, wbytes + 65536));
    CHECK(hipMalloc((void**)&S, sbytes));
    CHECK(hipMalloc((void**)&out, (size_t)G * 64 * 4));
    {
        std::vector<uint16_t> hw((wbytes + 65536) / 2), hs(sbytes / 2);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x1c00 + (uint16_t)((i * 2654435761u) >> 22 & 0x3ff);     // ~4e-3 .. 8e-3
        for (size_t i = 0; i < hs.size(); ++i) hs[i] = 0x3400 + (uint16_t)((i * 40503u) >> 6 & 0x3ff);
        CHECK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(S, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    }
    const int shm = NSLOT * SLOT_BYTES + NWAVES * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, shm));
    CHECK(hipFuncSetAttribute((const void*)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, shm));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int nwg = (G + NWAVES - 1) / NWAVES;
    struct Cfg { const char* name; int with_s, nblk, chain, valu; };
    const Cfg cfgs[] = {
        {"FiLM only, no S loads, 3 blocks/launch", 0, 3, 0, 0},
        {"FiLM only, S from global, 3 blocks/launch", 1, 3, 0, 0},
        {"FiLM + chain MFMA (5 chunks/block), no VALU", 1, 3, 5, 0},
        {"FiLM + chain + VALU x2", 1, 3, 5, 4},
        {"FiLM + chain + VALU x4", 1, 3, 5, 8},
        {"FiLM + chain + VALU x8", 1, 3, 5, 16},
        {"whole step in one launch: 24 blocks + chain + VALU x4", 1, 24, 5, 8},
    };
    for (const Cfg& cf : cfgs) {
        float best = 1e30f, tot = 0.f;
        printf("%s ...\n", cf.name);
        for (int r = 0; r < reps + 2; ++r) {
            const f16x8* w = W + (size_t)(r % 8) * 64 * SLOT_BYTES / 16 * (cf.nblk > 3 ? 0 : 1);
            CHECK(hipEventRecord(e0, 0));
            if (cf.with_s)
                hipLaunchKernelGGL(k_probe<1>, dim3(nwg), dim3(NWAVES * 64), shm, 0, w, S, out, G, cf.nblk, cf.chain, cf.valu);
            else
                hipLaunchKernelGGL(k_probe<0>, dim3(nwg), dim3(NWAVES * 64), shm, 0, w, S, out, G, cf.nblk, cf.chain, cf.valu);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { tot += ms; best = ms < best ? ms : best; }
        }
        const double mfma = (double)G * (cf.nblk * (256.0 + 16.0 * cf.chain));      // per wave-group: MFMAs per launch
        const double flops = mfma * 32768.0;
        printf("%-58s mean %8.1f us  best %8.1f us   %6.1f TFLOP/s (%4.1f %% of 2.5 PF)  [FiLM-only flops %5.1f %%]\n", cf.name,
               1e3 * tot / reps, 1e3 * best, flops / (tot / reps * 1e-3) / 1e12, 100.0 * flops / (tot / reps * 1e-3) / 2.5e15,
               100.0 * ((double)G * cf.nblk * 256 * 32768.0) / (tot / reps * 1e-3) / 2.5e15);
    }
    printf("reference points: k_film_gemm3 299 us per step (24 blocks); k_layer 51.5 us per layer\n");
    return 0;
}
