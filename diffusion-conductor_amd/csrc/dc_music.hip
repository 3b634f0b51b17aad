// dc_music.hip - MusicEncoder + encode_music on gfx950 (MI355X), hand-written MFMA kernels.
//
// Reference (restated, not translated): Diffusion_Stage/models/transformer.py
//   :289-311 Conv2dResLayer   reflect-pad 3x3 conv -> BatchNorm2d(eval) -> ReLU, plus identity / 1x1-conv+BN residual
//   :313-340 MusicEncoder     conv1 (1->16->16->16) pool(5,5)/(1,2); conv2 (16->32->32) pool(5,5)/(3,2);
//                             conv3 (32->32->32) pool(3,3)/(1,2); [B,32,T,16] -> [B,512,T] -> Conv1d 512->64 + BatchNorm1d
//   :447-459 encode_music     x = music_encoder(mel); x_proj = proj(x)   (eval mode: no token dropout)
//
// Layout.  mel is [B][Tm][128] fp32 = an image with H = time, W = mel bin, one channel.  Every activation lives in HBM
// as TWO bf16 planes (hi, lo: x ~ hi + lo, 16 mantissa bits) in pixel-major order [b][y][x][c]; a pixel's 16 channels of
// one plane are 32 contiguous bytes, i.e. exactly one k-step of a v_mfma_f32_32x32x16_bf16 B operand:
//   lane (n = l & 31, kh = l >> 5) loads the 16 bytes of channels 8kh..8kh+7 of pixel x0 + n (+ tap offset)
// so the implicit GEMM needs no staging or transposition: out[co][pixel] += W_tap[co][ci] * in[ci][pixel + tap] with the
// folded weights as the A operand (LDS-resident, natural-k fragments) and 3 MFMAs per k-step (hi*hi + lo*hi + hi*lo,
// "bf16x3": ~1e-5 relative, the conditioning must not cost the sampler its 1e-3 parity budget).
// A wave owns 32 consecutive pixels of one image row; the accumulator tile has the output channels on registers
// (row (r&3) + 8(r>>2) + 4(l>>5)) and those pixels on lanes, so bias / ReLU / residual / the hi-lo split are per-lane and
// the stores are 8-byte pieces that tile the row's bytes without gaps.
// Neighbouring taps re-read the same lines from L1/L2; HBM sees each plane about once per layer.
#include "dc_music.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "dc_common.h"

namespace {

#define DEV __device__ __forceinline__

DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEV f32x16 mma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 acc) {
    acc = mfma(ah, bh, acc);
    acc = mfma(al, bh, acc);
    return mfma(ah, bl, acc);
}
DEV int reflect(int i, int n) {        // torch 'reflect' padding of width 1 (no edge repeat)
    i = i < 0 ? -i : i;
    // (callers also ask for halo rows of tiles that reach past the image - up to ROWS + 2 beyond it, values nobody uses: with fewer rows
    // than that in the image the single reflection would leave it on the other side, hence the clamp)
    return max(i >= n ? 2 * n - 2 - i : i, 0);
}
DEV void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        hi[j] = h;
        lo[j] = (__bf16)(v[j] - (float)h);
    }
}
DEV f32x16 zero16() {
    f32x16 x;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.f;
    return x;
}

// ---------------------------------------------------------------------------------------------------------
// Plane formats.  SP = true: the split form above (two bf16 planes, three MFMAs per k-step: ~6e-6 at the encoder's output).
// SP = false: ONE fp16 plane and fp16 weights, one MFMA per k-step, half the LDS reads and half the HBM bytes: every activation is
// rounded to 11 bits once per layer - 3.7e-4 relative at the encoder's output, 1.3e-4 of x0 after DDIM-50 (tests/study_encoder_fp16.py
// reproduces both on the CPU), i.e. the order of the fp16 operand rounding the fp16 sampler applies to these features anyway.
// dc_music_encode picks the format per call (dc_music_set_format / DC_ME_PREC); the kernels are the same templates.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <bool SP> struct PL;
template <> struct PL<true> {
    using e = __bf16;
    using v8 = bf16x8;
    using v4 = bf16x4;
    static constexpr int N = 2;
};
template <> struct PL<false> {
    using e = _Float16;
    using v8 = f16x8;
    using v4 = f16x4;
    static constexpr int N = 1;
};
DEV f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// one k-step: split planes hi*hi + lo*hi + hi*lo; single plane one product (al, bl are then dead values and their loads disappear)
DEV f32x16 mma(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 acc) { return mma3(ah, al, bh, bl, acc); }
DEV f32x16 mma(f16x8 ah, f16x8, f16x8 bh, f16x8, f32x16 acc) { return mfma(ah, bh, acc); }
// four values -> one 8-byte piece per plane
DEV void put4(const float (&v)[4], __bf16* hi, __bf16* lo) {
    bf16x4 oh, ol;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        oh[e] = h;
        ol[e] = (__bf16)(v[e] - (float)h);
    }
    *reinterpret_cast<bf16x4*>(hi) = oh;
    *reinterpret_cast<bf16x4*>(lo) = ol;
}
DEV void put4(const float (&v)[4], _Float16* hi, _Float16*) {
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
    *reinterpret_cast<f16x4*>(hi) = o;
}
// value of element e of an 8-byte piece (pair)
DEV float val4(const bf16x4& h, const bf16x4& l, int e) { return (float)h[e] + (float)l[e]; }
DEV float val4(const f16x4& h, const f16x4&, int e) { return (float)h[e]; }

// ---------------------------------------------------------------------------------------------------------
// 3x3 reflect-padded convolution + folded BatchNorm + ReLU (+ residual).  RES: 0 none, 1 identity, 2 1x1 conv + BN.
// Weight fragments (A operand, natural k = tap*CIN + ci): [hi: NKS][lo: NKS] then for RES == 2 [hi: KC][lo: KC].
// CIN == 1 reads the fp32 mel directly: its single k-step holds the 9 taps (k = tap), split in registers.
// ---------------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int RES>
__global__ __launch_bounds__(256) void k_me_conv(const float* __restrict__ mel, const bf16x8* __restrict__ in_hi,
                                                 const bf16x8* __restrict__ in_lo, __bf16* __restrict__ out_hi,
                                                 __bf16* __restrict__ out_lo, const bf16x8* __restrict__ w,
                                                 const float* __restrict__ bias_ft, const float* __restrict__ rbias_ft,
                                                 int Bc, int H, int W, int tiles_per_wave) {
    constexpr int KC = CIN >= 16 ? CIN / 16 : 1;
    constexpr int NKS = CIN >= 16 ? 9 * KC : 1;
    constexpr int NF = 2 * NKS + (RES == 2 ? 2 * KC : 0);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    bf16x8* wl = reinterpret_cast<bf16x8*>(lds);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, kh = lane >> 5;
    for (int f = wave; f < NF; f += 4) wl[f * 64 + lane] = w[f * 64 + lane];
    __syncthreads();
    f32x16 bias, rbias;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bias[r] = bias_ft[kh * 16 + r];
        rbias[r] = RES == 2 ? rbias_ft[kh * 16 + r] : 0.f;
    }
    const int xt_per_row = W / 32;
    const long long total = (long long)Bc * H * xt_per_row;
    const long long t0 = ((long long)blockIdx.x * 4 + wave) * tiles_per_wave;
    for (int i = 0; i < tiles_per_wave; ++i) {
        const long long tile = t0 + i;
        if (tile >= total) break;
        const int xt = (int)(tile % xt_per_row);
        const int y = (int)((tile / xt_per_row) % H);
        const int b = (int)(tile / ((long long)xt_per_row * H));
        const int x = xt * 32 + n;
        f32x16 acc = zero16();
        if constexpr (CIN == 1) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = 8 * kh + j;
                const int yy = reflect(y + t / 3 - 1, H), xx = reflect(x + t % 3 - 1, W);
                v[j] = t < 9 ? mel[((size_t)b * H + yy) * W + xx] : 0.f;
            }
            bf16x8 bh, bl;
            split8(v, bh, bl);
            acc = mma3(wl[lane], wl[64 + lane], bh, bl, acc);
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = reflect(y + tap / 3 - 1, H), xx = reflect(x + tap % 3 - 1, W);
                const size_t base = (((size_t)b * H + yy) * W + xx) * (CIN / 8) + kh;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    const bf16x8 bh = in_hi[base + 2 * kc], bl = in_lo[base + 2 * kc];
                    const int ks = tap * KC + kc;
                    acc = mma3(wl[ks * 64 + lane], wl[(NKS + ks) * 64 + lane], bh, bl, acc);
                }
            }
        }
        const size_t pix = ((size_t)b * H + y) * W + x;
        f32x16 res = zero16();
        if constexpr (RES == 2) {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const bf16x8 bh = in_hi[pix * (CIN / 8) + kh + 2 * kc], bl = in_lo[pix * (CIN / 8) + kh + 2 * kc];
                res = mma3(wl[(2 * NKS + kc) * 64 + lane], wl[(2 * NKS + KC + kc) * 64 + lane], bh, bl, res);
            }
        }
#pragma unroll
        for (int q = 0; q < COUT / 8; ++q) {
            const int c0 = 8 * q + 4 * kh;          // this lane's 4 consecutive output channels of register group q
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * q + e] + bias[4 * q + e], 0.f);
            if constexpr (RES == 1) {
                const __bf16* ph = reinterpret_cast<const __bf16*>(in_hi) + pix * CIN + c0;
                const __bf16* pl = reinterpret_cast<const __bf16*>(in_lo) + pix * CIN + c0;
                const bf16x4 xh = *reinterpret_cast<const bf16x4*>(ph), xl = *reinterpret_cast<const bf16x4*>(pl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)xh[e] + (float)xl[e];
            }
            if constexpr (RES == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += res[4 * q + e] + rbias[4 * q + e];
            }
            bf16x4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 h = (__bf16)v[e];
                oh[e] = h;
                ol[e] = (__bf16)(v[e] - (float)h);
            }
            *reinterpret_cast<bf16x4*>(out_hi + pix * COUT + c0) = oh;
            *reinterpret_cast<bf16x4*>(out_lo + pix * COUT + c0) = ol;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same convolution for CIN >= 16 from an LDS-resident input tile.  k_me_conv reads every tap of every pixel from
// L1 / L2 (18 KiB of loads per 2 KiB of unique input: 2.3 TB/s of HBM at best); here a persistent 4-wave workgroup
// copies a (ROWS + 2) x (32 NXS + 2)-pixel halo tile of both planes into LDS by LDS-DMA (reflect padding by index) once
// and feeds all nine taps from there.  LDS piece order [plane][16-channel chunk][channel half][row][pixel] x 16 B: the
// B operand of one tap is 32 consecutive pieces per channel half - conflict-free ds_read_b128 - and a tap shift is a
// piece offset.  The weight fragments of a k-step are read once per wave for its ROWS NXS / 4 wave tiles; the identity
// residual and the 1x1 residual convolution take the centre pixels from the same tile.  Two workgroups per CU: one
// fills while the other computes.
// ---------------------------------------------------------------------------------------------------------
DEV void me_dma16(const void* gsrc /*per-lane*/, const char* lds_dst /*wave-uniform*/) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
template <int CIN, int COUT, int RES, int ROWS, int NXS, bool SP>
__global__ __launch_bounds__(256, 2) void k_me_conv_t(const typename PL<SP>::v8* __restrict__ in_hi, const typename PL<SP>::v8* __restrict__ in_lo,
                                                      typename PL<SP>::e* __restrict__ out_hi, typename PL<SP>::e* __restrict__ out_lo,
                                                      const typename PL<SP>::v8* __restrict__ w, const float* __restrict__ bias_ft,
                                                      const float* __restrict__ rbias_ft, int H, int W, int ntiles) {
    using V8 = typename PL<SP>::v8;
    using V4 = typename PL<SP>::v4;
    constexpr int NPL = PL<SP>::N;                  // planes (weights: [hi: NKS][lo: NKS][res hi: KC][res lo: KC], single plane: [NKS][res: KC])
    constexpr int KC = CIN / 16, NKS = 9 * KC, NF = NPL * NKS + (RES == 2 ? NPL * KC : 0);
    constexpr int TX = 32 * NXS, PXW = TX + 2, RH = ROWS + 2;
    constexpr int SEL = RH * PXW;                    // pieces per (plane, chunk, half)
    constexpr int NP = NPL * KC * 2 * SEL;
    constexpr int NDMA = (NP + 63) / 64;
    constexpr int TPW = ROWS * NXS / 4;
    static_assert(ROWS * NXS % 4 == 0, "wave tiles divide over 4 waves");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    V8* wl = reinterpret_cast<V8*>(lds);
    V8* tl = wl + NF * 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, kh = lane >> 5;
    for (int f = wave; f < NF; f += 4) wl[f * 64 + lane] = w[f * 64 + lane];
    f32x16 bias, rbias;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bias[r] = bias_ft[kh * 16 + r];
        rbias[r] = RES == 2 ? rbias_ft[kh * 16 + r] : 0.f;
    }
    const int txn = W / TX, tyn = (H + ROWS - 1) / ROWS;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % txn, ty = (t / txn) % tyn, b = t / (txn * tyn);
        const int x0 = tx * TX, y0 = ty * ROWS;
        for (int j = wave; j < NDMA; j += 4) {
            const int p = 64 * j + lane;
            if (p < NP) {
                const int px = p % PXW, row = (p / PXW) % RH, sel = p / SEL;
                const int yy = reflect(y0 - 1 + row, H), xx = reflect(x0 - 1 + px, W);
                const V8* src = ((sel >= 2 * KC) ? in_lo : in_hi) + (((size_t)b * H + yy) * W + xx) * (CIN / 8) + (sel % (2 * KC));
                me_dma16(src, reinterpret_cast<const char*>(tl + 64 * j));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x16 acc[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = zero16();
        // piece of (plane, chunk kc, this lane's half, tile row r + dy, pixel 32 seg + n + dx), halo coordinates
        auto piece = [&](int plane, int kc, int r, int seg, int dy, int dx) {
            return tl[((plane * KC + kc) * 2 + kh) * SEL + (r + dy) * PXW + seg * 32 + n + dx];
        };
        constexpr int LO = NPL - 1;                  // the second plane (single plane: the same piece, a dead value)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int ks = tap * KC + kc;
                const V8 ah = wl[ks * 64 + lane], al = wl[(LO * NKS + ks) * 64 + lane];
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int wt = wave * TPW + i, r = wt / NXS, seg = wt % NXS;
                    acc[i] = mma(ah, al, piece(0, kc, r, seg, tap / 3, tap % 3), piece(LO, kc, r, seg, tap / 3, tap % 3), acc[i]);
                }
            }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int wt = wave * TPW + i, r = wt / NXS, seg = wt % NXS;
            const int y = y0 + r, x = x0 + seg * 32 + n;
            f32x16 res = zero16();
            if constexpr (RES == 2) {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
                    res = mma(wl[(NPL * NKS + kc) * 64 + lane], wl[(NPL * NKS + LO * KC + kc) * 64 + lane], piece(0, kc, r, seg, 1, 1),
                              piece(LO, kc, r, seg, 1, 1), res);
            }
            if (y >= H) continue;
            const size_t pix = ((size_t)b * H + y) * W + x;
#pragma unroll
            for (int q = 0; q < COUT / 8; ++q) {
                const int c0 = 8 * q + 4 * kh;          // this lane's 4 consecutive output channels of register group q
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][4 * q + e] + bias[4 * q + e], 0.f);
                if constexpr (RES == 1) {               // input channels c0 .. c0+3 of the centre pixel: half kh of piece (q >> 1, q & 1)
                    const int pc = ((q >> 1) * 2 + (q & 1)) * SEL + (r + 1) * PXW + seg * 32 + n + 1;
                    const V4 xh = reinterpret_cast<const V4*>(tl + pc)[kh];
                    const V4 xl = reinterpret_cast<const V4*>(tl + LO * 2 * KC * SEL + pc)[kh];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += val4(xh, xl, e);
                }
                if constexpr (RES == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += res[4 * q + e] + rbias[4 * q + e];
                }
                put4(v, out_hi + pix * COUT + c0, out_lo + pix * COUT + c0);
            }
        }
        __syncthreads();          // everyone is done with the tile before the next fill lands
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv1 (1 -> 16 -> 16 -> 16 at full mel resolution) as ONE kernel: the three layers' activations of an 8 x 64-pixel
// output tile never leave the CU.  Separately they write / read 708 MB each per 16 clips (3.9 GB for the three layers);
// fused, HBM sees the mel (44 MB) and the third layer's output.  A persistent 8-wave workgroup per CU:
//   A: conv1.0 on the tile grown by 2 pixels (12 x 68), from mel (global, 9 taps per lane)       -> LDS tile TA
//   B: conv1.1 on the tile grown by 1 (10 x 66), taps + identity residual from TA                -> LDS tile TB
//   C: conv1.2 on the tile, taps + residual from TB                                              -> HBM
// Reflect padding: a position outside the image holds the layer's value AT THE REFLECTED POSITION, so the grown tiles are
// computed at reflected coordinates (a lane reads its taps around the reflected position's place in the tile below).
// 16 output channels = one v_mfma_f32_16x16x32_bf16 row block; a k-step holds two taps x 16 channels (lane group q4: tap
// 2 ks + (q4 >> 1), channels 8 (q4 & 1) ..): 5 k-steps x 3 split products per 16 pixels, none of it padding rows.
// The D tile has the pixel on the lane and channels 4 q4 .. 4 q4 + 3 in registers: one 8-byte piece per plane, the four
// lane groups complete a pixel's 32 bytes in one store instruction.
// ---------------------------------------------------------------------------------------------------------
DEV f32x4 mfma16b(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
DEV f32x4 mfma16b(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
DEV f32x4 mma_16(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 acc) {
    acc = mfma16b(ah, bh, acc);
    acc = mfma16b(al, bh, acc);
    return mfma16b(ah, bl, acc);
}
DEV f32x4 mma_16(f16x8 ah, f16x8, f16x8 bh, f16x8, f32x4 acc) { return mfma16b(ah, bh, acc); }
// SP = false (one fp16 plane): 51 KiB of LDS and two workgroups per CU - one's phase A (vector ALU) beside the other's MFMA phases.
template <int ROWS, int TX, bool SP>
__global__ __launch_bounds__(512, SP ? 2 : 4) void k_me_stem(const float* __restrict__ mel, typename PL<SP>::e* __restrict__ out_hi,
                                                               typename PL<SP>::e* __restrict__ out_lo,
                                                    const typename PL<SP>::v8* __restrict__ w /* split: 22 fragments (conv1.0: hi, lo; conv1.1: 5 hi, 5 lo; conv1.2: 5 hi, 5 lo); single plane: 10 (conv1.1: 5, conv1.2: 5) */,
                                                    const float* __restrict__ bias /*[3][16]*/,
                                                    const float* __restrict__ wa32 /* conv1.0 folded weights [16][9] */, int H, int W, int ntiles) {
    using E = typename PL<SP>::e;
    using V8 = typename PL<SP>::v8;
    using V4 = typename PL<SP>::v4;
    constexpr int NPL = PL<SP>::N, LO = NPL - 1;
    constexpr int RAH = ROWS + 4, RAW = TX + 4, NA = RAH * RAW;       // conv1.0 region
    constexpr int RBH = ROWS + 2, RBW = TX + 2, NB = RBH * RBW;       // conv1.1 region
    constexpr int NBP = (NB + 15) / 16 * 16;                          // plane stride of TB: a multiple of 256 B keeps the reads conflict-free
    constexpr int NC = ROWS * TX;
    constexpr int MH = ROWS + 6, MW = TX + 6, NM = MH * MW;           // mel region; positions outside the image hold the reflected values
    static_assert(NA % 16 == 0, "plane stride of TA");
    static_assert(NM <= 1024, "two mel values per thread");
    static_assert(TX == 64 && NC % 128 == 0, "phase C walks two rows per step");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* TA = lds;                                // pieces [plane][half][NA] x 16 B
    char* TB = TA + 2 * NPL * NA * 16;             // pieces [plane][half][NBP] x 16 B
    float* MT = reinterpret_cast<float*>(TB + 2 * NPL * NBP * 16);
    float* WA = MT + NM;                           // single plane: conv1.0's 144 weights (two workgroups per CU leave 128 registers per lane)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = lane & 15, q4 = lane >> 4;
    // conv1.1 / conv1.2 weight fragments stay in registers (the B operands are the only LDS reads of the MFMA phases)
    V8 wb_h[5], wb_l[5], wc_h[5], wc_l[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        if constexpr (SP) {
            wb_h[ks] = w[(2 + ks) * 64 + lane];
            wb_l[ks] = w[(7 + ks) * 64 + lane];
            wc_h[ks] = w[(12 + ks) * 64 + lane];
            wc_l[ks] = w[(17 + ks) * 64 + lane];
        } else {
            wb_h[ks] = wb_l[ks] = w[ks * 64 + lane];
            wc_h[ks] = wc_l[ks] = w[(5 + ks) * 64 + lane];
        }
    }
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + 16 + 4 * q4), bc = *reinterpret_cast<const f32x4*>(bias + 32 + 4 * q4);
    // phase A runs on the vector ALU in fp32: thread = (pixel, channel quad); its quad's 36 weights in registers
    const int quad = tid & 3;
    float wa[4][9];
    if constexpr (SP) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < 9; ++k) wa[c][k] = wa32[(4 * quad + c) * 9 + k];
    } else if (tid < 144) {
        WA[tid] = wa32[tid];
    }
    const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + 4 * quad);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // k-step ks of this lane: tap 2 ks + (q4 >> 1) (the ninth tap's partner slot has zero weights: it re-reads tap 8), channel half q4 & 1
    int ckA[5], ckB[5];          // piece offsets relative to the centre piece (TA: centre = the position itself; TB: centre = tile origin)
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = min(2 * ks + (q4 >> 1), 8);
        ckA[ks] = ((q4 & 1) * NA + (tap / 3 - 1) * RAW + tap % 3 - 1) * 16;
        ckB[ks] = ((q4 & 1) * NBP + (tap / 3) * RBW + tap % 3) * 16;
    }
    const int hp = (q4 >> 1), ho = (q4 & 1) * 8;         // this lane's output channels 4 q4 .. +3: half hp, byte offset ho of the piece
    const int txn = W / TX, tyn = (H + ROWS - 1) / ROWS;
    // mel region of tile t: thread i holds elements i and i + 512
    auto mel_fetch = [&](int t, float (&mv)[2]) {
        const int tx = t % txn, ty = (t / txn) % tyn, b = t / (txn * tyn);
        const float* mb = mel + (size_t)b * H * W;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 512 * k;
            mv[k] = 0.f;
            if (i < NM) mv[k] = mb[(size_t)reflect(ty * ROWS - 3 + i / MW, H) * W + reflect(tx * TX - 3 + i % MW, W)];
        }
    };
    float mv[2];
    if ((int)blockIdx.x < ntiles) mel_fetch(blockIdx.x, mv);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % txn, ty = (t / txn) % tyn, b = t / (txn * tyn);
        const int x0 = tx * TX, y0 = ty * ROWS;
        MT[tid] = mv[0];
        if (tid + 512 < NM) MT[tid + 512] = mv[1];
        __syncthreads();            // also: everyone has left the previous tile's phase C (TB) and phase A (MT)
        if (t + (int)gridDim.x < ntiles) mel_fetch(t + gridDim.x, mv);      // lands during the three phases
        // local coordinate of image position (tile origin - g + i) reflected into the image, in a region that starts at origin - G
        auto local = [](int i, int o0, int g, int G, int size, int lim) {
            const int r = reflect(o0 - g + i, size) - (o0 - G);
            return min(max(r, 1), lim - 2);
        };
        // ---- A: conv1.0 at the reflected positions of the 2-pixel-grown tile (fp32 FMAs; taps are plain neighbours in MT)
#ifndef DC_DIAG_STEM_SKIP
#define DC_DIAG_STEM_SKIP 0          // diagnostic builds (timing only, results invalid): bit 0 / 1 / 2 = phase A / B / C does not run
#endif
        if (!(DC_DIAG_STEM_SKIP & 1)) {
            if constexpr (!SP) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < 9; ++k) wa[c][k] = WA[(4 * quad + c) * 9 + k];
            }
            int row = (tid >> 2) / RAW, px = (tid >> 2) % RAW;
            for (int p = tid >> 2; p < NA; p += 128) {
                const int cy = local(row, y0, 2, 3, H, MH), cx = local(px, x0, 2, 3, W, MW);
                const float* m0 = MT + cy * MW + cx;
                f32x4 acc = ba;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float mval = m0[(k / 3 - 1) * MW + k % 3 - 1];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(wa[c][k], mval, acc[c]);
                }
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[c], 0.f);
                char* dst = TA + (size_t)(((quad >> 1) * NA + p) * 16 + (quad & 1) * 8);
                put4(v, reinterpret_cast<E*>(dst), reinterpret_cast<E*>(dst + 2 * NA * 16));
                px += 128 - RAW;
                row += 1;
                if (px >= RAW) {
                    px -= RAW;
                    row += 1;
                }
            }
        }
        __syncthreads();
        // ---- B: conv1.1 at the reflected positions of the 1-pixel-grown tile, from TA
        if (!(DC_DIAG_STEM_SKIP & 2)) {
            int row = (wave * 16 + n) / RBW, px = (wave * 16 + n) % RBW;
            for (int f = wave * 16 + n; f < NBP; f += 128) {
                const int rr = min(row, RBH - 1);
                const int ly = local(rr, y0, 1, 2, H, RAH), lx = local(px, x0, 1, 2, W, RAW);
                const char* ctr = TA + (size_t)(ly * RAW + lx) * 16;
                f32x4 acc = z4;
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    const V8 bh = *reinterpret_cast<const V8*>(ctr + ckA[ks]);
                    const V8 bl = *reinterpret_cast<const V8*>(ctr + ckA[ks] + LO * 2 * NA * 16);
                    acc = mma_16(wb_h[ks], wb_l[ks], bh, bl, acc);
                }
                const V4 rh = *reinterpret_cast<const V4*>(ctr + hp * NA * 16 + ho);
                const V4 rl = *reinterpret_cast<const V4*>(ctr + (LO * 2 + hp) * NA * 16 + ho);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[e] + bb[e], 0.f) + val4(rh, rl, e);
                if (f < NB) {
                    char* dst = TB + (size_t)((hp * NBP + f) * 16 + ho);
                    put4(v, reinterpret_cast<E*>(dst), reinterpret_cast<E*>(dst + 2 * NBP * 16));
                }
                px += 128 - RBW;
                row += 1;
                if (px >= RBW) {
                    px -= RBW;
                    row += 1;
                }
            }
        }
        __syncthreads();
        // ---- C: conv1.2 on the tile, from TB (its halo already holds the reflected values); two tile rows per step
        if (!(DC_DIAG_STEM_SKIP & 4)) {
            const int row0 = wave >> 2, px = (wave & 3) * 16 + n;
            const char* org = TB + (size_t)(row0 * RBW + px) * 16;
            size_t pix = ((size_t)b * H + y0 + row0) * W + x0 + px;
            constexpr int UNR = SP ? ROWS / 2 : 1;          // (single plane: 128 registers per lane)
#pragma unroll(UNR)
            for (int k = 0; k < ROWS / 2; ++k) {
                f32x4 acc = z4;
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    const V8 bh = *reinterpret_cast<const V8*>(org + ckB[ks]);
                    const V8 bl = *reinterpret_cast<const V8*>(org + ckB[ks] + LO * 2 * NBP * 16);
                    acc = mma_16(wc_h[ks], wc_l[ks], bh, bl, acc);
                }
                const char* ctr = org + (RBW + 1) * 16;
                const V4 rh = *reinterpret_cast<const V4*>(ctr + hp * NBP * 16 + ho);
                const V4 rl = *reinterpret_cast<const V4*>(ctr + (LO * 2 + hp) * NBP * 16 + ho);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[e] + bc[e], 0.f) + val4(rh, rl, e);
                if (y0 + row0 + 2 * k < H) put4(v, out_hi + pix * 16 + 4 * q4, out_lo + pix * 16 + 4 * q4);
                org += 2 * RBW * 16;
                pix += 2 * (size_t)W;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv2 (16 -> 32 with the 1x1-convolution residual, then 32 -> 32 with the identity residual, at 5400 x 64) as ONE kernel, built
// like k_me_stem: a persistent 8-wave workgroup per CU keeps the 32-channel intermediate of an 8 x 32-pixel tile (grown by one
// pixel) in LDS, so its 1.4 GB per 32 clips are neither written nor read back.
//   fill: the input tile grown by 2 (12 x 36 pixels, both planes) by LDS-DMA, reflect padding by index          -> TA
//   B:    conv2.0 + residual at the reflected positions of the 1-pixel-grown tile (10 x 34), taps from TA        -> TB
//   C:    conv2.1 + identity residual on the tile, taps from TB; the next tile's fill is in flight meanwhile     -> HBM
// v_mfma_f32_32x32x16_bf16 (32 output channels = the tile's rows), weights of both layers in LDS (56 KiB).
// ---------------------------------------------------------------------------------------------------------
// SP = false (one fp16 plane): 64 KiB of LDS, two workgroups per CU.
template <int ROWS, int TX, bool SP>
__global__ __launch_bounds__(512, SP ? 2 : 4) void k_me_mid(const typename PL<SP>::v8* __restrict__ in_hi, const typename PL<SP>::v8* __restrict__ in_lo,
                                                   typename PL<SP>::e* __restrict__ out_hi, typename PL<SP>::e* __restrict__ out_lo,
                                                   const typename PL<SP>::v8* __restrict__ w0 /* conv2.0: 9 hi, 9 lo, residual hi, lo (single plane: 9, residual) */,
                                                   const float* __restrict__ bias0_ft, const float* __restrict__ rbias0_ft,
                                                   const typename PL<SP>::v8* __restrict__ w1 /* conv2.1: 18 hi, 18 lo (single plane: 18) */,
                                                   const float* __restrict__ bias1_ft, int H, int W, int ntiles) {
    using V8 = typename PL<SP>::v8;
    using V4 = typename PL<SP>::v4;
    using E = typename PL<SP>::e;
    constexpr int NPL = PL<SP>::N, LO = NPL - 1;
    static_assert(TX == 32 && ROWS == 8, "phase C: one tile row per wave");
    constexpr int RAH = ROWS + 4, RAW = TX + 4, NA = RAH * RAW;           // input region (16 channels: halves kh)
    constexpr int RBH = ROWS + 2, RBW = TX + 2, NB = RBH * RBW;           // conv2.0 region (32 channels: chunks kc, halves kh)
    constexpr int NBP = (NB + 15) / 16 * 16;
    constexpr int NPA = 2 * NPL * NA;                                      // 16-byte pieces of TA: [plane][kh][NA]
    constexpr int NDMA = (NPA + 63) / 64;
    constexpr int NW0 = 10 * NPL, NW1 = 18 * NPL;
    static_assert(NA % 16 == 0, "plane stride of TA");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    V8* wl0 = reinterpret_cast<V8*>(lds);                                  // 20 (10) fragments
    V8* wl1 = wl0 + NW0 * 64;                                              // 36 (18) fragments
    V8* TA = wl1 + NW1 * 64;                                               // pieces [plane][kh][NA]
    V8* TB = TA + NDMA * 64;                                               // pieces [plane][kc][kh][NBP]
    float* BL = reinterpret_cast<float*>(TB + NPL * 4 * NBP);              // single plane: the three bias vectors (128 registers per lane)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = lane & 31, kh = lane >> 5;
    for (int f = wave; f < NW0; f += 8) wl0[f * 64 + lane] = w0[f * 64 + lane];
    for (int f = wave; f < NW1; f += 8) wl1[f * 64 + lane] = w1[f * 64 + lane];      // (in registers instead: 4.36 vs 4.25 ms per 32 clips)
    f32x16 bias0, rbias0, bias1;
    if constexpr (SP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bias0[r] = bias0_ft[kh * 16 + r];
            rbias0[r] = rbias0_ft[kh * 16 + r];
            bias1[r] = bias1_ft[kh * 16 + r];
        }
    } else if (tid < 96) {
        BL[tid] = tid < 32 ? bias0_ft[tid] : tid < 64 ? rbias0_ft[tid - 32] : bias1_ft[tid - 64];
    }
    auto ld16 = [&](int which) {                                           // one bias vector of this lane's half from LDS
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(BL + which * 32 + kh * 16 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = t4[e];
        }
        return v;
    };
    const int txn = W / TX, tyn = (H + ROWS - 1) / ROWS;
    auto fill = [&](int t) {
        const int tx = t % txn, ty = (t / txn) % tyn, b = t / (txn * tyn);
        for (int j = wave; j < NDMA; j += 8) {
            const int p = 64 * j + lane;
            if (p < NPA) {
                const int idx = p % NA, sel = p / NA;                      // sel = plane * 2 + kh
                const int yy = reflect(ty * ROWS - 2 + idx / RAW, H), xx = reflect(tx * TX - 2 + idx % RAW, W);
                const V8* src = ((sel >> 1) ? in_lo : in_hi) + (((size_t)b * H + yy) * W + xx) * 2 + (sel & 1);
                me_dma16(src, reinterpret_cast<const char*>(TA + 64 * j));
            }
        }
    };
    auto local = [](int i, int o0, int g, int G, int size, int lim) {      // as in k_me_stem
        const int r = reflect(o0 - g + i, size) - (o0 - G);
        return min(max(r, 1), lim - 2);
    };
    if ((int)blockIdx.x < ntiles) fill(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % txn, ty = (t / txn) % tyn, b = t / (txn * tyn);
        const int x0 = tx * TX, y0 = ty * ROWS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                // TA has landed; everyone has left the previous tile's phase C (TB)
        // ---- B: conv2.0 (+ 1x1 residual) at the reflected positions of the 1-pixel-grown tile
        for (int f0 = wave * 32; f0 < NB; f0 += 256) {
            const int f = f0 + n, fr = min(f, NB - 1);
            const int row = fr / RBW, px = fr % RBW;
            const int ly = local(row, y0, 1, 2, H, RAH), lx = local(px, x0, 1, 2, W, RAW);
            const V8* ctr = TA + kh * NA + ly * RAW + lx;
            f32x16 acc = zero16();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int o = (tap / 3 - 1) * RAW + tap % 3 - 1;
                acc = mma(wl0[tap * 64 + lane], wl0[(LO * 9 + tap) * 64 + lane], ctr[o], ctr[LO * 2 * NA + o], acc);
            }
            const f32x16 res = mma(wl0[NPL * 9 * 64 + lane], wl0[(NPL * 9 + LO) * 64 + lane], ctr[0], ctr[LO * 2 * NA], zero16());
            if constexpr (!SP) {
                bias0 = ld16(0);
                rbias0 = ld16(1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {           // output channels 8 q + 4 kh .. + 3: chunk q >> 1, half q & 1, byte offset 8 kh
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * q + e] + bias0[4 * q + e], 0.f) + res[4 * q + e] + rbias0[4 * q + e];
                if (f < NB) {
                    V8* dst = TB + ((q >> 1) * 2 + (q & 1)) * NBP + f;
                    put4(v, reinterpret_cast<E*>(dst) + 4 * kh, reinterpret_cast<E*>(dst + 4 * NBP) + 4 * kh);
                }
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fill(t + gridDim.x);       // TA is free: the next tile's input lands during phase C
        // ---- C: conv2.1 + identity residual: wave = tile row
        {
            const int row = wave;
            const V8* org = TB + kh * NBP + row * RBW + n;          // piece of (kc 0, this half, halo row `row`, halo column n)
            f32x16 acc = zero16();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int ks = tap * 2 + kc, o = kc * 2 * NBP + (tap / 3) * RBW + tap % 3;
                    acc = mma(wl1[ks * 64 + lane], wl1[(LO * 18 + ks) * 64 + lane], org[o], org[LO * 4 * NBP + o], acc);
                }
            const int y = y0 + row;
            if constexpr (!SP) bias1 = ld16(2);
            if (y < H) {
                const size_t pix = ((size_t)b * H + y) * W + x0 + n;
                const V8* ctr = TB + (row + 1) * RBW + n + 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const V4 xh = reinterpret_cast<const V4*>(ctr + ((q >> 1) * 2 + (q & 1)) * NBP)[kh];
                    const V4 xl = reinterpret_cast<const V4*>(ctr + (LO * 4 + (q >> 1) * 2 + (q & 1)) * NBP)[kh];
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * q + e] + bias1[4 * q + e], 0.f) + val4(xh, xl, e);
                    put4(v, out_hi + pix * 32 + 8 * q + 4 * kh, out_lo + pix * 32 + 8 * q + 4 * kh);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// max_pool2d on the plane pair; padding never wins (torch pads with -inf).  One thread per (pixel, 8 channels).
// hi + lo is exact in fp32 (<= 17 significant bits), so the maximum re-splits into the very planes it came from.
// ---------------------------------------------------------------------------------------------------------
// Sliding window along time: a thread owns (clip, output column, 8 channels) and walks NY consecutive output rows,
// keeping the last KH row maxima (each the maximum over the KW taps of one input row) in registers; an output row
// costs SH new input rows instead of KH (5x5 stride 1: 5x fewer loads than one thread per output).
// XO adjacent output columns per thread: their windows overlap ((XO - 1) SW + KW input columns instead of XO KW) - two columns need 7
// loads per plane and input row instead of 10, but 173 VGPRs instead of 48 cost the occupancy that hides the loads: no faster (round 5).
template <int KH, int KW, int SH, int SW, int PH, int PW, int XO>
__global__ __launch_bounds__(256) void k_me_pool(const bf16x8* __restrict__ in_hi, const bf16x8* __restrict__ in_lo,
                                                 bf16x8* __restrict__ out_hi, bf16x8* __restrict__ out_lo, int Bc, int H, int W,
                                                 int C8, int Ho, int Wo, int NY) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int strips = (Ho + NY - 1) / NY;
    const int Wq = (Wo + XO - 1) / XO;
    const long long total = (long long)Bc * strips * Wq * C8;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int xo0 = (int)((idx / C8) % Wq) * XO;
    const int sy = (int)((idx / ((long long)C8 * Wq)) % strips);
    const int b = (int)(idx / ((long long)C8 * Wq * strips));
    constexpr int NC = (XO - 1) * SW + KW;      // input columns the thread's XO windows cover
    float ring[KH][XO][8];                      // ring[k][i] = row maximum of input row (next_row - KH + k) over output i's window, oldest first
    auto row_max = [&](int yy, float (&m)[XO][8]) {
#pragma unroll
        for (int i = 0; i < XO; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) m[i][j] = -INFINITY;
        if (yy < 0 || yy >= H) return;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int xx = xo0 * SW - PW + c;
            if (xx < 0 || xx >= W) continue;
            const size_t o = (((size_t)b * H + yy) * W + xx) * C8 + c8;
            const bf16x8 h = in_hi[o], l = in_lo[o];
#pragma unroll
            for (int i = 0; i < XO; ++i)
                if (c >= i * SW && c < i * SW + KW) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[i][j] = fmaxf(m[i][j], (float)h[j] + (float)l[j]);
                }
        }
    };
    const int yo0 = sy * NY, yo1 = min(yo0 + NY, Ho);
#pragma unroll
    for (int k = 0; k < KH; ++k) row_max(yo0 * SH - PH + k, ring[k]);
    for (int yo = yo0; yo < yo1; ++yo) {
#pragma unroll
        for (int i = 0; i < XO; ++i) {
            if (xo0 + i >= Wo) continue;
            float m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                m[j] = ring[0][i][j];
#pragma unroll
                for (int k = 1; k < KH; ++k) m[j] = fmaxf(m[j], ring[k][i][j]);
            }
            bf16x8 oh, ol;
            split8(m, oh, ol);
            const size_t oidx = (((size_t)b * Ho + yo) * Wo + xo0 + i) * C8 + c8;
            out_hi[oidx] = oh;
            out_lo[oidx] = ol;
        }
        if (yo + 1 < yo1) {              // advance the window by SH input rows
#pragma unroll
            for (int k = 0; k + SH < KH; ++k)
#pragma unroll
                for (int i = 0; i < XO; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) ring[k][i][j] = ring[k + SH][i][j];
#pragma unroll
            for (int k = (KH > SH ? KH - SH : 0); k < KH; ++k) row_max((yo + 1) * SH - PH + k, ring[k]);
        }
    }
}
// The same sliding window on ONE fp16 plane: the maximum of fp16 values is one of them, so the window works on the packed values
// (v_pk_max_f16: 4 instructions per 8 channels and tap, 4 registers per row maximum) and the output is exact.
template <int KH, int KW, int SH, int SW, int PH, int PW>
__global__ __launch_bounds__(256) void k_me_pool16(const f16x8* __restrict__ in, f16x8* __restrict__ out, int Bc, int H, int W, int C8, int Ho,
                                                   int Wo, int NY) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int strips = (Ho + NY - 1) / NY;
    const long long total = (long long)Bc * strips * Wo * C8;
    if (idx >= total) return;
    const int c8 = (int)(idx % C8);
    const int xo = (int)((idx / C8) % Wo);
    const int sy = (int)((idx / ((long long)C8 * Wo)) % strips);
    const int b = (int)(idx / ((long long)C8 * Wo * strips));
    f16x8 ninf;
#pragma unroll
    for (int j = 0; j < 8; ++j) ninf[j] = (_Float16)(-INFINITY);
    f16x8 ring[KH];
    auto row_max = [&](int yy) {
        f16x8 m = ninf;
        if (yy < 0 || yy >= H) return m;
#pragma unroll
        for (int c = 0; c < KW; ++c) {
            const int xx = xo * SW - PW + c;
            if (xx < 0 || xx >= W) continue;
            m = __builtin_elementwise_max(m, in[(((size_t)b * H + yy) * W + xx) * C8 + c8]);
        }
        return m;
    };
    const int yo0 = sy * NY, yo1 = min(yo0 + NY, Ho);
#pragma unroll
    for (int k = 0; k < KH; ++k) ring[k] = row_max(yo0 * SH - PH + k);
    for (int yo = yo0; yo < yo1; ++yo) {
        f16x8 m = ring[0];
#pragma unroll
        for (int k = 1; k < KH; ++k) m = __builtin_elementwise_max(m, ring[k]);
        out[(((size_t)b * Ho + yo) * Wo + xo) * C8 + c8] = m;
        if (yo + 1 < yo1) {
#pragma unroll
            for (int k = 0; k + SH < KH; ++k) ring[k] = ring[k + SH];
#pragma unroll
            for (int k = (KH > SH ? KH - SH : 0); k < KH; ++k) ring[k] = row_max((yo + 1) * SH - PH + k);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv4: Conv1d(512 -> 64, k=1) + BatchNorm1d over the flattened (channel, bin) axis.  The planes hold a frame's 512
// features in (bin, channel) order, so the host permutes the weight columns instead of moving data.
// One wave = 32 frames; weights (natural-k fragments [ot][ks], hi then lo) stream from L2.
// Single fp16 plane: the weights stay split (fp16 hi + lo, two products per k-step) - the layer is 1 % of the encoder.
// ---------------------------------------------------------------------------------------------------------
template <bool SP>
__global__ __launch_bounds__(256) void k_me_head(const typename PL<SP>::v8* __restrict__ in_hi, const typename PL<SP>::v8* __restrict__ in_lo,
                                                 const typename PL<SP>::v8* __restrict__ w4, const float* __restrict__ bias_ft,
                                                 float* __restrict__ xf_out, int M) {
    using V8 = typename PL<SP>::v8;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, kh = lane >> 5;
    const int tok = (blockIdx.x * 4 + wave) * 32 + n;
    const size_t t = tok < M ? tok : M - 1;
    f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll 4
    for (int ks = 0; ks < 32; ++ks) {
        const V8 bh = in_hi[t * 64 + 2 * ks + kh];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            const V8 ah = w4[(ot * 32 + ks) * 64 + lane], al = w4[(64 + ot * 32 + ks) * 64 + lane];
            if constexpr (SP) {
                acc[ot] = mma3(ah, al, bh, in_lo[t * 64 + 2 * ks + kh], acc[ot]);
            } else {
                acc[ot] = mfma(ah, bh, acc[ot]);
                acc[ot] = mfma(al, bh, acc[ot]);
            }
        }
    }
    if (tok >= M) return;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[ot][4 * q + e] + bias_ft[(ot * 2 + kh) * 16 + 4 * q + e];
            *reinterpret_cast<f32x4*>(xf_out + (size_t)tok * 64 + 32 * ot + 8 * q + 4 * kh) = v;
        }
}

// proj: Linear(64 -> 64) on the fp32 features (transformer.py:457), operands split in registers.
__global__ __launch_bounds__(256) void k_me_proj(const float* __restrict__ xf, const bf16x8* __restrict__ wp,
                                                 const float* __restrict__ bias_ft, float* __restrict__ xf_proj, int M) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 31, kh = lane >> 5;
    const int tok = (blockIdx.x * 4 + wave) * 32 + n;
    const size_t t = tok < M ? tok : M - 1;
    f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float v[8];
        const f32x4 a = *reinterpret_cast<const f32x4*>(xf + t * 64 + 16 * ks + 8 * kh);
        const f32x4 c = *reinterpret_cast<const f32x4*>(xf + t * 64 + 16 * ks + 8 * kh + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = a[e];
            v[4 + e] = c[e];
        }
        bf16x8 bh, bl;
        split8(v, bh, bl);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
            acc[ot] = mma3(wp[(ot * 4 + ks) * 64 + lane], wp[(8 + ot * 4 + ks) * 64 + lane], bh, bl, acc[ot]);
    }
    if (tok >= M) return;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[ot][4 * q + e] + bias_ft[(ot * 2 + kh) * 16 + 4 * q + e];
            *reinterpret_cast<f32x4*>(xf_proj + (size_t)tok * 64 + 32 * ot + 8 * q + 4 * kh) = v;
        }
}

// ---- host side ------------------------------------------------------------------------------------------
inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline int tile_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
inline uint16_t f2h(float f) {   // fp32 -> fp16 bits, round to nearest even (the compiler's conversion)
    const _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
inline float h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

// natural-k A fragments of Wm [n_out][k_in] (row-major): frag (ot, ks), lane (i = l & 31, hh = l >> 5), element j
//   = Wm[32 ot + i][16 ks + 8 hh + j]; order [hi: ot][ks] then [lo: ot][ks]; fmt 0: bf16 hi + lo, 1: fp16 hi only, 2: fp16 hi + lo
std::vector<uint16_t> pack_nat(const std::vector<float>& Wm, int n_out, int k_in, int OT, int KS, int fmt = 0) {
    const size_t ne = (size_t)OT * KS * 512;
    std::vector<uint16_t> out((fmt == 1 ? 1 : 2) * ne, 0);
    for (int ot = 0; ot < OT; ++ot)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int row = 32 * ot + (l & 31), col = 16 * ks + 8 * (l >> 5) + j;
                    const float v = (row < n_out && col < k_in) ? Wm[(size_t)row * k_in + col] : 0.f;
                    const size_t o = (((size_t)ot * KS + ks) * 64 + l) * 8 + j;
                    const uint16_t h = fmt ? f2h(v) : f2bf(v);
                    out[o] = h;
                    if (fmt != 1) out[ne + o] = fmt ? f2h(v - h2f(h)) : f2bf(v - bf2f(h));
                }
    return out;
}
std::vector<float> ftvec(const std::vector<float>& v, int NT) {
    std::vector<float> out((size_t)NT * 32, 0.f);
    for (int t = 0; t < NT; ++t)
        for (int hh = 0; hh < 2; ++hh)
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * t + tile_row(r, hh);
                out[(t * 2 + hh) * 16 + r] = f < (int)v.size() ? v[f] : 0.f;
            }
    return out;
}

struct ConvDev {
    const bf16x8* w = nullptr;            // split planes: bf16 hi + lo fragments
    const f16x8* w16 = nullptr;           // single plane: fp16 fragments (conv1.1 ... conv3.1)
    const float *bias = nullptr, *rbias = nullptr;
};

}  // namespace

struct dc_music {
    uint8_t* arena = nullptr;
    ConvDev conv[7];
    const bf16x8 *w4 = nullptr, *wp = nullptr;
    const f16x8* w4_16 = nullptr;         // conv4 as fp16 hi + lo fragments (single-plane format)
    const float *b4 = nullptr, *bp = nullptr;
    const bf16x8* stem_w = nullptr;       // k_me_stem: 22 fragments (conv1.0: hi, lo; conv1.1: 5 hi, 5 lo; conv1.2: 5 hi, 5 lo)
    const f16x8* stem_w16 = nullptr;      // single-plane format: 10 fp16 fragments (conv1.1: 5, conv1.2: 5)
    const float* stem_b = nullptr;        // [3][16] folded biases
    const float* stem_wa = nullptr;       // conv1.0 folded weights [16][9] fp32 (that layer runs on the vector ALU)
    // ping-pong plane pairs, sized for `cap` clips of `cap_tm` mel frames
    bf16x8 *a_hi = nullptr, *a_lo = nullptr, *b_hi = nullptr, *b_lo = nullptr;
    int cap = 0, cap_tm = 0;
    long long ws_bytes = 0;
    int format = 0;                       // 0: split bf16 planes, 1: one fp16 plane (dc_music_set_format)
};

namespace {

struct ConvSpec {
    const char* name;
    int cin, cout;
    bool res_conv;
};
const ConvSpec kConvs[7] = {{"conv1.0", 1, 16, false},  {"conv1.1", 16, 16, false}, {"conv1.2", 16, 16, false},
                            {"conv2.0", 16, 32, true},  {"conv2.1", 32, 32, false}, {"conv3.0", 32, 32, false},
                            {"conv3.1", 32, 32, false}};
constexpr float kBnEps = 1e-5f;   // nn.BatchNorm default

}  // namespace

std::vector<std::pair<std::string, size_t>> dc_music_required(int music_dim) {
    std::vector<std::pair<std::string, size_t>> r;
    const std::string me = "music_encoder.";
    for (const ConvSpec& c : kConvs) {
        const std::string p = me + c.name;
        r.push_back({p + ".conv2d_layer.0.weight", (size_t)c.cout * c.cin * 9});
        r.push_back({p + ".conv2d_layer.0.bias", (size_t)c.cout});
        for (const char* s : {"weight", "bias", "running_mean", "running_var"}) r.push_back({p + ".conv2d_layer.1." + s, (size_t)c.cout});
        if (c.res_conv) {
            r.push_back({p + ".residual.0.weight", (size_t)c.cout * c.cin});
            r.push_back({p + ".residual.0.bias", (size_t)c.cout});
            for (const char* s : {"weight", "bias", "running_mean", "running_var"}) r.push_back({p + ".residual.1." + s, (size_t)c.cout});
        }
    }
    r.push_back({me + "conv4.0.weight", (size_t)music_dim * 512});
    r.push_back({me + "conv4.0.bias", (size_t)music_dim});
    for (const char* s : {"weight", "bias", "running_mean", "running_var"}) r.push_back({me + "conv4.1." + s, (size_t)music_dim});
    r.push_back({"proj.weight", (size_t)music_dim * music_dim});
    r.push_back({"proj.bias", (size_t)music_dim});
    return r;
}

dc_music* dc_music_build(const std::map<std::string, std::vector<float>>& params, int music_dim, std::string* err) {
    if (music_dim != DC_C) {
        *err = "music encoder kernels are built for 64 output channels";
        return nullptr;
    }
    for (const auto& rq : dc_music_required(music_dim)) {
        auto it = params.find(rq.first);
        if (it == params.end()) {
            *err = "missing parameter '" + rq.first + "'";
            return nullptr;
        }
        if (it->second.size() != rq.second) {
            *err = "parameter '" + rq.first + "' has the wrong size";
            return nullptr;
        }
    }
    auto P = [&](const std::string& n) -> const std::vector<float>& { return params.find(n)->second; };
    std::vector<uint8_t> host;
    auto add = [&](const void* p, size_t bytes) {
        const size_t off = (host.size() + 255) & ~(size_t)255;
        host.resize(off + bytes);
        memcpy(host.data() + off, p, bytes);
        return off;
    };
    // eval-mode BatchNorm folded into the convolution in front of it: s = gamma / sqrt(var + eps)
    auto bn_fold = [&](const std::string& bn, const std::vector<float>& cb, std::vector<float>& scale, std::vector<float>& bias) {
        const auto &g = P(bn + ".weight"), &be = P(bn + ".bias"), &mu = P(bn + ".running_mean"), &var = P(bn + ".running_var");
        const size_t n = g.size();
        scale.resize(n);
        bias.resize(n);
        for (size_t c = 0; c < n; ++c) {
            scale[c] = g[c] / std::sqrt(var[c] + kBnEps);
            bias[c] = (cb[c] - mu[c]) * scale[c] + be[c];
        }
    };
    struct Off {
        size_t w, w16, bias, rbias;
    } off[7];
    // v_mfma_f32_16x16x32 A fragments of the fused conv1 kernel: lane (co = l & 15, q4 = l >> 4), element j
    std::vector<uint16_t> stem_hi[3], stem_lo[3], stem_16;
    std::vector<float> stem_bias, stem_wa32;
    auto stem_pack = [&](int layer, const std::vector<float>& Wm, int K, int cin) {
        const int nks = cin == 1 ? 1 : 5;
        stem_hi[layer].assign((size_t)nks * 512, 0);
        stem_lo[layer].assign((size_t)nks * 512, 0);
        for (int ks = 0; ks < nks; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int co = l & 15, q4 = l >> 4;
                    float v = 0.f;
                    if (cin == 1) {
                        const int k = 8 * q4 + j;
                        if (k < 9) v = Wm[(size_t)co * K + k];
                    } else {
                        const int tap = 2 * ks + (q4 >> 1), ci = 8 * (q4 & 1) + j;
                        if (tap < 9) v = Wm[(size_t)co * K + tap * 16 + ci];
                    }
                    const uint16_t h = f2bf(v);
                    stem_hi[layer][((size_t)ks * 64 + l) * 8 + j] = h;
                    stem_lo[layer][((size_t)ks * 64 + l) * 8 + j] = f2bf(v - bf2f(h));
                    if (cin != 1) stem_16.push_back(f2h(v));          // (layers 1, 2 in order: [layer][ks][lane][j])
                }
    };
    const std::string me = "music_encoder.";
    for (int i = 0; i < 7; ++i) {
        const ConvSpec& c = kConvs[i];
        const std::string p = me + c.name;
        std::vector<float> sc, bi;
        bn_fold(p + ".conv2d_layer.1", P(p + ".conv2d_layer.0.bias"), sc, bi);
        const auto& w = P(p + ".conv2d_layer.0.weight");                 // [cout][cin][3][3]
        const int K = c.cin >= 16 ? 9 * c.cin : 16, KS = K / 16;
        std::vector<float> Wm((size_t)c.cout * K, 0.f);
        for (int co = 0; co < c.cout; ++co)
            for (int ci = 0; ci < c.cin; ++ci)
                for (int tap = 0; tap < 9; ++tap)
                    Wm[(size_t)co * K + tap * c.cin + ci] = w[((size_t)co * c.cin + ci) * 9 + tap] * sc[co];
        std::vector<uint16_t> frags = pack_nat(Wm, c.cout, K, 1, KS);
        std::vector<uint16_t> frags16 = pack_nat(Wm, c.cout, K, 1, KS, 1);
        if (i == 0) {
            stem_wa32.resize(16 * 9);
            for (int co = 0; co < 16; ++co)
                for (int tap = 0; tap < 9; ++tap) stem_wa32[co * 9 + tap] = Wm[(size_t)co * K + tap];
        }
        if (i < 3) {
            stem_pack(i, Wm, K, c.cin);
            stem_bias.insert(stem_bias.end(), bi.begin(), bi.end());
        }
        off[i].rbias = (size_t)-1;
        std::vector<float> rb_ft;
        if (c.res_conv) {
            std::vector<float> rs, rb;
            bn_fold(p + ".residual.1", P(p + ".residual.0.bias"), rs, rb);
            const auto& rw = P(p + ".residual.0.weight");                // [cout][cin][1][1]
            std::vector<float> Rm((size_t)c.cout * c.cin);
            for (int co = 0; co < c.cout; ++co)
                for (int ci = 0; ci < c.cin; ++ci) Rm[(size_t)co * c.cin + ci] = rw[(size_t)co * c.cin + ci] * rs[co];
            const std::vector<uint16_t> rf = pack_nat(Rm, c.cout, c.cin, 1, c.cin / 16);
            frags.insert(frags.end(), rf.begin(), rf.end());
            const std::vector<uint16_t> rf16 = pack_nat(Rm, c.cout, c.cin, 1, c.cin / 16, 1);
            frags16.insert(frags16.end(), rf16.begin(), rf16.end());
            rb_ft = ftvec(rb, 1);
        }
        off[i].w = add(frags.data(), frags.size() * 2);
        off[i].w16 = add(frags16.data(), frags16.size() * 2);
        const std::vector<float> b_ft = ftvec(bi, 1);
        off[i].bias = add(b_ft.data(), b_ft.size() * 4);
        if (c.res_conv) off[i].rbias = add(rb_ft.data(), rb_ft.size() * 4);
    }
    // conv4 + BatchNorm1d; reference feature index c*16 + bin  ->  plane order bin*32 + c
    std::vector<float> s4, b4;
    bn_fold(me + "conv4.1", P(me + "conv4.0.bias"), s4, b4);
    const auto& w4 = P(me + "conv4.0.weight");
    std::vector<float> W4((size_t)64 * 512);
    for (int o = 0; o < 64; ++o)
        for (int c = 0; c < 32; ++c)
            for (int bin = 0; bin < 16; ++bin) W4[(size_t)o * 512 + bin * 32 + c] = w4[(size_t)o * 512 + c * 16 + bin] * s4[o];
    const std::vector<uint16_t> f4 = pack_nat(W4, 64, 512, 2, 32);
    const size_t o_w4 = add(f4.data(), f4.size() * 2);
    const std::vector<uint16_t> f4h = pack_nat(W4, 64, 512, 2, 32, 2);
    const size_t o_w4h = add(f4h.data(), f4h.size() * 2);
    const std::vector<float> b4_ft = ftvec(b4, 2);
    const size_t o_b4 = add(b4_ft.data(), b4_ft.size() * 4);
    std::vector<uint16_t> stem_frags;
    for (int i = 0; i < 3; ++i) {
        stem_frags.insert(stem_frags.end(), stem_hi[i].begin(), stem_hi[i].end());
        stem_frags.insert(stem_frags.end(), stem_lo[i].begin(), stem_lo[i].end());
    }
    const size_t o_stem_w = add(stem_frags.data(), stem_frags.size() * 2);
    const size_t o_stem_w16 = add(stem_16.data(), stem_16.size() * 2);
    const size_t o_stem_b = add(stem_bias.data(), stem_bias.size() * 4);
    const size_t o_stem_wa = add(stem_wa32.data(), stem_wa32.size() * 4);
    const std::vector<uint16_t> fp = pack_nat(P("proj.weight"), 64, 64, 2, 4);
    const size_t o_wp = add(fp.data(), fp.size() * 2);
    const std::vector<float> bp_ft = ftvec(P("proj.bias"), 2);
    const size_t o_bp = add(bp_ft.data(), bp_ft.size() * 4);

    dc_music* m = new dc_music();
    if (hipMalloc((void**)&m->arena, host.size()) != hipSuccess ||
        hipMemcpy(m->arena, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        *err = "device allocation/upload of the music encoder weights failed";
        dc_music_destroy(m);
        return nullptr;
    }
    m->ws_bytes = (long long)host.size();
    for (int i = 0; i < 7; ++i) {
        m->conv[i].w = reinterpret_cast<const bf16x8*>(m->arena + off[i].w);
        m->conv[i].w16 = reinterpret_cast<const f16x8*>(m->arena + off[i].w16);
        m->conv[i].bias = reinterpret_cast<const float*>(m->arena + off[i].bias);
        m->conv[i].rbias = off[i].rbias == (size_t)-1 ? nullptr : reinterpret_cast<const float*>(m->arena + off[i].rbias);
    }
    m->stem_w = reinterpret_cast<const bf16x8*>(m->arena + o_stem_w);
    m->stem_w16 = reinterpret_cast<const f16x8*>(m->arena + o_stem_w16);
    m->w4_16 = reinterpret_cast<const f16x8*>(m->arena + o_w4h);
    m->stem_b = reinterpret_cast<const float*>(m->arena + o_stem_b);
    m->stem_wa = reinterpret_cast<const float*>(m->arena + o_stem_wa);
    m->w4 = reinterpret_cast<const bf16x8*>(m->arena + o_w4);
    m->b4 = reinterpret_cast<const float*>(m->arena + o_b4);
    m->wp = reinterpret_cast<const bf16x8*>(m->arena + o_wp);
    m->bp = reinterpret_cast<const float*>(m->arena + o_bp);
    return m;
}

void dc_music_destroy(dc_music* m) {
    if (!m) return;
    for (void* p : {(void*)m->arena, (void*)m->a_hi, (void*)m->a_lo, (void*)m->b_hi, (void*)m->b_lo})
        if (p) hipFree(p);
    delete m;
}

int dc_music_frames(int Tm) { return (Tm - 1) / 3 + 1; }   // max_pool2d (5,5), stride 3, padding 2 along time
long long dc_music_workspace_bytes(const dc_music* m) { return m ? m->ws_bytes : 0; }

namespace {

template <int CIN, int COUT, int RES>
hipError_t launch_conv(hipStream_t st, const ConvDev& c, const float* mel, const bf16x8* ih, const bf16x8* il, bf16x8* oh,
                       bf16x8* ol, int Bc, int H, int W) {
    constexpr int KC = CIN >= 16 ? CIN / 16 : 1, NKS = CIN >= 16 ? 9 * KC : 1, NF = 2 * NKS + (RES == 2 ? 2 * KC : 0);
    constexpr int TPW = 4;
    const long long tiles = (long long)Bc * H * (W / 32);
    const unsigned grid = (unsigned)((tiles + 4 * TPW - 1) / (4 * TPW));
    k_me_conv<CIN, COUT, RES><<<dim3(grid), dim3(256), NF * 1024, st>>>(mel, ih, il, reinterpret_cast<__bf16*>(oh),
                                                                      reinterpret_cast<__bf16*>(ol), c.w, c.bias, c.rbias, Bc, H, W,
                                                                      TPW);
    return hipGetLastError();
}
// per-device state of the launchers: CU count and the large-LDS opt-in (hipFuncSetAttribute applies per device)
struct MeDev {
    int dev = 0, ncu = 0;
};
hipError_t me_device(MeDev& d) {
    static int ncu_of[64] = {0};
    if (hipError_t e = hipGetDevice(&d.dev)) return e;
    int& n = ncu_of[d.dev & 63];
    if (!n)
        if (hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d.dev)) return e;
    d.ncu = n;
    return hipSuccess;
}
hipError_t me_optin(const void* fn, int bytes, unsigned long long& done, int dev) {
    if (dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) return e;
    if (dev < 64) done |= 1ull << dev;
    return hipSuccess;
}
// LDS-tiled form (CIN >= 16); W must be a multiple of the tile width
template <int CIN, int COUT, int RES, int ROWS, int NXS>
hipError_t launch_conv_t(hipStream_t st, const ConvDev& c, const bf16x8* ih, const bf16x8* il, bf16x8* oh, bf16x8* ol, int Bc, int H,
                         int W) {
    constexpr int KC = CIN / 16, NKS = 9 * KC, NF = 2 * NKS + (RES == 2 ? 2 * KC : 0);
    constexpr int NP = 2 * KC * 2 * (ROWS + 2) * (32 * NXS + 2), NDMA = (NP + 63) / 64;
    constexpr int SHM = (NF + NDMA) * 1024;
    if (W % (32 * NXS) != 0) return launch_conv<CIN, COUT, RES>(st, c, nullptr, ih, il, oh, ol, Bc, H, W);
    static unsigned long long optin = 0;
    MeDev d;
    if (hipError_t e = me_device(d)) return e;
    const int ncu = d.ncu;
    const auto fn = k_me_conv_t<CIN, COUT, RES, ROWS, NXS, true>;
    if (hipError_t e = me_optin((const void*)fn, SHM, optin, d.dev)) return e;
    const long long ntiles = (long long)Bc * ((H + ROWS - 1) / ROWS) * (W / (32 * NXS));
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 2LL * ncu);
    fn<<<dim3(grid), dim3(256), SHM, st>>>(ih, il, reinterpret_cast<__bf16*>(oh), reinterpret_cast<__bf16*>(ol), c.w, c.bias, c.rbias, H, W,
                                           (int)ntiles);
    return hipGetLastError();
}
// ... on one fp16 plane (W a multiple of the tile width: the mel has 128 bins)
template <int CIN, int COUT, int RES, int ROWS, int NXS>
hipError_t launch_conv_t16(hipStream_t st, const ConvDev& c, const f16x8* in, f16x8* out, int Bc, int H, int W) {
    constexpr int KC = CIN / 16, NKS = 9 * KC, NF = NKS + (RES == 2 ? KC : 0);
    constexpr int NP = KC * 2 * (ROWS + 2) * (32 * NXS + 2), NDMA = (NP + 63) / 64;
    constexpr int SHM = (NF + NDMA) * 1024;
    if (W % (32 * NXS) != 0) return hipErrorInvalidValue;
    static unsigned long long optin = 0;
    MeDev d;
    if (hipError_t e = me_device(d)) return e;
    const auto fn = k_me_conv_t<CIN, COUT, RES, ROWS, NXS, false>;
    if (hipError_t e = me_optin((const void*)fn, SHM, optin, d.dev)) return e;
    const long long ntiles = (long long)Bc * ((H + ROWS - 1) / ROWS) * (W / (32 * NXS));
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 2LL * d.ncu);
    fn<<<dim3(grid), dim3(256), SHM, st>>>(in, nullptr, reinterpret_cast<_Float16*>(out), nullptr, c.w16, c.bias, c.rbias, H, W, (int)ntiles);
    return hipGetLastError();
}
// conv1.0 -> conv1.1 -> conv1.2 fused (W a multiple of 64); DC_ME_NO_STEM=1 keeps the three separate launches (split planes only)
template <bool SP>
hipError_t launch_stem(hipStream_t st, const dc_music* m, const float* mel, typename PL<SP>::v8* oh, typename PL<SP>::v8* ol, int Bc, int H, int W) {
    constexpr int ROWS = 8, TX = 64, NPL = PL<SP>::N;
    constexpr int NBP = ((ROWS + 2) * (TX + 2) + 15) / 16 * 16;
    constexpr int SHM = 2 * NPL * (ROWS + 4) * (TX + 4) * 16 + 2 * NPL * NBP * 16 + (ROWS + 6) * (TX + 6) * 4 + (SP ? 0 : 144 * 4);
    static unsigned long long optin = 0;
    MeDev d;
    if (hipError_t e = me_device(d)) return e;
    if (W % TX != 0) return hipErrorInvalidValue;
    if (hipError_t e = me_optin((const void*)k_me_stem<ROWS, TX, SP>, SHM, optin, d.dev)) return e;
    const long long ntiles = (long long)Bc * ((H + ROWS - 1) / ROWS) * (W / TX);
    const unsigned grid = (unsigned)std::min<long long>(ntiles, (SP ? 1 : 2) * (long long)d.ncu);
    using E = typename PL<SP>::e;
    const typename PL<SP>::v8* w;
    if constexpr (SP) w = m->stem_w; else w = m->stem_w16;
    k_me_stem<ROWS, TX, SP><<<dim3(grid), dim3(512), SHM, st>>>(mel, reinterpret_cast<E*>(oh), reinterpret_cast<E*>(ol), w, m->stem_b, m->stem_wa, H,
                                                                W, (int)ntiles);
    return hipGetLastError();
}
// conv2.0 -> conv2.1 fused (W a multiple of 32); DC_ME_NO_MID=1 keeps the two separate launches (split planes only)
template <bool SP>
hipError_t launch_mid(hipStream_t st, const dc_music* m, const typename PL<SP>::v8* ih, const typename PL<SP>::v8* il, typename PL<SP>::v8* oh,
                      typename PL<SP>::v8* ol, int Bc, int H, int W) {
    constexpr int ROWS = 8, TX = 32, NPL = PL<SP>::N;
    constexpr int NPA = 2 * NPL * (ROWS + 4) * (TX + 4), NDMA = (NPA + 63) / 64, NBP = ((ROWS + 2) * (TX + 2) + 15) / 16 * 16;
    constexpr int SHM = 28 * NPL * 1024 + NDMA * 1024 + 4 * NPL * NBP * 16 + (SP ? 0 : 96 * 4);
    static unsigned long long optin = 0;
    MeDev d;
    if (hipError_t e = me_device(d)) return e;
    if (W % TX != 0) return hipErrorInvalidValue;
    if (hipError_t e = me_optin((const void*)k_me_mid<ROWS, TX, SP>, SHM, optin, d.dev)) return e;
    const long long ntiles = (long long)Bc * ((H + ROWS - 1) / ROWS) * (W / TX);
    const unsigned grid = (unsigned)std::min<long long>(ntiles, (SP ? 1 : 2) * (long long)d.ncu);
    using E = typename PL<SP>::e;
    const typename PL<SP>::v8 *w0, *w1;
    if constexpr (SP) { w0 = m->conv[3].w; w1 = m->conv[4].w; } else { w0 = m->conv[3].w16; w1 = m->conv[4].w16; }
    k_me_mid<ROWS, TX, SP><<<dim3(grid), dim3(512), SHM, st>>>(ih, il, reinterpret_cast<E*>(oh), reinterpret_cast<E*>(ol), w0, m->conv[3].bias,
                                                               m->conv[3].rbias, w1, m->conv[4].bias, H, W, (int)ntiles);
    return hipGetLastError();
}
constexpr int kPoolNY = 24;                // output rows per thread: (NY + KH - 1) / NY of the minimum input traffic (3 ... 24: same time)
template <int KH, int KW, int SH, int SW, int PH, int PW>
hipError_t launch_pool(hipStream_t st, const bf16x8* ih, const bf16x8* il, bf16x8* oh, bf16x8* ol, int Bc, int H, int W, int C,
                       int Ho, int Wo) {
    constexpr int NY = kPoolNY;
#ifndef DC_ME_POOL_XO
#define DC_ME_POOL_XO 1                    // adjacent output columns per thread (2 measured 4.43 vs 4.37 ms per encode_music: 173 VGPRs, profiles/r05_ab_pool.txt)
#endif
    constexpr int XO = DC_ME_POOL_XO;
    const long long total = (long long)Bc * ((Ho + NY - 1) / NY) * ((Wo + XO - 1) / XO) * (C / 8);
    k_me_pool<KH, KW, SH, SW, PH, PW, XO><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(ih, il, oh, ol, Bc, H, W, C / 8, Ho, Wo, NY);
    return hipGetLastError();
}
template <int KH, int KW, int SH, int SW, int PH, int PW>
hipError_t launch_pool16(hipStream_t st, const f16x8* in, f16x8* out, int Bc, int H, int W, int C, int Ho, int Wo) {
    // (DC_ME_POOL_NY: diagnostic.  Rows per thread 6 / 12 / 24 / 48 / 96: the three pools together 719 / 730 / 682 / 805 / 1 116 us per 32 clips)
    static const int NY = getenv("DC_ME_POOL_NY") ? std::max(1, atoi(getenv("DC_ME_POOL_NY"))) : kPoolNY;
    const long long total = (long long)Bc * ((Ho + NY - 1) / NY) * Wo * (C / 8);
    k_me_pool16<KH, KW, SH, SW, PH, PW><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(in, out, Bc, H, W, C / 8, Ho, Wo, NY);
    return hipGetLastError();
}

}  // namespace

// plane format of this encoder's next dc_music_encode calls: 0 = split bf16 planes (~6e-6), 1 = one fp16 plane (~4e-4, about half the
// time); DC_ME_PREC=f16 | split (read per call) overrides it
void dc_music_set_format(dc_music* m, int single_fp16) {
    if (m) m->format = single_fp16 ? 1 : 0;
}
int dc_music_format(const dc_music* m) {
    if (const char* e = getenv("DC_ME_PREC")) return (!strcmp(e, "f16") || !strcmp(e, "fp16")) ? 1 : 0;
    return m ? m->format : 0;
}

hipError_t dc_music_encode(dc_music* m, const float* d_mel, int B, int Tm, float* d_xf_proj, float* d_xf_out, hipStream_t st,
                           std::string* err) {
    const int T = dc_music_frames(Tm);
    // chunk of clips whose largest activation (16 channels x Tm x 128 bins, two bf16 planes) stays under ~1.4 GB per buffer (32 clips of 60 s in one pass: 4.5 ms; 16 per pass: 4.7 ms, 8: 5.0 ms)
    int chunk = std::max(1, std::min(B, 32 * 5400 / std::max(Tm, 1)));
    if (const char* e = getenv("DC_ME_CHUNK")) chunk = std::max(1, std::min(B, atoi(e)));
    if (chunk > m->cap || Tm > m->cap_tm) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        const int ncap = std::max(chunk, m->cap), ntm = std::max(Tm, m->cap_tm);
        const size_t plane = (size_t)ncap * ntm * 128 * 16 * 2;
        for (bf16x8** p : {&m->a_hi, &m->a_lo, &m->b_hi, &m->b_lo}) {
            if (*p) hipFree(*p);
            *p = nullptr;
            if ((e = hipMalloc((void**)p, plane)) != hipSuccess) {
                *err = "activation plane allocation failed";
                m->cap = m->cap_tm = 0;
                return e;
            }
        }
        m->ws_bytes += 4 * (long long)plane - 4LL * m->cap * m->cap_tm * 128 * 16 * 2;
        m->cap = ncap;
        m->cap_tm = ntm;
    }
    bf16x8 *ah = m->a_hi, *al = m->a_lo, *bh = m->b_hi, *bl = m->b_lo;
#define ME_TRY(x)                         \
    do {                                  \
        hipError_t e_ = (x);              \
        if (e_ != hipSuccess) return e_;  \
    } while (0)
    if (dc_music_format(m)) {              // one fp16 plane per activation (the hi buffers; half of each is used)
        f16x8 *pa = reinterpret_cast<f16x8*>(m->a_hi), *pb = reinterpret_cast<f16x8*>(m->b_hi);
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int Bc = std::min(chunk, B - b0);
            const float* mel = d_mel + (size_t)b0 * Tm * 128;
            ME_TRY(launch_stem<false>(st, m, mel, pa, nullptr, Bc, Tm, 128));
            ME_TRY((launch_pool16<5, 5, 1, 2, 2, 2>(st, pa, pb, Bc, Tm, 128, 16, Tm, 64)));
            ME_TRY(launch_mid<false>(st, m, pb, nullptr, pa, nullptr, Bc, Tm, 64));
            ME_TRY((launch_pool16<5, 5, 3, 2, 2, 2>(st, pa, pb, Bc, Tm, 64, 32, T, 32)));
            ME_TRY((launch_conv_t16<32, 32, 1, 8, 1>(st, m->conv[5], pb, pa, Bc, T, 32)));
            ME_TRY((launch_conv_t16<32, 32, 1, 8, 1>(st, m->conv[6], pa, pb, Bc, T, 32)));
            ME_TRY((launch_pool16<3, 3, 1, 2, 1, 1>(st, pb, pa, Bc, T, 32, 32, T, 16)));
            const int M = Bc * T;
            const unsigned grid = (unsigned)((M + 127) / 128);
            float* xo = d_xf_out + (size_t)b0 * T * 64;
            float* xp = d_xf_proj + (size_t)b0 * T * 64;
            k_me_head<false><<<dim3(grid), dim3(256), 0, st>>>(pa, nullptr, m->w4_16, m->b4, xo, M);
            ME_TRY(hipGetLastError());
            k_me_proj<<<dim3(grid), dim3(256), 0, st>>>(xo, m->wp, m->bp, xp, M);
            ME_TRY(hipGetLastError());
        }
        return hipSuccess;
    }
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        const float* mel = d_mel + (size_t)b0 * Tm * 128;
        if (!getenv("DC_ME_NO_STEM")) {
            ME_TRY(launch_stem<true>(st, m, mel, ah, al, Bc, Tm, 128));
        } else {
            ME_TRY((launch_conv<1, 16, 0>(st, m->conv[0], mel, nullptr, nullptr, ah, al, Bc, Tm, 128)));
            ME_TRY((launch_conv_t<16, 16, 1, 8, 2>(st, m->conv[1], ah, al, bh, bl, Bc, Tm, 128)));
            ME_TRY((launch_conv_t<16, 16, 1, 8, 2>(st, m->conv[2], bh, bl, ah, al, Bc, Tm, 128)));
        }
        ME_TRY((launch_pool<5, 5, 1, 2, 2, 2>(st, ah, al, bh, bl, Bc, Tm, 128, 16, Tm, 64)));
        if (!getenv("DC_ME_NO_MID")) {
            ME_TRY(launch_mid<true>(st, m, bh, bl, ah, al, Bc, Tm, 64));
            std::swap(ah, bh);      // (the two-launch form leaves conv2.1's output in b)
            std::swap(al, bl);
        } else {
            ME_TRY((launch_conv_t<16, 32, 2, 8, 2>(st, m->conv[3], bh, bl, ah, al, Bc, Tm, 64)));
            ME_TRY((launch_conv_t<32, 32, 1, 8, 1>(st, m->conv[4], ah, al, bh, bl, Bc, Tm, 64)));
        }
        ME_TRY((launch_pool<5, 5, 3, 2, 2, 2>(st, bh, bl, ah, al, Bc, Tm, 64, 32, T, 32)));
        ME_TRY((launch_conv_t<32, 32, 1, 8, 1>(st, m->conv[5], ah, al, bh, bl, Bc, T, 32)));
        ME_TRY((launch_conv_t<32, 32, 1, 8, 1>(st, m->conv[6], bh, bl, ah, al, Bc, T, 32)));
        ME_TRY((launch_pool<3, 3, 1, 2, 1, 1>(st, ah, al, bh, bl, Bc, T, 32, 32, T, 16)));
        const int M = Bc * T;
        const unsigned grid = (unsigned)((M + 127) / 128);
        float* xo = d_xf_out + (size_t)b0 * T * 64;
        float* xp = d_xf_proj + (size_t)b0 * T * 64;
        k_me_head<true><<<dim3(grid), dim3(256), 0, st>>>(bh, bl, m->w4, m->b4, xo, M);
        ME_TRY(hipGetLastError());
        k_me_proj<<<dim3(grid), dim3(256), 0, st>>>(xo, m->wp, m->bp, xp, M);
        ME_TRY(hipGetLastError());
    }
#undef ME_TRY
    return hipSuccess;
}
