// dc_kernels.hip - hand-written gfx950 (CDNA4) kernels of the DDIM denoising step.
//
// Design: every activation lives in "FT" form - a stack of 32x32 fp32
// tiles in the v_mfma_f32_32x32x16 accumulator layout with the TOKEN on the lane and the
// FEATURES in registers.  Consequences:
//   * per-token reductions over features (LayerNorm, softmax over head_dim, FiLM) are
//     in-lane register reductions plus ONE v_permlane32_swap with lane^32;
//   * an accumulator tile converts in registers (v_cvt_pk) into the B operand of the next
//     MFMA (W * X) or the A operand (X^T * W), so the whole
//     LN -> QKV -> softmax -> attention -> FiLM -> out-proj -> FFN chain of one token
//     group never leaves the register file; weights are pre-packed host-side in the
//     matching fragment order so each operand is one 16-B-per-lane access;
//   * the only cross-token dependency per layer (softmax over the sequence of K and
//     K^T V of the linear attention) is a two-phase reduction: per-group partial
//     records (k_embed_front / k_layer) -> k_attn_combine.
//
// Operand formats: T16 = __bf16 or _Float16 (same MFMA rate on gfx950), optionally SPLIT
// (x = hi + lo, three MFMAs per product: hi*hi + lo*hi + hi*lo) for ~fp32 accuracy.
#include "dc_dev.h"


// ------------------------------------------------------------------------------------
// per-step bookkeeping: iteration counter -> timestep per clip, DDIM scalars, snapshot slot
// ------------------------------------------------------------------------------------
__global__ void k_begin_step(int* __restrict__ iter, const int* __restrict__ t_of_iter,
                             const float* __restrict__ coef_of_t, const int* __restrict__ snap_of_iter,
                             int* __restrict__ t_clip, float* __restrict__ coef_cur, int* __restrict__ snap_cur, int B) {
    const int it = *iter;
    const int t = t_of_iter[it];
    for (int b = threadIdx.x; b < B; b += blockDim.x) t_clip[b] = t;
    if (threadIdx.x < DC_COEF) coef_cur[threadIdx.x] = coef_of_t[t * DC_COEF + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        snap_cur[0] = snap_of_iter ? snap_of_iter[it] : -1;
        snap_cur[1] = it;                  // the iteration the step kernels are in (indexes the per-iteration noise, eta > 0)
        *iter = it + 1;
    }
}

// end of a captured graph of k steps whose kernels indexed the iteration tables themselves (iter_base): next replay starts k later
__global__ void k_advance_iter(int* __restrict__ iter, int k) { *iter += k; }
__global__ void k_set_ptr(const float** slot, const float* p, unsigned long long seed, unsigned long long first) {
    slot[0] = p;                                                   // [0]: base of the step noise
    reinterpret_cast<unsigned long long*>(slot)[1] = seed;         // [1]: the generator's seed
    reinterpret_cast<unsigned long long*>(slot)[2] = first;        // [2]: index of z[0] in the whole batch's [B][T][P] draw (clip shards)
}

// ------------------------------------------------------------------------------------
// eta > 0 without a caller-supplied noise tensor: the N(0, 1) draws of ONE iteration (the reference's th.randn_like(x),
// gaussian_diffusion.py:822), generated at the head of the step that consumes them - a [B][Tx][P] buffer instead of the
// [S][B][Tx][P] tensor (6 GB at S = 1000, bs = 32).  Philox4x32-10 keyed by the seed, counter = (element quad, iteration);
// Box-Muller on the four words.  Element e of iteration it depends on (seed, it, e) only: batch layout, graph form and
// launch geometry do not enter.  `first` (seed_slot[1], or the argument) is the index of z[0] in the WHOLE batch's draw: a rank that
// samples clips [lo, hi) of a sharded batch passes lo*T*P and gets exactly the rows the unsharded run would have drawn for them.
// ------------------------------------------------------------------------------------
DEV void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned)p1;
    c[3] = (unsigned)p0;
    c[0] = n0;
    c[2] = n2;
}
__global__ __launch_bounds__(256) void k_step_noise(float* __restrict__ z, size_t n, unsigned long long seed,
                                                    const unsigned long long* __restrict__ seed_slot /* the seed, when given (a captured graph
                                                    must not bake it in) */, const int* __restrict__ iter_base, int step,
                                                    const int* __restrict__ snap_cur, unsigned long long first) {
    if (seed_slot) {
        seed = seed_slot[0];
        first = seed_slot[1];
    }
    const unsigned it = (unsigned)(iter_base ? step + *iter_base : (snap_cur ? snap_cur[1] : step));
    const size_t q0 = first >> 2, q1 = (first + n + 3) >> 2;          // the global element quads that cover [first, first + n)
    for (size_t q = q0 + (size_t)blockIdx.x * 256 + threadIdx.x; q < q1; q += (size_t)gridDim.x * 256) {
        unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), it, 0x5eedu};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float o[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);            // (0, 1]
            const float u2 = (float)(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);                 // [0, 1)
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            o[2 * h] = rad * cs;
            o[2 * h + 1] = rad * sn;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t e = 4 * q + i;
            if (e >= first && e - first < n) z[e - first] = o[i];
        }
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
// `self.linear` (64 -> 512, transformer.py:479-480) of one 32-token group, straight into the fragment-major operand images the
// step kernels read - one kernel instead of linear + row statistics + pack with a [M][512] fp32 round trip in between:
//   MODE 0 (xf_proj): the fp32 image of emb's step-invariant term (layout: ld_pp);
//   MODE 1 (xf_out):  text_norm without its affine (transformer.py:149; folded into the K/V projections) as bf16 hi + lo
//                     operand images [g][32 ks][64 lanes][8]: lane (c, hh), element j = nhat[32 g + c][16 ks + 8 hh + j].
// 256 threads: thread t owns output features t and t + 256 for all 32 tokens (64 accumulators), the weight rows it needs
// are coalesced L2 reads, x comes from LDS as broadcasts; the [32][512] result tile is turned through LDS (rows padded to
// 516 floats: a lane's 32-byte reads of 16 different rows then fall on different banks).
template <int MODE>
__global__ __launch_bounds__(256, 2)
void k_cond_embed(const float* __restrict__ xf /*[M][64]*/, const float* __restrict__ wt /*[64][512]*/, const float* __restrict__ b,
                  float* __restrict__ out_f32, bf16x8* __restrict__ out_hi, bf16x8* __restrict__ out_lo, int M, int T, int Tx) {
    // T: clip stride of the operand images (a multiple of 32 when the clips are padded), Tx <= T: frames per clip of xf; rows of the
    // padding (n >= Tx) and past M read as zeros
    constexpr int YS = 516;
    __shared__ __attribute__((aligned(16))) float xs[32 * 64];
    __shared__ __attribute__((aligned(16))) float ys[32 * YS];
    __shared__ float mu_s[32], rs_s[32];
    const int g = blockIdx.x, t = threadIdx.x;
    {   // x tile: 32 tokens x 64 floats = 512 16-byte pieces; tokens past M read as zeros
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = t + 256 * i;
            const int tok = 32 * g + (p >> 4);
            const int bb = tok / T, nn = tok - bb * T;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tok < M && nn < Tx) v = reinterpret_cast<const f32x4*>(xf)[((size_t)bb * Tx + nn) * 16 + (p & 15)];
            reinterpret_cast<f32x4*>(xs)[p] = v;
        }
    }
    __syncthreads();
    float acc0[32], acc1[32];
    const float b0 = b[t], b1 = b[t + 256];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        acc0[k] = b0;
        acc1[k] = b1;
    }
#pragma unroll 2
    for (int i4 = 0; i4 < 16; ++i4) {
        float w0[4], w1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w0[i] = wt[(4 * i4 + i) * 512 + t];
            w1[i] = wt[(4 * i4 + i) * 512 + t + 256];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const f32x4 x4 = reinterpret_cast<const f32x4*>(xs)[k * 16 + i4];       // broadcast
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0[k] = fmaf(x4[i], w0[i], acc0[k]);
                acc1[k] = fmaf(x4[i], w1[i], acc1[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const bool live = 32 * g + k < M && (32 * g + k) % T < Tx;      // rows past M and padding rows: zeros
        ys[k * YS + t] = live ? acc0[k] : 0.f;
        ys[k * YS + t + 256] = live ? acc1[k] : 0.f;
    }
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    if constexpr (MODE == 1) {      // per-token mean / rstd over the 512 features (two passes, as nn.LayerNorm): a wave per 8 tokens
        for (int k = wave * 8; k < wave * 8 + 8; ++k) {
            float v[8], sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = ys[k * YS + lane + 64 * i];
                sum += v[i];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mu = sum * (1.f / 512.f);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) q += (v[i] - mu) * (v[i] - mu);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if (lane == 0) {
                mu_s[k] = mu;
                rs_s[k] = rsqrtf(q * (1.f / 512.f) + 1e-5f);
            }
        }
        __syncthreads();
    }
    // copy-out: fragment (g, ks), lane (c = token, hh): the 8 features 16 ks + 8 hh .. + 7 of row c; a wave writes 1-KiB pieces
    const int c = lane & 31, hh = lane >> 5;
    for (int ks = wave; ks < 32; ks += 4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(ys + c * YS + 16 * ks + 8 * hh);
        const f32x4 a = src[0], d = src[1];
        if constexpr (MODE == 0) {
            f32x4* o = reinterpret_cast<f32x4*>(out_f32) + ((size_t)g * 32 + ks) * 128 + lane;
            o[0] = a;
            o[64] = d;
        } else {
            const float mu = mu_s[c], rs = rs_s[c];
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = ((j < 4 ? a[j] : d[j - 4]) - mu) * rs;
                hi[j] = (__bf16)n;
                lo[j] = (__bf16)(n - (float)hi[j]);
            }
            out_hi[((size_t)g * 32 + ks) * 64 + lane] = hi;
            out_lo[((size_t)g * 32 + ks) * 64 + lane] = lo;
        }
    }
}

// Cross-attention K/V for every layer + their partial records (transformer.py:149-155):
// K = Wk' nhat + bk', V = Wv' nhat + bv' with text_norm's affine folded into Wk'/Wv'.
// grid (ceil(G/8), L), 512 threads: one wave per (group, layer); the 16 weight fragments of a k-step (K hi, K lo, V hi, V lo x
// 4 feature tiles) are copied L2 -> LDS once per workgroup by LDS-DMA (double-buffered, one barrier per k-step) and serve
// all 8 waves - read straight from L2 by every wave they were 8.3 GB of traffic per batch and the kernel took 1.66 ms.
// CA has no mask: every real token is valid.  One-time cost per batch, so always split-bf16 (plain bf16 here alone costs
// ~2e-3 on the matrices).
__global__ __launch_bounds__(512, 1) void k_cond_ca_partials(const DcModel* __restrict__ dm, const bf16x8* __restrict__ nh_hi,
                                                             const bf16x8* __restrict__ nh_lo, float* __restrict__ recs,
                                                             int M, int T, int G, int Tx /* frames per clip (<= the clip stride T) */) {
    __shared__ __attribute__((aligned(16))) char wbuf[2 * 16384];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int g = blockIdx.x * 8 + wave;
    const bool active = g < G;
    if (!active) g = G - 1;
    const int l = blockIdx.y;
    const DcLayer& L = dm->layer[l];
    const GroupCtx cx = make_ctx(g, lane, M, T);
    constexpr int NF = 4 * DC_KS_E;
    // fragment f of k-step ks: f = which * 4 + oc, which = 0 K hi, 1 K lo, 2 V hi, 3 V lo; natural pack index oc * 32 + ks
    auto stage = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = 2 * wave + i, which = f >> 2, oc = f & 3;
            const bf16x8* src = ((which & 2) ? L.ca_wv : L.ca_wk) + (size_t)(((which & 1) ? NF : 0) + oc * DC_KS_E + ks) * 64 + lane;
            lds_dma16(src, wbuf + buf * 16384 + f * 1024);
        }
    };
    f32x16 K[4], V[4];
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
        K[oc] = splat(L.ca_bk[32 * oc + cx.c]);
        V[oc] = splat(L.ca_bv[32 * oc + cx.c]);
    }
    stage(0, 0);
    bf16x8 a = nh_hi[((size_t)g * DC_KS_E) * 64 + lane], al = nh_lo[((size_t)g * DC_KS_E) * 64 + lane];
    for (int ks = 0; ks < DC_KS_E; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                          // k-step ks has landed for every wave; everyone left buffer (ks + 1) & 1
        const bf16x8 a_cur = a, al_cur = al;
        if (ks + 1 < DC_KS_E) {
            stage(ks + 1, (ks + 1) & 1);
            a = nh_hi[((size_t)g * DC_KS_E + ks + 1) * 64 + lane];
            al = nh_lo[((size_t)g * DC_KS_E + ks + 1) * 64 + lane];
        }
        const bf16x8* w = reinterpret_cast<const bf16x8*>(wbuf + (ks & 1) * 16384) + lane;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            const bf16x8 bk = w[(0 + oc) * 64], bkl = w[(4 + oc) * 64], bv = w[(8 + oc) * 64], bvl = w[(12 + oc) * 64];
            K[oc] = mfma(a_cur, bk, K[oc]);
            V[oc] = mfma(a_cur, bv, V[oc]);
            K[oc] = mfma(al_cur, bk, K[oc]);
            V[oc] = mfma(al_cur, bv, V[oc]);
            K[oc] = mfma(a_cur, bkl, K[oc]);
            V[oc] = mfma(a_cur, bvl, V[oc]);
        }
    }
    if (!active) return;
    float* rec = recs + ((size_t)l * G + g) * 2 * DC_REC_FLOATS;
    const int nslot = cx.straddle ? 2 : 1;
    for (int slot = 0; slot < nslot; ++slot) {
        const RowRange valid = valid_rows(cx, slot, M, T, nullptr, Tx);
        float* R = rec + (size_t)slot * DC_REC_FLOATS;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) emit_partial<__bf16, true>(K[oc], V[oc], oc, valid, R, cx);
    }
}

// MODE 0 of k_cond_embed on the matrix pipe (round 5): pp = linear(xf_proj) for one 32-token group per wave as 16 tiles x 4 k-steps of
// split-bf16 products (weights on the rows: the accumulator has the token on the lane), the image's [token][8 features] pieces
// assembled by a swap between the lane halves; 128 KiB of weight fragments per 8-wave workgroup in LDS.  ~16 mantissa bits: the image
// feeds SiLU(temb + pp), which is rounded to 11.
__global__ __launch_bounds__(512, 1) void k_cond_pp64(const float* __restrict__ xf /*[B][Tx][64]*/, const bf16x8* __restrict__ wp, const float* __restrict__ b,
                                                      float* __restrict__ out_f32, int M, int T, int G, int Tx) {
    extern __shared__ __attribute__((aligned(16))) char wbuf[];          // [hi: ot 16][ks 4][lo: ...] x 1 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int g = blockIdx.x * 8 + wave;
    const bool active = g < G;
    if (!active) g = G - 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) lds_dma16(wp + (size_t)(16 * wave + i) * 64 + lane, wbuf + (16 * wave + i) * 1024);
    const int c = lane & 31, hh = lane >> 5;
    const int tok = 32 * g + c;
    const int bb = tok / T, nn = tok - bb * T;
    const bool live = tok < M && nn < Tx;
    bf16x8 xh[4], xl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (live) {
            const f32x4* px = reinterpret_cast<const f32x4*>(xf) + ((size_t)bb * Tx + nn) * 16 + 4 * ks + 2 * hh;
            v0 = px[0];
            v1 = px[1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? v0[j] : v1[j - 4];
            const __bf16 h = (__bf16)v;
            xh[ks][j] = h;
            xl[ks][j] = (__bf16)(v - (float)h);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) return;
    const bf16x8* w = reinterpret_cast<const bf16x8*>(wbuf) + lane;
    f32x4* out = reinterpret_cast<f32x4*>(out_f32) + (size_t)g * 32 * 128 + lane;
#pragma unroll 2
    for (int ot = 0; ot < 16; ++ot) {
        f32x16 acc = splat(0.f);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 ah = w[(ot * 4 + ks) * 64], al = w[(64 + ot * 4 + ks) * 64];
            acc = mfma(ah, xh[ks], acc);
            acc = mfma(al, xh[ks], acc);
            acc = mfma(ah, xl[ks], acc);
        }
        // rows (r & 3) + 8 (r >> 2) + 4 hh of the tile -> this lane's 8 consecutive features 16 ks + 8 hh .. of k-steps 2 ot, 2 ot + 1
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 lo4, hi4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * q + i]), __float_as_uint(acc[8 * q + 4 + i]), false, false);
                lo4[i] = __uint_as_float(r[0]);
                hi4[i] = __uint_as_float(r[1]);
            }
            const int ks = 2 * ot + q;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + 16 * ks + 8 * hh), b1 = *reinterpret_cast<const f32x4*>(b + 16 * ks + 8 * hh + 4);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            out[(size_t)ks * 128] = live ? lo4 + b0 : z;              // rows of the padding and past M: zeros
            out[(size_t)ks * 128 + 64] = live ? hi4 + b1 : z;
        }
    }
}

// The same records from the 64 music features themselves (round 5).  `linear` (64 -> 512) is shared by all layers and the
// LayerNorm behind it is affine in its input up to the per-token 1 / std, so
//   K = W' n-hat + b' = rstd (A x + d) + b',   A = W' Wc [128][64], d = W' bc        (host: dc_api.hip, build_model)
// - 4 k-steps of 16 per layer instead of 32, no [tokens][512] image written and read back: the pre-pass GEMM does an eighth of the
// products.  rstd comes from k_cond_rstd: the variance of linear(x) over its 512 outputs as a quadratic form of x,
//   var = x^T Gc x + 2 gv^T x + c      (Gc = Wc^T Wc / 512 is positive semi-definite: no cancellation beyond the sum's own)
// Same record format and masking as above; grid (ceil(G/8), L), one wave per (group, layer); the layer's 64 weight fragments
// (K hi, K lo, V hi, V lo x 4 tiles x 4 k-steps, 64 KiB) are copied to LDS once per workgroup.
__global__ __launch_bounds__(256) void k_cond_rstd(const float* __restrict__ xf /*[B][Tx][64]*/, const float* __restrict__ gram, float* __restrict__ rstd /*[G * 32]*/,
                                                   int M, int T, int Tx, int ntok) {
    __shared__ __attribute__((aligned(16))) float gs[64 * 64 + 64 + 4];
    for (int i = threadIdx.x; i < 64 * 64 + 64 + 1; i += 256) gs[i] = gram[i];
    __syncthreads();
    const int tok = blockIdx.x * 256 + threadIdx.x;
    if (tok >= ntok) return;
    const int bb = tok / T, nn = tok - bb * T;
    float r = 0.f;                                      // rows of the padding and past M: K = b', masked by the records' row ranges anyway
    if (tok < M && nn < Tx) {
        f32x4 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = reinterpret_cast<const f32x4*>(xf)[((size_t)bb * Tx + nn) * 16 + i];
        float var = gs[64 * 64 + 64];
#pragma unroll      // (fully: x must stay in registers - a partly unrolled loop indexes it dynamically and it moves to scratch: 157 us instead of ~15)
        for (int i = 0; i < 64; ++i) {
            float u = 2.f * gs[64 * 64 + i];            // 2 gv_i + sum_j Gc[i][j] x_j
#pragma unroll
            for (int j4 = 0; j4 < 16; ++j4) {
                const f32x4 g4 = reinterpret_cast<const f32x4*>(gs)[i * 16 + j4];      // broadcast
                u = fmaf(g4[0], x[j4][0], u);
                u = fmaf(g4[1], x[j4][1], u);
                u = fmaf(g4[2], x[j4][2], u);
                u = fmaf(g4[3], x[j4][3], u);
            }
            var = fmaf(u, x[i >> 2][i & 3], var);
        }
        r = rsqrtf(fmaxf(var, 0.f) + 1e-5f);
    }
    rstd[tok] = r;
}
__global__ __launch_bounds__(512, 1) void k_cond_ca_partials64(const DcModel* __restrict__ dm, const float* __restrict__ xf /*[B][Tx][64]*/,
                                                               const float* __restrict__ rstd, float* __restrict__ recs, int M, int T, int G,
                                                               int Tx /* frames per clip (<= the clip stride T) */) {
    extern __shared__ __attribute__((aligned(16))) char wbuf[];          // 64 fragments: [which: K hi, K lo, V hi, V lo][oc 4][ks 4]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int g = blockIdx.x * 8 + wave;
    const bool active = g < G;
    if (!active) g = G - 1;
    const int l = blockIdx.y;
    const DcLayer& L = dm->layer[l];
    const GroupCtx cx = make_ctx(g, lane, M, T);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = 8 * wave + i, which = f >> 4, rest = f & 15;                    // natural pack: [hi: oc 4][ks 4][lo: ...]
        const bf16x8* src = ((which & 2) ? L.ca_av : L.ca_ak) + (size_t)(((which & 1) ? 16 : 0) + rest) * 64 + lane;
        lds_dma16(src, wbuf + f * 1024);
    }
    // this lane's token (c) as the A operand of the four k-steps: features 16 ks + 8 hh .. + 7, split in registers
    const int tok = 32 * g + cx.c;
    const int bb = tok / T, nn = tok - bb * T;
    const bool live = tok < M && nn < Tx;
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (live) {
            const f32x4* px = reinterpret_cast<const f32x4*>(xf) + ((size_t)bb * Tx + nn) * 16 + 4 * ks + 2 * cx.hh;
            v0 = px[0];
            v1 = px[1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? v0[j] : v1[j - 4];
            const __bf16 h = (__bf16)v;
            ah[ks][j] = h;
            al[ks][j] = (__bf16)(v - (float)h);
        }
    }
    // 1 / std of this lane's 16 token rows (tile_row(r, hh) = 4 consecutive rows per register quad)
    f32x16 rs;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rstd + (size_t)32 * g + 8 * q + 4 * cx.hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[4 * q + i] = v[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) return;
    const bf16x8* w = reinterpret_cast<const bf16x8*>(wbuf) + lane;
    f32x16 K[4], V[4];
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
        K[oc] = splat(0.f);
        V[oc] = splat(0.f);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 bk = w[(0 * 16 + oc * 4 + ks) * 64], bkl = w[(1 * 16 + oc * 4 + ks) * 64], bv = w[(2 * 16 + oc * 4 + ks) * 64],
                         bvl = w[(3 * 16 + oc * 4 + ks) * 64];
            K[oc] = mfma(ah[ks], bk, K[oc]);
            V[oc] = mfma(ah[ks], bv, V[oc]);
            K[oc] = mfma(al[ks], bk, K[oc]);
            V[oc] = mfma(al[ks], bv, V[oc]);
            K[oc] = mfma(ah[ks], bkl, K[oc]);
            V[oc] = mfma(ah[ks], bvl, V[oc]);
        }
        const float dk = L.ca_dk[32 * oc + cx.c], dv = L.ca_dv[32 * oc + cx.c], bk = L.ca_bk[32 * oc + cx.c], bv = L.ca_bv[32 * oc + cx.c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            K[oc][r] = fmaf(rs[r], K[oc][r] + dk, bk);
            V[oc][r] = fmaf(rs[r], V[oc][r] + dv, bv);
        }
    }
    float* rec = recs + ((size_t)l * G + g) * 2 * DC_REC_FLOATS;
    const int nslot = cx.straddle ? 2 : 1;
    for (int slot = 0; slot < nslot; ++slot) {
        const RowRange valid = valid_rows(cx, slot, M, T, nullptr, Tx);
        float* R = rec + (size_t)slot * DC_REC_FLOATS;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) emit_partial<__bf16, true>(K[oc], V[oc], oc, valid, R, cx);
    }
}

// ------------------------------------------------------------------------------------
// combine partial records of one (set, clip, 32-feature tile) into attention operand frags
//   A[d][l] = sum_g w_g[d] P_g[d][l] / sum_g w_g[d] ssum_g[d],   w_g = exp(m_g - max_g m_g)
// recs: [nset][NU units][2][DC_REC_FLOATS] (a unit = `gran` consecutive tokens); afrag out: [nset][B][16 frags (8 hi, 8 lo)][64 lanes][8] T16.
// grid (B, 4, nset), 1024 threads: thread (q8, lane, hq) sums group-eighth q8 of kept values 4hq..4hq+3 of its lane.
// All sums run in a fixed order: re-running is bit-identical.
// ------------------------------------------------------------------------------------
template <class T16>
__global__ __launch_bounds__(1024) void k_attn_combine(const float* __restrict__ recs, v8<T16>* __restrict__ afrag,
                                                       int T, int NU, int B, int gran) {
    extern __shared__ float sm[];   // w[ng][32], z[32], red[32][32], pacc[4][256][4]
    const int b = blockIdx.x, oc = blockIdx.y, set = blockIdx.z;
    const int g_lo = (b * T) / gran, g_hi = ((b + 1) * T - 1) / gran;     // record units (gran tokens each) of this clip
    const int ng = g_hi - g_lo + 1;
    const float* base = recs + (size_t)set * NU * 2 * DC_REC_FLOATS;
    float* w = sm;
    float* z = sm + ng * 32;
    float* red = z + 32;
    float* pacc = red + 1024;
    const int tid = threadIdx.x;
    auto rec_of = [&](int gi) -> const float* {
        const int g = g_lo + gi;
        const int slot = ((gran * g) / T == b) ? 0 : 1;
        return base + ((size_t)g * 2 + slot) * DC_REC_FLOATS;
    };
    // every load of the block is issued before the first dependent use: one memory round trip in all.
    const int f = tid & 31, part = tid >> 5;          // phases 1-2: 32 features x 32 group-parts
    const int q8 = tid >> 7, t7 = tid & 127;          // phase 3: 8 group-eighths x (lane, half)
    const int lane = t7 & 63, hq = t7 >> 6;           // this thread: kept values 4hq..4hq+3 of the lane
    const int c = lane & 31, hh = lane >> 5;
    // kept value j of lane (c,hh) = tile register r = 8*(c>>4) + j -> E-feature row tile_row(r, hh)
    const int r0 = 8 * (c >> 4) + 4 * hq;
    const int row0 = tile_row(r0, hh);                // rows row0 .. row0+3 (r0 is a multiple of 4)
    const int per = (ng + 7) / 8, gb = q8 * per, ge = min(gb + per, ng);
    constexpr int PMAX = 8;                           // records preloaded per thread (covers T <= 1984; longer clips loop)
    f32x4 p[PMAX];
#pragma unroll
    for (int u = 0; u < PMAX; ++u)
        if (gb + u < ge) p[u] = reinterpret_cast<const f32x4*>(rec_of(gb + u) + 256)[(oc * 64 + lane) * 2 + hq];
    float mreg[4], sreg[4];                            // this thread's groups: part, part+32, ...  (ng <= 128)
    float mloc = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gi = part + 32 * k;
        mreg[k] = 0.f;
        sreg[k] = 0.f;
        if (gi < ng) {
            const float* R = rec_of(gi);
            sreg[k] = R[128 + 32 * oc + f];
            mreg[k] = R[32 * oc + f];
            if (sreg[k] > 0.f) mloc = fmaxf(mloc, mreg[k]);
        }
    }
    red[part * 32 + f] = mloc;
    __syncthreads();
    float mstar = red[f];
#pragma unroll 8
    for (int k = 1; k < 32; ++k) mstar = fmaxf(mstar, red[k * 32 + f]);
    __syncthreads();
    // phase 2: weights w_g = exp2(m_g - m*) and the normaliser (fixed summation order)
    float zloc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gi = part + 32 * k;
        if (gi < ng) {
            const float ww = sreg[k] > 0.f ? exp2f_fast(mreg[k] - mstar) : 0.f;
            w[gi * 32 + f] = ww;
            zloc += ww * sreg[k];
        }
    }
    red[part * 32 + f] = zloc;
    __syncthreads();
    if (tid < 32) {
        float zz = 0.f;
        for (int k = 0; k < 32; ++k) zz += red[k * 32 + tid];
        z[tid] = zz;
    }
    // phase 3: weighted sum of the partial K^T V blocks; 8 group-eighths in parallel, each in group order
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < PMAX; ++u)
        if (gb + u < ge) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[(gb + u) * 32 + row0 + i], p[u][i], acc[i]);
        }
    for (int gi = gb + PMAX; gi < ge; ++gi) {          // clips longer than PMAX*8 groups
        const f32x4 pp = reinterpret_cast<const f32x4*>(rec_of(gi) + 256)[(oc * 64 + lane) * 2 + hq];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[gi * 32 + row0 + i], pp[i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) pacc[(q8 * 128 + t7) * 4 + i] = acc[i];
    __syncthreads();
    if (q8 != 0) return;
    // thread (lane, hq) writes registers r0..r0+3 of the operand tile; the other 8 registers of the lane are
    // cross-head entries = 0 (written by the same thread: its partner half)
    v4<T16> hi, lo, zero;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float tot = pacc[t7 * 4 + i];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += pacc[(k * 128 + t7) * 4 + i];
        const float zz = z[row0 + i];
        const float a = zz > 0.f ? tot / zz : 0.f;
        hi[i] = (T16)a;
        lo[i] = (T16)(a - (float)hi[i]);
        zero[i] = (T16)0.f;
    }
    // register r of the tile  ->  k-step s = r>>3, element j = r&7 of the A-operand frag
    v8<T16>* out = afrag + ((size_t)set * B + b) * 16 * 64;
    const int s = r0 >> 3, j0 = r0 & 7;               // kept block: s == c>>4
    reinterpret_cast<v4<T16>*>(out + ((oc * 2 + s) * 64 + lane))[j0 >> 2] = hi;
    reinterpret_cast<v4<T16>*>(out + ((8 + oc * 2 + s) * 64 + lane))[j0 >> 2] = lo;
    reinterpret_cast<v4<T16>*>(out + ((oc * 2 + (s ^ 1)) * 64 + lane))[j0 >> 2] = zero;
    reinterpret_cast<v4<T16>*>(out + ((8 + oc * 2 + (s ^ 1)) * 64 + lane))[j0 >> 2] = zero;
}

// ------------------------------------------------------------------------------------
// per step: S = SiLU(time_embed[t] + xf_proj')  (transformer.py:482 + StylizationBlock's nn.SiLU, :57-58)
// as the 16-bit B-operand image of the FiLM GEMM, [G][32 ks][64 lanes][8]
// ------------------------------------------------------------------------------------
template <class T16, bool SPLIT>
__global__ void k_silu_emb(const float* __restrict__ pp /*frag-major fp32*/, const float* __restrict__ temb,
                           const int* __restrict__ t_clip, v8<T16>* __restrict__ s_hi, v8<T16>* __restrict__ s_lo,
                           int G, int T, int B, const int* __restrict__ iter_base) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)G * 32 * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) & 31;
    const int g = (int)(idx >> 11);
    const int tok = g * 32 + (lane & 31);
    const int b = min(tok / T, B - 1);
    // graph-captured loop: one timestep for all clips, t_clip = this step's slot of the iteration table, *iter_base = the
    // iteration at which the replay began
    const float* te = temb + (size_t)t_clip[iter_base ? *iter_base : b] * 512 + 16 * ks + 8 * (lane >> 5);
    const f32x8 p = ld_pp(pp, idx >> 6, lane);
    const f32x8 tv = *reinterpret_cast<const f32x8*>(te);
    v8<T16> hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = silu(p[j] + tv[j]);
        hi[j] = (T16)v;
        if constexpr (SPLIT) lo[j] = (T16)(v - (float)hi[j]);
    }
    s_hi[idx] = hi;
    if constexpr (SPLIT) s_lo[idx] = lo;
}

// ------------------------------------------------------------------------------------
// FiLM GEMM: all 3*L StylizationBlocks at once (StylizationBlock.emb_layers + norm, transformer.py:57-60,74-78).
// The host folds the block's LayerNorm affine and the emb_layers bias into the operands (dc_api.hip build_model), so the
// GEMM yields the modulation tiles directly: G'-1 = g*(1+scale)-1 and H' = beta*(1+scale)+shift, stored as fp16 FT
// tiles (the "-1" keeps the fp16 rounding on the small modulation, not on the ~1 multiplier),
// E[g][blk][G'0..3, H'0..3][64][16].  The weight image interleaves each block's tiles as
// (G'0, H'0, G'1, H'1, ...) so that a wave holds matching tiles; bias_ft holds the accumulators' initial values.
// v1: operands straight from L2 into registers; wave tile 4 feature tiles x 2 groups.
// grid (NT/8, ceil(G/4)), 256 threads = 2x2 waves.
// ------------------------------------------------------------------------------------
template <class T16, bool SPLIT>
__global__ __launch_bounds__(256) void k_film_gemm(const v8<T16>* __restrict__ W, const float* __restrict__ bias_ft,
                                                   const v8<T16>* __restrict__ S_hi, const v8<T16>* __restrict__ S_lo,
                                                   f16x16* __restrict__ E, int G, int NT, int* __restrict__ status) {
    using OP = v8<T16>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ot0 = (blockIdx.x * 2 + (wave >> 1)) * 4;   // interleaved order: (s_j, h_j, s_j+1, h_j+1)
    const int g0 = (blockIdx.y * 2 + (wave & 1)) * 2;
    if (g0 >= G) return;
    const bool two = g0 + 1 < G;
    const int g1 = two ? g0 + 1 : g0;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = ld_ft(bias_ft, ot0 + i, lane >> 5);
    const size_t nfw = (size_t)NT * DC_KS_E;
#pragma unroll 2
    for (int ks = 0; ks < DC_KS_E; ++ks) {
        OP a[4], al[4], b[2], bl[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t fi = (size_t)(ot0 + i) * DC_KS_E + ks;                      // natural pack: [ot][ks]
            a[i] = W[fi * 64 + lane];
            if constexpr (SPLIT) al[i] = W[(nfw + fi) * 64 + lane];
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
    const int blk = ot0 >> 3, pair0 = (ot0 & 7) >> 1;       // this wave holds feature tiles pair0, pair0+1 of block blk
    {   // fp16 storage range check (DC_STATUS_F16_SAT): a modulation beyond +-65504 would be stored as inf
        float am = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(acc[i][j][r]));
        if (status && am > 65504.f) atomicOr(status, DC_STATUS_F16_SAT);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two) continue;
            f16x16 og, oh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                og[r] = (_Float16)acc[2 * p][j][r];
                oh[r] = (_Float16)acc[2 * p + 1][j][r];
            }
            store_etile(E, (size_t)(g0 + j) * NT + blk * 8 + pair0 + p, lane, og);
            store_etile(E, (size_t)(g0 + j) * NT + blk * 8 + 4 + pair0 + p, lane, oh);
        }
}


// Work shares of the persistent FiLM GEMM: every wave derives the same integer boundaries (see k_film_gemm3).
DEV void film_shares(int& u0, int& u1, long long nunit, const float* __restrict__ rate_in, int lane, int nw /* GEMM workgroups: the first nw of the grid */) {
    const int b = blockIdx.x;
    float xs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // lane i covers workgroups i, i + 64, ...: all on XCD i & 7
    int nzero = 0;
    for (int i = lane; i < nw; i += 64) {
        const float r = rate_in ? rate_in[i] : 0.f;
        xs[0] += r;
        nzero += !(r > 0.f);
    }
    // sum over the lanes of the same XCD (lane & 7 fixed): xor-shuffles with 8, 16, 32
    float xsum = xs[0];
#pragma unroll
    for (int m = 32; m >= 8; m >>= 1) xsum += __shfl_xor(xsum, m);
    int cnt = 0;
    for (int i = lane; i < nw; i += 64) ++cnt;
#pragma unroll
    for (int m = 32; m >= 8; m >>= 1) cnt += __shfl_xor(cnt, m);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nzero += __shfl_xor(nzero, m);
    float tot = xsum;                                           // sum over all workgroups
#pragma unroll
    for (int m = 4; m >= 1; m >>= 1) tot += __shfl_xor(tot, m);
    // (few units per workgroup - small batches - : equal shares; a +-20 % weight there only moves WHOLE units, e.g. at one clip,
    // 180 units on 180 workgroups, it left some workgroups with two units and others with none: 25 vs 18 us per launch)
    const bool adaptive = rate_in && nzero == 0 && nw >= 64 && nunit >= 8ll * nw;
    // this lane's XCD weight (relative speed, clamped), as an integer
    int wx = 4096;
    if (adaptive) wx = (int)(fminf(fmaxf((xsum / (float)cnt) * ((float)nw / tot), 0.8f), 1.2f) * 4096.f + 0.5f);
    int before = 0, mine = 0, total = 0;
    for (int i = lane; i < nw; i += 64) {                       // workgroup i is on XCD i & 7 == lane & 7
        total += wx;
        before += i < b ? wx : 0;
        mine += i == b ? wx : 0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        total += __shfl_xor(total, m);
        before += __shfl_xor(before, m);
        mine += __shfl_xor(mine, m);
    }
    u0 = __builtin_amdgcn_readfirstlane((int)(nunit * before / total));
    u1 = __builtin_amdgcn_readfirstlane((int)(nunit * (before + mine) / total));
}

// ------------------------------------------------------------------------------------
// FiLM GEMM (non-split formats), S-stationary and persistent, on v_mfma_f32_16x16x32.  A workgroup = 8 waves works on 4 token
// groups (128 tokens): their operand slab S = SiLU(temb[t] + linear(xf_proj)) (128 KiB as f16) is built once in LDS and stays
// there while the waves sweep the feature-tile PAIRS (scale tile, shift tile); weight fragments stream L2 -> registers through
// a PF-deep software prefetch ring (each fragment is used by exactly one wave, so LDS staging would buy nothing); there is
// no barrier in the sweep.  Every workgroup owns a contiguous share of the (token block x round of 8 pairs) units, sized by
// the per-XCD speeds measured in earlier launches (film_shares), and refills its slab when it crosses into the next block.
// (The 32x32x16 form of the same schedule, k_film_gemm2 of round 1, measured 6.4 % slower: see DESIGN.md.)
// Why: this kernel is power-limited (it holds 1.6-1.8 GHz), and on this part a loop of 16x16x32 MFMAs sustains a
// higher clock than the same FLOPs issued as 32x32x16 (MI355X_MICROARCH.md, DVFS item 7).  Same bytes, same cycles:
// per 32-deep k-step a wave feeds 32 MFMAs (4 weight fragments x 8 slab fragments) from 4 weight loads + 8 LDS reads.
//   weight image  [tile][ks32][fb][64][8]: lane l = A[16 fb + pi(l & 15)][32 ks32 + 8 (l >> 4) + j]   (pi: see below)
//   slab          [(g, tb16, ks32)][64][8]: lane l = S[token 32 g + 16 tb16 + (l & 15)][32 ks32 + 8 (l >> 4) + j]
//   accumulators  32 blocks of 16 x 16: [tile 2][fb 2][g 4][tb16 2] x f32x4, lane l = rows 4 (l >> 4) + i, token l & 15
// The E tiles keep their layout (32 x 32 accumulator order, dc_common.h): a v_permlane16_swap between the two token
// halves of a block row puts tokens 0..31 on lanes 0..31 / 32..63, and the host orders the weight rows of each
// 16-row block as pi = (0..3, 8..11, 4..7, 12..15), which is where the 32 x 32 layout expects them.
// ------------------------------------------------------------------------------------
template <class T16>
DEV void film_gemm3_body(const v8<T16>* __restrict__ W, const float* __restrict__ bias16,
                         f16x16* __restrict__ E, int G, int NT, int round0, int nround,
                         const float* __restrict__ pp, const float* __restrict__ temb,
                         const int* __restrict__ t_clip, int T, int B,
                         unsigned long long* __restrict__ clk, const float* __restrict__ rate_in,
                         float* __restrict__ rate_out, const int* __restrict__ iter_base, unsigned long long t_begin,
                         int* __restrict__ status, int nw_film) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using OP = v8<T16>;
#ifndef DC_FILM3_PF
#define DC_FILM3_PF 2
#endif
    constexpr int PF = DC_FILM3_PF;        // weight ring depth in 32-deep k-steps
    constexpr int KS = DC_E / 32;          // 16 k-steps
    if (clk && blockIdx.x == 5 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime();
        clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    if (clk && threadIdx.x == 0 && blockIdx.x < 256) {
        clk[1036 + blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        clk[1036 + blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const OP* slab = reinterpret_cast<const OP*>(lds);
    const int nblk = (G + 3) / 4;
    const long long nunit = (long long)nblk * nround;
    int u0, u1;
    film_shares(u0, u1, nunit, rate_in, lane, nw_film);
    OP a[PF][4];                           // ring slot q: (tile 0 fb 0, tile 0 fb 1, tile 1 fb 0, tile 1 fb 1)
    auto wpair = [&](int p) { return W + (size_t)(2 * p) * 2 * KS * 64 + lane; };        // tile pair p = round * 8 + slot
    auto wfrag = [&](const OP* w, int ks, int i) { return w[((size_t)(i >> 1) * 2 * KS + ks * 2 + (i & 1)) * 64]; };
    if (u0 < u1) {
        const OP* w0 = wpair((round0 + u0 % nround) * 8 + wave);
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[q][i] = wfrag(w0, q, i);
    }
    // The eight waves share the tile pairs of a token block through a counter in LDS instead of owning one pair per round:
    // under the SIMD's oldest-first arbitration one wave of each pair runs ~10 % ahead, and with fixed ownership it then
    // waited ~20 us at every slab boundary while its partner finished alone.  A wave starts with pair `wave` of the segment
    // (so its first weight fragments can be prefetched across the boundary) and claims the following ones when it starts a
    // pair; which wave computes a pair does not change the result.
    int* pair_cnt = reinterpret_cast<int*>(lds + 4 * DC_KS_E * 1024);
    for (int useg = u0; useg < u1;) {
        const int tb = useg / nround, ra = useg % nround;
        const int nr = min(nround - ra, u1 - useg);               // rounds of this token block inside the workgroup's range
        const int npairs = nr * 8;
        const bool next_seg = useg + nr < u1;                     // (its first round is round 0 of the next block)
        const int g0 = tb * 4;
        {
            // slab fill: S = SiLU(temb[t] + linear(xf_proj)) (transformer.py:73-74,482) from the fp32 fragment image, which is in
            // the 32x32x16 operand order [g][ks16][2][64][4]: this lane's 8 values of (token, 32 ks32 + 8 (l >> 4) ..) are
            // the two 16-byte pieces of lane' = 32 ((l >> 4) & 1) + 16 tb16 + (l & 15) in fragment ks16 = 2 ks32 + (l >> 5)
            v8<T16>* slab_w = reinterpret_cast<v8<T16>*>(lds);
            const float* trow[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gg = min(g0 + (i >> 1), G - 1);
                const int b = min((gg * 32 + 16 * (i & 1) + (lane & 15)) / T, B - 1);
                trow[i] = temb + (size_t)t_clip[iter_base ? *iter_base : b] * 512 + 8 * (lane >> 4);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x8 pv[8], tv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = wave + 8 * (8 * half + i);                 // fragment (g, tb16, ks32) = (f >> 5, (f >> 4) & 1, f & 15)
                    const int gi = 2 * half + (i >> 2), t16 = (f >> 4) & 1, ks = f & 15;
                    const int gg = min(g0 + gi, G - 1);
                    pv[i] = ld_pp(pp, (size_t)gg * DC_KS_E + 2 * ks + (lane >> 5), 32 * ((lane >> 4) & 1) + 16 * t16 + (lane & 15));
                    tv[i] = *reinterpret_cast<const f32x8*>(trow[2 * gi + t16] + 32 * ks);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (half == 0) __syncthreads();                              // everyone is done with the previous slab
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = wave + 8 * (8 * half + i);
                    v8<T16> hi;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 z = silu_pair(pv[i][2 * j] + tv[i][2 * j], pv[i][2 * j + 1] + tv[i][2 * j + 1]);
                        hi[2 * j] = (T16)z.x;
                        hi[2 * j + 1] = (T16)z.y;
                    }
                    slab_w[f * 64 + lane] = hi;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (threadIdx.x == 0) *pair_cnt = 8;                  // pairs 0..7 of the segment are taken (pair w by wave w)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
      for (int my = wave; my < npairs;) {
        int nxt = 0;
        if (lane == 0) nxt = __hip_atomic_fetch_add(pair_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nxt = __builtin_amdgcn_readfirstlane(nxt);
        const int p = (round0 + ra + (my >> 3)) * 8 + (my & 7);
        const int ps = nxt < npairs ? (round0 + ra + (nxt >> 3)) * 8 + (nxt & 7) : (next_seg ? round0 * 8 + wave : p);
        const OP* w0 = wpair(p);
        const OP* wn = wpair(ps);                                 // successor: its first k-steps are prefetched in this pair's tail
        f32x4 acc[2][2][4][2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(bias16 + (((size_t)(2 * p + ti) * 2 + fb) * 4 + (lane >> 4)) * 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[ti][fb][g][0] = acc[ti][fb][g][1] = c;
            }
        // slab fragment (g, t16, ks) at ((g * 2 + t16) * KS + ks)
        auto sfrag = [&](int gp, int i, int ks) { return slab[((size_t)((2 * gp + (i >> 1)) * 2 + (i & 1)) * KS + ks) * 64 + lane]; };
        // all eight slab fragments of a k-step are double-buffered (64 registers) and the weight fragment is outermost in the
        // MFMA nest: eight consecutive MFMAs share their A operand.  (Measured: slab fragment outermost +0.8 %; the fragments of
        // one group pair at a time, half a k-step ahead in 32 registers, +1.6 %.)
        OP bb[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bb[0][i] = sfrag(i >> 2, i & 3, 0);
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS; ks0 += PF) {
            const bool tail = ks0 + PF >= KS;
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int ks = ks0 + q, ksn = (ks + 1) & (KS - 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) bb[(q + 1) & 1][i] = sfrag(i >> 2, i & 3, ksn);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        acc[k >> 1][k & 1][i >> 1][i & 1] = mfma16(a[q][k], bb[q & 1][i], acc[k >> 1][k & 1][i >> 1][i & 1]);
                __builtin_amdgcn_sched_barrier(0);
                {
                    const OP* src = tail ? wn : w0;
                    const int ksl = tail ? ks + PF - KS : ks + PF;
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[q][i] = wfrag(src, ksl, i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: fp16, token halves swapped into place, store in the 32 x 32 tile order
        const int blk = p >> 2, t = p & 3;
#ifdef DC_FILM_SAT_CHECK
        // (off in the production build: measured at 12 us per launch = 4 % of this kernel, 1.6 % of the loop, same box; a saturated
        // tile reaches x0 as inf / nan, DC_STATUS_NONFINITE, and dc_sampler_status then scans the tiles: k_scan_f16_nonfinite)
        {   // fp16 storage range check (DC_STATUS_F16_SAT): |value| > 65504 would be stored as inf.  64 v_max3_f32 per 512 MFMAs.
            float am = 0.f;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 x = acc[ti][fb][g][0], y = acc[ti][fb][g][1];
                        am = absmax3(absmax3(absmax3(am, x[0], x[1]), x[2], x[3]), y[0], y[1]);
                        am = absmax3(am, y[2], y[3]);
                    }
            if (status && am > 65504.f) atomicOr(status, DC_STATUS_F16_SAT);
        }
#endif
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
                        const f32x4 x = acc[ti][fb][g][0], y = acc[ti][fb][g][1];
                        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                        const h2v xp = {(_Float16)x[2 * h2], (_Float16)x[2 * h2 + 1]}, yp = {(_Float16)y[2 * h2], (_Float16)y[2 * h2 + 1]};
                        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, xp), __builtin_bit_cast(uint32_t, yp), false, false);
                        o[4 * fb + h2] = r[0];            // registers 8 fb + 2 h2, +1     (block rows 0..3 | 8..11 of this lane half)
                        o[4 * fb + 2 + h2] = r[1];        // registers 8 fb + 4 + 2 h2, +1 (block rows 4..7 | 12..15)
                    }
                store_etile(E, (size_t)(g0 + g) * NT + blk * 8 + 4 * ti + t, lane, __builtin_bit_cast(f16x16, o));
            }
        }
        my = nxt;
      }
        useg += nr;
    }
    if (clk && blockIdx.x == 5 && threadIdx.x == 0) {
        clk[2] = __builtin_amdgcn_s_memtime();
        clk[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (clk && threadIdx.x == 0 && blockIdx.x < 256) {
        clk[1036 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
        clk[1036 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
        clk[1036 + 1024 + blockIdx.x] = (unsigned long long)(u1 - u0);
    }
    if (rate_out && threadIdx.x == 0) {
        const float ticks = (float)(long long)(__builtin_amdgcn_s_memrealtime() - t_begin);
        const float old = rate_in ? rate_in[blockIdx.x] : 0.f;
        const float now = (u1 > u0 && ticks > 0.f) ? (float)(u1 - u0) / ticks : 0.f;
        rate_out[blockIdx.x] = now > 0.f ? (old > 0.f ? 0.5f * old + 0.5f * now : now) : old;
    }
}
template <class T16>
__global__ __launch_bounds__(512, 2) void k_film_gemm3(const v8<T16>* __restrict__ W, const float* __restrict__ bias16,
                                                       f16x16* __restrict__ E, int G, int NT, int round0, int nround,
                                                       const float* __restrict__ pp, const float* __restrict__ temb,
                                                       const int* __restrict__ t_clip, int T, int B,
                                                       unsigned long long* __restrict__ clk, const float* __restrict__ rate_in,
                                                       float* __restrict__ rate_out, const int* __restrict__ iter_base,
                                                       int* __restrict__ status) {
    film_gemm3_body<T16>(W, bias16, E, G, NT, round0, nround, pp, temb, t_clip, T, B, clk, rate_in, rate_out, iter_base,
                         __builtin_amdgcn_s_memrealtime(), status, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------
// step prologue: h = joint_embed(x) + sequence_embedding[:T] (transformer.py:488-490),
// then layer 0's self-attention front half.  One wave per group.
// The K=26 input projection always runs split: rounding x_t itself to 8/11 mantissa bits is the
// single largest error source otherwise, and the GEMM is tiny.
// ------------------------------------------------------------------------------------
// FROMH (test hook, per-group records only): the residual stream is taken from hbuf as it stands instead of being embedded
// from x, and the front half is that of layer l0 - lets a test start a single decoder layer from a given h.
// NARROW (WGR, non-split): 4-wave workgroups = 128-token units, one wave per SIMD - for batches small enough that every
// unit gets a CU of its own (see k_layer).
template <class T16, bool SPLIT, bool WGR, bool FROMH = false, bool NARROW = false>
DEV void embed_front_body(const DcModel* __restrict__ dm, const float* __restrict__ x /*[M][P]*/, float* __restrict__ hbuf,
                          float* __restrict__ recs, const int* __restrict__ length, int M, int T, int G, int B,
                          unsigned long long* __restrict__ clk /* diagnostic: 100-MHz stamps of one wave, or nullptr */, int l0,
                          int Tx /* frames per clip of x (<= the clip stride T) */,
                          int upc /* workgroups per clip (clip-aligned units, WgMap) or 0 */,
                          int wg_fixed /* >= 0: this workgroup's unit (the caller's grid is not the unit grid) */) {
    static_assert(!(FROMH && WGR), "the h-injection hook exists for the per-group-record form only");
    static_assert(!NARROW || (WGR && !SPLIT), "narrow workgroups exist for the workgroup-record form of the non-split formats");
    constexpr int NW = (NARROW || (SPLIT && !WGR)) ? 4 : 8;
    constexpr int WM = SPLIT ? 2 : 1;
    const bool stamping = clk && blockIdx.x == 100 && threadIdx.x == 0;
    auto stamp = [&](int i) {
        if (stamping) clk[i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    using W = v8<T16>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wg = wg_fixed >= 0 ? wg_fixed : (WGR ? wg_index() : (int)blockIdx.x);
    const WgMap wm = wg_map(wg, wave, NW, G, M, T, B, WGR ? upc : 0);
    const int g = wm.g;
    const bool active = wm.active;
    const GroupCtx cx = make_ctx(g, lane, M, T);
    const int P = dm->input_feats;
    const int xb = min(cx.tok, M - 1) / T, xn = cx.tok - xb * T;              // clip / frame of this lane's token
    const bool live = cx.tok < M && xn < Tx;                                     // (frames Tx .. T-1 are padding)
    const int n = live ? xn : 0;
    const size_t xrow = (size_t)xb * Tx + n;                                     // row of x
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // LDS (workgroup records): mx 8 KiB | pst (64 KiB; the key image lies here) | xp 8 KiB (non-split) | ss 4.5 KiB | value image.
    // Split formats: the images are 65 KiB (hi + lo fragments + constants), pst overlays the key image once it is consumed, and no
    // wave spans two clips (clip-aligned units), so there is no xp.
    constexpr int NIMG = 32 * WM + 1;
    constexpr int PST_SZ = SPLIT ? NIMG * 1024 : 65536, XP_SZ = SPLIT ? 0 : 8192;
    constexpr int OFF_SS = 8192 + PST_SZ + XP_SZ;
    constexpr int OFF_IMG_K = 8192, OFF_IMG_V = OFF_SS + 9 * 4 * 32 * 4;
    if constexpr (WGR) {
        // layer 0's key / value images (32 fragments (x 2: hi, lo) + 1 KiB of bias each) go to LDS by LDS-DMA while the embedding is
        // computed; the key image lies in the pst region, which is first written after the barrier that ends its use
        const W* gk = reinterpret_cast<const W*>(dm->layer[0].img_sa_k);
        const W* gv = reinterpret_cast<const W*>(dm->layer[0].img_sa_v);
        for (int f = wave; f < NIMG; f += NW) {
            lds_dma16(gk + f * 64 + lane, lds + OFF_IMG_K + f * 1024);
            lds_dma16(gv + f * 64 + lane, lds + OFF_IMG_V + f * 1024);
        }
    }
    // sequence_embedding rows: issued first, consumed after the embedding GEMM
    const float* se = dm->seq_emb + (size_t)n * DC_D;
    f32x4 sev[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) sev[4 * t + q] = *reinterpret_cast<const f32x4*>(se + 32 * t + 8 * q + 4 * cx.hh);
    // x as chained-order operand: element j of k-step s <-> pose feature 16s + 8(j>>2) + 4hh + (j&3)
    f32x16 h[4];
    if constexpr (FROMH) {
        load_h(h, hbuf, g, lane);
    } else {
        XFrag<T16, true> xf[1];
        f32x16 xv;
        bool staged = false;
        if constexpr (WGR && !SPLIT) {
            // A group's 32 x P floats are contiguous: fetch them with 16-byte loads (<= 4 per lane) and turn them through a
            // wave-private LDS patch.  (Per-element loads touch 32+ cache lines per instruction; the 8 waves' 128 such
            // instructions queue in the CU's address unit for ~4 us.)  The last, partial group keeps the element loads.
            constexpr int OFF_XS = 8192 + 34 * 1024;               // inside pst, behind the key image; 3.5 KiB per wave
            const int gb = (32 * g) / T, gn = 32 * g - gb * T;      // the group's first token
            const size_t row0 = (size_t)gb * Tx + gn;
            // (the patch holds 32 x P floats for P <= 28; x itself must be 16-byte aligned for the f32x4 loads: a caller may pass a slice)
            staged = active && 32 * g + 32 <= M && gn + 32 <= Tx && (row0 * P) % 4 == 0 && 32 * P * 4 <= 3584 &&
                     (reinterpret_cast<size_t>(x) & 15) == 0;                                     // wave-uniform: 32 real
            if (staged) {                                                                                       // frames of one clip, 16-B aligned
                float* xs = reinterpret_cast<float*>(lds + OFF_XS + wave * 3584);
                const f32x4* src = reinterpret_cast<const f32x4*>(x + row0 * P);
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
            }
        }
        if (!staged) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = tile_row(r, cx.hh);
                xv[r] = (live && f < P) ? x[xrow * P + f] : 0.f;
            }
        }
        make_frag<T16, true>(xv, xf[0]);
        const W* img = reinterpret_cast<const W*>(dm->img_je);
        const float* je_b = reinterpret_cast<const float*>(img + 16 * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = ld_ft(je_b, t, cx.hh);
        gemm_wa<4, 1, T16, true>(h, img, xf, lane);
    }
    stamp(1);
    if constexpr (!FROMH) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) h[t][4 * q + i] += sev[4 * t + q][i];
        if (active) store_h(h, hbuf, g, lane);
    }
    const DcLayer& L = dm->layer[FROMH ? l0 : 0];
    XFrag<T16, SPLIT> nf[4];
    ln_frags<T16, SPLIT>(nf, h);
    stamp(2);
    const W* wk = reinterpret_cast<const W*>(L.img_sa_k);
    const W* wv = reinterpret_cast<const W*>(L.img_sa_v);
    if constexpr (!WGR) {
        front_stage<T16, SPLIT>(nf, wk, wv, reinterpret_cast<const float*>(wk + 32 * WM * 64),
                                reinterpret_cast<const float*>(wv + 32 * WM * 64), cx, M, T, length, recs, active);
    } else {     // workgroup-level record (see wg_* helpers); LDS map above
        float* mx = reinterpret_cast<float*>(lds);
        f32x8* pst = reinterpret_cast<f32x8*>(lds + 8192);
        f32x8* xp = reinterpret_cast<f32x8*>(lds + 8192 + PST_SZ);
        float* ss = reinterpret_cast<float*>(lds + OFF_SS);
        const int ub0 = wm.ub0;
        const W* lk = reinterpret_cast<const W*>(lds + OFF_IMG_K);
        const W* lv = reinterpret_cast<const W*>(lds + OFF_IMG_V);
        const float* bk = reinterpret_cast<const float*>(lk + 32 * WM * 64);
        const float* bv = reinterpret_cast<const float*>(lv + 32 * WM * 64);
        const RowRange vr[2] = {valid_rows_clip(cx, ub0, B, M, T, length, active), valid_rows_clip(cx, ub0 + 1, B, M, T, length, active)};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's image pieces have landed
        __syncthreads();
        stamp(3);
        f32x16 K[4];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            K[oc] = splat(bk[32 * oc + cx.c]);
            mmb_oc<4, 4, T16, SPLIT>(K[oc], lk, oc, nf, lane);
        }
        wg_put_maxes<NW>(K, cx, vr, mx, wave);
        __syncthreads();
        stamp(4);
        const int s0 = cx.b0 - ub0;
        RowRange vr_own = vr[0];
        if (s0) vr_own = vr[1];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            f32x16 V = splat(bv[32 * oc + cx.c]);
            mmb_oc<4, 4, T16, SPLIT>(V, lv, oc, nf, lane);
            float ssum;
            f32x8 keep;
            partial_tile<T16, SPLIT>(K[oc], V, vr_own, wg_colmax<NW>(mx, oc, s0, cx.c), cx, ssum, keep);
            pst[(wave * 4 + oc) * 64 + lane] = keep;
            if (cx.hh == 0) ss[(wave * 4 + oc) * 32 + cx.c] = ssum;
            if (!SPLIT && active && cx.straddle) {
                partial_tile<T16, SPLIT>(K[oc], V, vr[1], wg_colmax<NW>(mx, oc, 1, cx.c), cx, ssum, keep);
                xp[oc * 64 + lane] = keep;
                if (cx.hh == 0) ss[(NW * 4 + oc) * 32 + cx.c] = ssum;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(5);
        __syncthreads();
        stamp(6);
        wg_write_record<NW>(recs, mx, pst, xp, ss, wave, lane, ub0, wm.nact, wm.Mu, wm.Tu, wg);
        stamp(7);
    }
}
template <class T16, bool SPLIT, bool WGR, bool FROMH = false, bool NARROW = false>
__global__ __launch_bounds__((NARROW || (SPLIT && !WGR)) ? 256 : 512, (SPLIT && !WGR) ? 1 : 2)      // (NARROW: as k_layer - no AGPR half)
void k_embed_front(const DcModel* __restrict__ dm, const float* __restrict__ x, float* __restrict__ hbuf, float* __restrict__ recs,
                   const int* __restrict__ length, int M, int T, int G, int B, unsigned long long* __restrict__ clk, int l0, int Tx, int upc) {
    embed_front_body<T16, SPLIT, WGR, FROMH, NARROW>(dm, x, hbuf, recs, length, M, T, G, B, clk, l0, Tx, upc, -1);
}
// The step's two kernels that depend on nothing but x and the conditioning, as ONE launch (default where the layers run wide flat
// units; DC_NO_FUSE_EMBED=1 keeps them apart): the first `ne` workgroups
// embed their unit (k_embed_front's body), then all sweep their share of the FiLM GEMM - one kernel boundary less per step.  The
// GEMM's work shares follow the measured per-XCD speeds, which include the embedding time.
// TS / SP: operand type and split flag of the embedding (the "mixed" mode embeds in split bf16 beside an f16 FiLM GEMM; its
// units are clip-aligned: ea.upc workgroups per clip).
template <class T16, class TS = T16, bool SP = false>
__global__ __launch_bounds__(512, 2) void k_film_embed(const v8<T16>* __restrict__ W, const float* __restrict__ bias16,
                                                       f16x16* __restrict__ E, int G, int NT, int round0, int nround,
                                                       const float* __restrict__ pp, const float* __restrict__ temb,
                                                       const int* __restrict__ t_clip, int T, int B,
                                                       unsigned long long* __restrict__ clk, const float* __restrict__ rate_in,
                                                       float* __restrict__ rate_out, const int* __restrict__ iter_base, const DcEmbedArgs ea,
                                                       int* __restrict__ status) {
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    if constexpr (!SP) {
        if (ea.extra) {
            // small batches (narrow 128-token units): the embedding's units are EXTRA workgroups behind the GEMM's (beside them while the
            // launch fits the chip, dispatched as they retire otherwise: film_extra_workgroups), four of their eight waves at work - the
            // step loses a 15-us launch and a kernel boundary
            const int nf = (int)gridDim.x - ea.ne;
            if ((int)blockIdx.x >= nf) {
                if (threadIdx.x >= 256) return;      // (ended waves leave the workgroup's barriers)
                embed_front_body<TS, false, true, false, true>(ea.dm, ea.x, ea.hbuf, ea.recs, ea.length, ea.M, T, G, B, nullptr, 0, ea.Tx, ea.upc,
                                                               (int)blockIdx.x - nf);
                return;
            }
            film_gemm3_body<T16>(W, bias16, E, G, NT, round0, nround, pp, temb, t_clip, T, B, clk, rate_in, rate_out, iter_base, t_begin, status, nf);
            return;
        }
    }
    if ((int)blockIdx.x < ea.ne) {
        embed_front_body<TS, SP, true, false, false>(ea.dm, ea.x, ea.hbuf, ea.recs, ea.length, ea.M, T, G, B, nullptr, 0, ea.Tx, ea.upc,
                                                     (int)blockIdx.x);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();            // the embedding's LDS use is over before the slab fill
    }
    film_gemm3_body<T16>(W, bias16, E, G, NT, round0, nround, pp, temb, t_clip, T, B, clk, rate_in, rate_out, iter_base, t_begin, status,
                         (int)gridDim.x);
}

// ------------------------------------------------------------------------------------
// one decoder layer for NW token groups (one per wave), from the attention matrices onward:
//   SA back half (transformer.py:104,109,119-121) -> CA (:147,150,156-157) -> FFN (:170-173)
//   then either the next layer's SA front half, or the output projection (:496) fused with the
//   DDIM update (gaussian_diffusion.py:812-830).
// out_mode 0: write pred_xstart to xout;  1: DDIM update of xio in place (+ snapshot).
//
// Structure: NW waves per workgroup (8 = 256 tokens, 2 waves per SIMD; 4 in the split modes, whose
// doubled fragments need the 512-register budget).  The weight images of the GEMM stages stream
// L2 -> LDS by LDS-DMA (global_load_lds, 16 B/lane) one stage ahead into two buffers shared by the
// waves; each stage ends with vmcnt(0) + barrier.  All activations stay in registers, and the
// residual stream h IS the accumulator of the three out-projections (h += W_o * a + b_o).
// ------------------------------------------------------------------------------------
// DBG = true builds the test-hook variant (early exits after a block, leading blocks skipped); the
// production instantiation has none of them - the extra exits alone cost 160 spilled registers.
// NARROW (WGR, non-split, production build only): 4 waves per workgroup = 128-token units, ONE wave per SIMD.  The kernel is
// bound by instruction issue, so a wave that has its SIMD to itself runs the layer in about half the time; worth it
// whenever the batch is small enough for every unit to get its own CU (<= 32 K tokens: e.g. the reference's one clip per call).
#ifndef DC_SPLIT_STYL_PF
#define DC_SPLIT_STYL_PF 2       // split formats: stylization with the FiLM tiles prefetched two k-tiles ahead (0: all at the point of use;
#endif                           // 1: first request at the head of the block; 2: by the preceding stage, in flight across its closer)
#ifndef DC_SPLIT_NW
#define DC_SPLIT_NW 8      // waves per k_layer workgroup in the split modes (4: one wave per SIMD, round 1-2's form)
#endif
template <class T16, bool SPLIT, bool DBG, bool STAMP, bool WGR, bool NARROW = false, bool G1 = false /* FiLM scale tiles hold G' (film_affine) */>
__global__ __launch_bounds__((NARROW || (SPLIT && DC_SPLIT_NW == 4)) ? 256 : 512, (SPLIT && DC_SPLIT_NW == 4) ? 1 : 2)      // (NARROW runs one wave per SIMD by its LDS; bounds of 2 keep hipcc off the AGPR half: 247 VGPRs instead of 247 + 112 and 1 400 accvgpr moves)
void k_layer(const DcModel* __restrict__ dm, int l, float* __restrict__ hbuf, const f16x16* __restrict__ E, int NT,
             const v8<T16>* __restrict__ a_sa /*[B][16][64]*/, const v8<T16>* __restrict__ a_ca /*[L][B][16][64]*/,
             float* __restrict__ recs, const int* __restrict__ length, const float* __restrict__ xin,
             float* __restrict__ xout, int out_mode, const float* __restrict__ coef_cur, const int* __restrict__ snap_cur,
             float* __restrict__ snaps, int M, int T, int G, int B, int dbg,
             unsigned long long* __restrict__ stamps, size_t rec_stride, const int* __restrict__ iter_base,
             int Tx /* frames per clip of xin / xout / snaps (<= the clip stride T) */, int upc /* workgroups per clip (WgMap) or 0 */,
             const DcUpdate upd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(!NARROW || (WGR && !SPLIT && !DBG), "narrow workgroups: workgroup-record form, non-split formats, no test hooks");
    constexpr int NW = NARROW ? 4 : (SPLIT ? DC_SPLIT_NW : 8);
    constexpr int WM = SPLIT ? 2 : 1;            // operand images per matrix (hi [+ lo])
    constexpr int NFW = 32 * WM;
    // diagnostic build aid: 100 MHz timestamps per stage for the waves of workgroup 3 (stamps == nullptr normally)
#define DC_STAMP(k)                                                                                        \
    do {                                                                                                   \
        if (STAMP && stamps && blockIdx.x == 3 && (threadIdx.x & 63) == 0 && l == 3)                       \
            stamps[(threadIdx.x >> 6) * 32 + (k)] = __builtin_amdgcn_s_memrealtime();                      \
    } while (0)
    // ... and begin / end stamps of every workgroup of layers 3 and 4 (two consecutive launches): [layer][workgroup][2] from slot 264
#define DC_WGSTAMP(e)                                                                                      \
    do {                                                                                                   \
        if (STAMP && stamps && threadIdx.x == 0 && (l == 3 || l == 4) && blockIdx.x < 256)                 \
            stamps[264 + ((l - 3) * 256 + blockIdx.x) * 2 + (e)] = __builtin_amdgcn_s_memrealtime();       \
    } while (0)
    constexpr int WSZ = (NFW + 1) * 1024;
    constexpr int OFF_AF = 2 * WSZ;              // attention frags: non-split 8 hi frags of each of the workgroup's <= 2 clips; split (clip-aligned
                                                 // units, one clip per workgroup) its 8 hi + 8 lo frags (16 KiB either way)
    constexpr int OFF_ER = OFF_AF + 16384;       // non-split: per-wave FiLM tile rings (8 KiB each)
    constexpr int OFF_SS = SPLIT ? OFF_AF + 16384 : OFF_ER + 8 * 8192;    // column sums of the workgroup record (4.5 KiB)
    static_assert(!(SPLIT && WGR) || NW == 8, "split workgroup records: 8-wave workgroups");
    using W = v8<T16>;
    // (stamped builds: the clock at the kernel's first instruction - before the kernel arguments, the model record and the workgroup map
    // have been read - so that a workgroup's start-up can be told from the gap between two launches)
    const unsigned long long t_first = STAMP ? __builtin_amdgcn_s_memrealtime() : 0ull;
#ifndef DC_NO_EARLY_ARGS
    // Every launch starts with a cold scalar cache, and hipcc sinks each kernel-argument load to its first use: the prologue then pays
    // several scalar-memory round trips one after the other (arguments in three batches, among them the grid size) before its first load
    // is issued - 1.7 us per workgroup (profiles/r04_diag_launch_gap.txt).  Naming the arguments the prologue needs as inputs of one
    // asm statement makes them ONE round trip.  (Not `volatile`, kept alive - and early - through `T`, which the workgroup map uses at once:
    // a volatile asm counts as a possible store and
    // turns every scalar load of the model record behind it into a vector load with a vmcnt wait; and not through `l`: an opaque `l`
    // keeps the layer's image-pointer loads from being hoisted in front of the LDS-DMA statements, with the same effect - as does naming
    // `dm` itself: a pointer that has been an asm input counts as escaped, and the LDS-DMA statements' memory clobbers then cover it.)
    {
        const unsigned grid_x = gridDim.x;
        asm("" : "+s"(T) : "s"(l), "s"(hbuf), "s"(E), "s"(recs), "s"(a_ca), "s"(NT), "s"(M), "s"(Tx), "s"(G), "s"(B), "s"(rec_stride), "s"(upc), "s"(grid_x));
    }
#endif
    const int nl = dm->num_layers;
    f32x16 h[4];
    {
        const int g0 = wg_map(WGR ? wg_index() : (int)blockIdx.x, threadIdx.x >> 6, NW, G, M, T, B, WGR ? upc : 0).g;
        load_h(h, hbuf, g0, threadIdx.x & 63);      // in flight across the first prologue
    }
  {
    const int tid_ = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), lane = tid_ & 63;
    const int wg = WGR ? wg_index() : (int)blockIdx.x;
    const WgMap wm = wg_map(wg, wave, NW, G, M, T, B, WGR ? upc : 0);
    const int g = wm.g;
    const bool active = wm.active;               // idle waves still take part in the staging and barriers
    const GroupCtx cx = make_ctx(g, lane, M, T);
    char* buf0 = lds;
    char* buf1 = lds + WSZ;
    const W* w0 = reinterpret_cast<const W*>(buf0);
    const W* w1 = reinterpret_cast<const W*>(buf1);
    const float* c0 = reinterpret_cast<const float*>(buf0 + NFW * 1024);    // constants block of the image in buf0
    const float* c1 = reinterpret_cast<const float*>(buf1 + NFW * 1024);
    // attention frags come through LDS when the workgroup can span at most 2 clips, else straight from L2
    const bool wg_lds = WGR || (!SPLIT && T >= NW * 32);       // WGR is only launched with T >= NW * 32
    const int ub0 = wm.ub0;
    char* ring = lds + OFF_ER + wave * 8192;
    auto stage_attn = [&](const W* a) {          // frags of clips ub0, ub0+1 -> AF region
        if constexpr (SPLIT) {                   // (workgroup records, clip-aligned: hi + lo frags of the one clip)
            stage_frags<NW>(a + (size_t)ub0 * 16 * 64, lds + OFF_AF, 16, wave, lane);
        } else {
            const int c1i = min(ub0 + 1, B - 1);
            stage_frags<NW>(a + (size_t)ub0 * 16 * 64, lds + OFF_AF, 8, wave, lane);
            stage_frags<NW>(a + (size_t)c1i * 16 * 64, lds + OFF_AF + 8192, 8, wave, lane);
        }
    };
    const W* af = reinterpret_cast<const W*>(lds + OFF_AF);
    const DcLayer& L = dm->layer[l];
    const bool last = l + 1 >= nl;
    const f16x8* Eg = reinterpret_cast<const f16x8*>(E) + ((size_t)g * NT + (size_t)l * 24) * 128;   // 3 blocks x 8 tiles
    const W* acl = a_ca + (size_t)l * B * 16 * 64;
    const float* recs_in = recs + (size_t)(l & 1) * rec_stride;
    float* recs_out = recs + (size_t)((l + 1) & 1) * rec_stride;

    DC_STAMP(0);
    if (STAMP && stamps && blockIdx.x == 3 && (threadIdx.x & 63) == 0 && l == 3) stamps[(threadIdx.x >> 6) * 32 + 26] = __builtin_amdgcn_s_memtime();
    DC_WGSTAMP(0);
    if (STAMP && stamps && threadIdx.x == 0 && (l == 3 || l == 4) && blockIdx.x < 256) stamps[2576 + (l - 3) * 256 + blockIdx.x] = t_first;
    stage_frags<NW>(L.img_sa_q, buf0, NFW + 1, wave, lane);
    if constexpr (NARROW)
        wg_combine_attn_narrow<T16>(recs_in, reinterpret_cast<W*>(lds + OFF_AF), reinterpret_cast<float*>(buf1), ub0, wm.Mu, wm.Tu, tid_, wg);
    else if constexpr (WGR)      // self-attention matrices from the previous layer's workgroup records (scratch: buf1)
        wg_combine_attn<T16, SPLIT>(recs_in, reinterpret_cast<W*>(lds + OFF_AF), reinterpret_cast<float*>(buf1), ub0, B, wm.Mu, wm.Tu, tid_, wg,
                                    (STAMP && stamps && blockIdx.x == 3 && l == 3) ? stamps : nullptr);
    else if (wg_lds)
        stage_attn(a_sa);
    DC_STAMP(14);
    constexpr bool use_ring = !SPLIT;             // FiLM tiles through the per-wave LDS ring (else: registers)
#ifdef DC_PROLOGUE_WAIT_ALL
    if constexpr (use_ring) {
        ering_issue(Eg, 0, ring, lane);
        ering_issue(Eg, 1, ring + 4096, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DC_STAMP(15);
    stage_sync();
#else
    // The first two FiLM ring tiles are this wave's 8 youngest vector-memory operations and are first read in stage 2, behind
    // stage 1's closing vmcnt(0): the prologue waits for everything BUT them (vmcnt counts in issue order) - at kernel start
    // every workgroup pulls its 128 KiB of residual stream at once, and the ring tiles (a third of that burst's bytes) no
    // longer sit between the workgroup and its first barrier.
    if constexpr (use_ring) {
        ering_issue(Eg, 0, ring, lane);
        ering_issue(Eg, 1, ring + 4096, lane);
#ifdef DC_DIAG_NO_ELOAD
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (no ring loads were issued: the 8 youngest operations are others)
#else
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DC_STAMP(15);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads(); sprio<3>();
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef DC_DIAG_HALF_SHIFT
    // diagnostic build (timing only, results invalid: the upper half reads weight images that are being overwritten): waves 4-7 run
    // stages 1-6 ONE STAGE behind waves 0-3 (an extra barrier in front for them, one behind stage 6 for the others), so that the two
    // waves of a SIMD are never in the same stage - the bound on what a staggered form (MI355X_MICROARCH.md, two waves per SIMD, item 9)
    // could gain over the lockstep of the shipped form
    if constexpr (WGR && !SPLIT && !NARROW && !DBG) {
        if (wave >= NW / 2) { for (int i_ = 0; i_ < DC_DIAG_HALF_SHIFT; ++i_) __builtin_amdgcn_s_barrier(); }
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    DC_STAMP(1);

    // ---- stage 1: SA query + attention apply [buf0]; prefetch SA out-proj -> buf1
    stage_frags<NW>(L.img_sa_o, buf1, NFW + 1, wave, lane);
    EPre ep;
    constexpr bool split_pf = DC_SPLIT_STYL_PF && SPLIT && !DBG;      // split formats: FiLM tiles through registers, two k-tiles ahead
    EPf epf;
    if constexpr (use_ring) epre_load(ep, Eg, lane);
    ytile<SPLIT> y[4];
    float y_rstd, y_shift;
    // test hook (DBG builds): (dbg >> 16) & 3 = number of leading blocks of the layer to skip (1: no self-attention,
    // 2: neither attention) - the staging, rings and barriers of a skipped block still run
    const int skip_blocks = DBG ? (dbg >> 16) & 3 : 0;
    if (DBG && skip_blocks >= 1) {
    } else if (wg_lds)     // two instantiations so that each keeps its address space (a generic pointer means flat loads)
        query_attend<T16, SPLIT>(y, y_rstd, y_shift, h, c0, w0, af + (SPLIT ? 0 : (size_t)(cx.b0 - ub0) * 8 * 64),
                                 af + (SPLIT ? 0 : (size_t)(cx.b1 - ub0) * 8 * 64), cx);
    else
        query_attend<T16, SPLIT>(y, y_rstd, y_shift, h, c0, w0, a_sa + (size_t)cx.b0 * 16 * 64,
                                 a_sa + (size_t)cx.b1 * 16 * 64, cx);
    DC_STAMP(2);
    if constexpr (split_pf && DC_SPLIT_STYL_PF == 2) {       // the next block's first two FiLM tile pairs stay in flight across the closer
        __builtin_amdgcn_sched_barrier(0);
        epf_fetch(epf, Eg, 0, 0, lane);
        epf_fetch(epf, Eg, 1, 1, lane);
        stage_sync_keep8();
    } else
        stage_sync();
    DC_STAMP(3);
    if constexpr (use_ring) epre_landed(ep);
    // ---- stage 2: SA stylization [buf1]; prefetch CA query -> buf0 (+ cross-attention frags)
    if constexpr (!use_ring) {
        stage_frags<NW>(L.img_ca_q, buf0, NFW + 1, wave, lane);
        if (wg_lds) stage_attn(acl);
        if (!(DBG && skip_blocks >= 1)) { if constexpr (split_pf) styl_accumulate_pf<T16, SPLIT, DC_SPLIT_STYL_PF == 2>(h, y, y_rstd, y_shift, Eg, epf, c1, w1, lane, cx.hh); else styl_accumulate<T16, SPLIT, G1>(h, y, y_rstd, y_shift, Eg, c1, w1, lane, cx.hh); }
    } else {
        auto next2 = [&]() {
            stage_frags<NW>(L.img_ca_q, buf0, NFW + 1, wave, lane);
            if (wg_lds) stage_attn(acl);
            ering_issue(Eg + 8 * 128, 0, ring, lane);
            ering_issue(Eg + 8 * 128, 1, ring + 4096, lane);
        };
        if (DBG && skip_blocks >= 1)
            next2();
        else
            styl_accumulate_ring<T16, SPLIT, G1>(h, y, y_rstd, y_shift, ep, ring, c1, w1, lane, cx.hh, next2);
    }
    if constexpr (DBG) if ((dbg & 0xff) == 1) { if (active) store_h(h, hbuf, g, lane); return; }   // test hook: stop after self-attention
    DC_STAMP(4);
    stage_sync();
    DC_STAMP(5);
    // ---- stage 3: CA query + attention apply [buf0]; prefetch CA out-proj -> buf1
    stage_frags<NW>(L.img_ca_o, buf1, NFW + 1, wave, lane);
    if constexpr (use_ring) epre_load(ep, Eg + 8 * 128, lane);
    if (DBG && skip_blocks >= 2) {
    } else if (wg_lds)
        query_attend<T16, SPLIT>(y, y_rstd, y_shift, h, c0, w0, af + (SPLIT ? 0 : (size_t)(cx.b0 - ub0) * 8 * 64),
                                 af + (SPLIT ? 0 : (size_t)(cx.b1 - ub0) * 8 * 64), cx);
    else
        query_attend<T16, SPLIT>(y, y_rstd, y_shift, h, c0, w0, acl + (size_t)cx.b0 * 16 * 64,
                                 acl + (size_t)cx.b1 * 16 * 64, cx);
    DC_STAMP(6);
    if constexpr (split_pf && DC_SPLIT_STYL_PF == 2) {       // the next block's first two FiLM tile pairs stay in flight across the closer
        __builtin_amdgcn_sched_barrier(0);
        epf_fetch(epf, Eg + 8 * 128, 0, 0, lane);
        epf_fetch(epf, Eg + 8 * 128, 1, 1, lane);
        stage_sync_keep8();
    } else
        stage_sync();
    if constexpr (use_ring) epre_landed(ep);
    // ---- stage 4: CA stylization [buf1]; prefetch FFN W1|W2 (+ b1|b2) -> buf0
    if constexpr (!use_ring) {
        stage_frags<NW>(L.img_ffn_w1, buf0, 16 * WM, wave, lane);
        stage_frags<NW>(L.img_ffn_w2, buf0 + 16 * WM * 1024, 16 * WM + 1, wave, lane);
        if (!(DBG && skip_blocks >= 2)) { if constexpr (split_pf) styl_accumulate_pf<T16, SPLIT, DC_SPLIT_STYL_PF == 2>(h, y, y_rstd, y_shift, Eg + 8 * 128, epf, c1, w1, lane, cx.hh); else styl_accumulate<T16, SPLIT, G1>(h, y, y_rstd, y_shift, Eg + 8 * 128, c1, w1, lane, cx.hh); }
    } else {
        auto next4 = [&]() {
            stage_frags<NW>(L.img_ffn_w1, buf0, 16 * WM, wave, lane);
            stage_frags<NW>(L.img_ffn_w2, buf0 + 16 * WM * 1024, 16 * WM + 1, wave, lane);
            ering_issue(Eg + 16 * 128, 0, ring, lane);
            ering_issue(Eg + 16 * 128, 1, ring + 4096, lane);
        };
        if (DBG && skip_blocks >= 2)
            next4();
        else
            styl_accumulate_ring<T16, SPLIT, G1>(h, y, y_rstd, y_shift, ep, ring, c1, w1, lane, cx.hh, next4);
    }
    if constexpr (DBG) if ((dbg & 0xff) == 2) { if (active) store_h(h, hbuf, g, lane); return; }   // test hook: stop after cross-attention
    DC_STAMP(7);
    stage_sync();
    DC_STAMP(8);
    // ---- stage 5: FFN [buf0]; prefetch FFN out-proj -> buf1
    stage_frags<NW>(L.img_ffn_o, buf1, NFW + 1, wave, lane);
    if constexpr (use_ring) epre_load(ep, Eg + 16 * 128, lane);
    {
        f32x16 u[2];
        {
            XFrag<T16, SPLIT> hf[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) make_frag<T16, SPLIT>(h[kt], hf[kt]);
#pragma unroll
            for (int t = 0; t < 2; ++t) u[t] = ld_ft(c0, t, cx.hh);                 // b1
            gemm_wa<2, 4, T16, SPLIT>(u, w0, hf, lane);
        }
        sprio<2>();
        XFrag<T16, SPLIT> uf[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f32x2 gg = gelu_erf_pair(u[kt][2 * r], u[kt][2 * r + 1]);
                u[kt][2 * r] = gg.x;
                u[kt][2 * r + 1] = gg.y;
            }
            make_frag<T16, SPLIT>(u[kt], uf[kt]);
        }
        RowStats st;
        sprio<1>();
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                              // one output tile at a time
            if (t == 2) sprio<0>();
            f32x16 yf = ld_ft(c0 + 64, t, cx.hh);                                  // b2
            mma_ot<4, 2, T16, SPLIT>(yf, w0 + 16 * WM * 64, t, uf, lane);
            st.add(yf);
            put_y<SPLIT>(y[t], yf);
        }
        st.finish(y_rstd, y_shift);
    }
    DC_STAMP(9);
    if constexpr (split_pf && DC_SPLIT_STYL_PF == 2) {       // the next block's first two FiLM tile pairs stay in flight across the closer
        __builtin_amdgcn_sched_barrier(0);
        epf_fetch(epf, Eg + 16 * 128, 0, 0, lane);
        epf_fetch(epf, Eg + 16 * 128, 1, 1, lane);
        stage_sync_keep8();
    } else
        stage_sync();
    if constexpr (use_ring) epre_landed(ep);
    // ---- stage 6: FFN stylization [buf1]; prefetch next layer's key projection (or the output projection) -> buf0
    {
        auto next_w = [&]() {
            if (!last)
                stage_frags<NW>(dm->layer[l + 1].img_sa_k, buf0, NFW + 1, wave, lane);
            else
                stage_frags<NW>(dm->img_out, buf0, 17, wave, lane);      // 8 hi + 8 lo frags + bias: always runs split
        };
        if constexpr (!use_ring) {
            next_w();
            { if constexpr (split_pf) styl_accumulate_pf<T16, SPLIT, DC_SPLIT_STYL_PF == 2>(h, y, y_rstd, y_shift, Eg + 16 * 128, epf, c1, w1, lane, cx.hh); else styl_accumulate<T16, SPLIT, G1>(h, y, y_rstd, y_shift, Eg + 16 * 128, c1, w1, lane, cx.hh); }
        } else {
            styl_accumulate_ring<T16, SPLIT, G1>(h, y, y_rstd, y_shift, ep, ring, c1, w1, lane, cx.hh, next_w);
        }
    }
    if constexpr (DBG) if ((dbg & 0xff) == 3) { if (active) store_h(h, hbuf, g, lane); return; }   // test hook: stop after the FFN
    DC_STAMP(10);
    stage_sync();
#ifdef DC_DIAG_HALF_SHIFT
    if constexpr (WGR && !SPLIT && !NARROW && !DBG) {
        if (wave < NW / 2) { for (int i_ = 0; i_ < DC_DIAG_HALF_SHIFT; ++i_) __builtin_amdgcn_s_barrier(); }
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    DC_STAMP(11);

    // The last layer of a step can also do the NEXT step's front work (dc_api.hip, DC_UPD_EMBED_NEXT; production form of the wide non-split
    // kernel, captured and eager loops): x_{t-1} is in this wave's registers when the DDIM update has been applied, so the embedding
    // (joint_embed + sequence_embedding, the same split-operand GEMM in the same order as embed_front_body: h bit-identical) and layer 0's
    // self-attention front half (stage 7 below, with layer 0's images) follow at once - the next step's FiLM launch is then the bare GEMM.
    // What that saves is the front kernel's cold start (x, the embedding image and the sequence rows through a cold cache: 8 of its 22 us)
    // and the difference between its record pass and this kernel's.
    bool embed_next = false;
    if constexpr (WGR && !SPLIT && !NARROW && !DBG) embed_next = last && out_mode == 1 && (upd.flags & DC_UPD_EMBED_NEXT) != 0;
    if (last) {
    if (embed_next) {      // layer 0's key image -> buf1 (the FFN out-projection image, consumed); the embedding image -> the idle FiLM rings
        stage_frags<NW>(dm->layer[0].img_sa_k, buf1, NFW + 1, wave, lane);
        stage_frags<NW>(dm->img_je, lds + OFF_ER, 17, wave, lane);
    }
    // ---- output projection [buf0, split] + DDIM update
    f32x16 x0[1];
    const int P = dm->input_feats;
    // embed_next: x_t of this lane's token is requested in front of the projection, so that its latency runs under the MFMAs and the barrier
    // behind them; the sequence_embedding rows behind the barrier (64 registers: in front of the projection they spill)
    const int e_xb = min(cx.tok, M - 1) / T, e_xn = cx.tok - e_xb * T;
    const bool e_live = cx.tok < M && e_xn < Tx;                      // (frames Tx .. T-1 of a clip stride are padding)
    f32x4 sev[16];
    float xtv[16];
    {
        XFrag<T16, true> hf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) make_frag<T16, true>(h[kt], hf[kt]);
        if (embed_next) {
            const size_t xrow = (size_t)e_xb * Tx + e_xn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = tile_row(r, cx.hh);
                xtv[r] = (active && e_live && f < P) ? xin[xrow * P + f] : 0.f;
            }
        }
        x0[0] = ld_ft(reinterpret_cast<const float*>(buf0 + 16 * 1024), 0, cx.hh);
        gemm_wa<1, 4, T16, true>(x0, w0, hf, lane);
    }
    if (embed_next) {
        if constexpr (WGR && !SPLIT && !NARROW && !DBG) {
            // the two images have landed (nothing else of this wave is in flight: stage 6's closer drained it) and nobody reads the
            // output image in buf0 any more: layer 0's value image goes there, ahead of everything the update and the embedding issue
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); sprio<3>();
            __builtin_amdgcn_sched_barrier(0);
            stage_frags<NW>(dm->layer[0].img_sa_v, buf0, NFW + 1, wave, lane);
            const int xb = e_xb, xn = e_xn;
            const bool live = e_live;
            {
                const float* se = dm->seq_emb + (size_t)(e_live ? e_xn : 0) * DC_D;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sev[4 * t + q] = *reinterpret_cast<const f32x4*>(se + 32 * t + 8 * q + 4 * cx.hh);
            }
            f32x16 xv = splat(0.f);
            if (active && live) {
                const size_t xrow = (size_t)xb * Tx + xn;
                const int ib = iter_base ? *iter_base : 0;
                const float* cc = coef_cur + DC_COEF * ib;
                const int snap = snap_cur[ib];
                const bool noisy = (upd.flags & DC_UPD_NOISY) != 0;
                const float* zrow = nullptr;
                if (noisy) zrow = *upd.zslot + ((upd.flags & DC_UPD_ZSTEP) ? (size_t)0 : (size_t)(iter_base ? upd.step + ib : snap_cur[1]) * B * Tx * P);
                bool bad = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = tile_row(r, cx.hh);
                    if (f < P) {
                        const size_t o = xrow * P + f;
                        const float xnew = ddim_update(x0[0][r], xtv[r], cc, upd.flags, noisy, noisy ? zrow[o] : 0.f, bad);
                        xout[o] = xnew;
                        if (snap >= 0) snaps[(size_t)snap * B * Tx * P + o] = xnew;
                        xv[r] = xnew;
                    }
                }
                if (bad && upd.status) atomicOr(upd.status, DC_STATUS_NONFINITE);
            }
            if (active) {      // (idle waves keep the h they have: their rows are outside every valid range below)
                XFrag<T16, true> xf[1];
                make_frag<T16, true>(xv, xf[0]);
                const W* img = reinterpret_cast<const W*>(lds + OFF_ER);      // (pst, which overlays it, is first written behind stage 7's barrier)
                const float* je_b = reinterpret_cast<const float*>(img + 16 * 64);
#pragma unroll
                for (int t = 0; t < 4; ++t) h[t] = ld_ft(je_b, t, cx.hh);
                gemm_wa<4, 1, T16, true>(h, img, xf, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) h[t][4 * q + i] += sev[4 * t + q][i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    if (!active || cx.tok >= M) return;
    const int xb = cx.tok / T, xn = cx.tok - xb * T;
    if (xn >= Tx) return;                                           // padding frame
    const size_t xrow = (size_t)xb * Tx + xn;                       // row of xin / xout / snaps
    if (out_mode == 0) {
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            if (f < P) {
                xout[xrow * P + f] = x0[0][r];
                bad = bad || !(fabsf(x0[0][r]) <= 3.0e38f);
            }
        }
        if (bad && upd.status) atomicOr(upd.status, DC_STATUS_NONFINITE);
    } else {
        // graph-captured loop: coef_cur / snap_cur point at this step's slot of the per-iteration tables and *iter_base is
        // the iteration at which the graph replay began; otherwise they are the scalars k_begin_step prepared
        const int ib = iter_base ? *iter_base : 0;
        coef_cur += DC_COEF * ib;
        const int snap = snap_cur[ib];
        const bool noisy = (upd.flags & DC_UPD_NOISY) != 0;
        const float* zrow = nullptr;
        if (noisy) zrow = *upd.zslot + ((upd.flags & DC_UPD_ZSTEP) ? (size_t)0 : (size_t)(iter_base ? upd.step + ib : snap_cur[1]) * B * Tx * P);
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            if (f < P) {
                const size_t o = xrow * P + f;
                const float xnew = ddim_update(x0[0][r], xin[o], coef_cur, upd.flags, noisy, noisy ? zrow[o] : 0.f, bad);
                xout[o] = xnew;
                if (snap >= 0) snaps[(size_t)snap * B * Tx * P + o] = xnew;
            }
        }
        if (bad && upd.status) atomicOr(upd.status, DC_STATUS_NONFINITE);
    }
    return;
    }      // (!embed_next)
    }          // (last)
    {
        // (embed_next: the front half of the NEXT STEP's layer 0 - key image in buf1, value image into buf0)
        const DcLayer& Ln = dm->layer[last ? 0 : l + 1];
        char* const bufK = last ? buf1 : buf0;
        char* const bufV = last ? buf0 : buf1;
        const W* const wK = reinterpret_cast<const W*>(bufK);
        const W* const wV = reinterpret_cast<const W*>(bufV);
        const float* const cK = reinterpret_cast<const float*>(bufK + NFW * 1024);
        const float* const cV = reinterpret_cast<const float*>(bufV + NFW * 1024);
        // ---- stage 7: next layer's SA front half: K [buf0] and V [buf1], partial records
        if (!last) stage_frags<NW>(Ln.img_sa_v, bufV, NFW + 1, wave, lane);      // (embed_next: requested behind the output projection)
        // h goes out right behind the value-image DMA: vmcnt counts in issue order, so waiting until only the 16
        // youngest operations (the 16 dwordx4 stores of h) are outstanding means "the image has landed" while the
        // stores keep draining behind the K/V projections.
        __builtin_amdgcn_sched_barrier(0);
#ifdef DC_DIAG_NO_HSTORE
        const bool st_h = false;          // diagnostic build (timing only, results invalid): what the residual-stream stores cost the tail
#else
        const bool st_h = active;
#endif
        // Wide non-split workgroups stagger the stores: the 128 KiB of a workgroup take the CU's store path ~1 us, and issued by
        // all eight waves at once they cost the second wave of each SIMD ~2.5 us of queueing in front of its LayerNorm - on the
        // path to the barrier below (stage stamps).  Waves 4-7 therefore keep h in registers through the key pass and store it
        // in front of the wait (their stores are then still the wave's 16 youngest operations).
#ifdef DC_NO_STORE_STAGGER
        constexpr bool stagger = false;
#else
        constexpr bool stagger = WGR && !SPLIT && !NARROW && !DBG;
#endif
        const bool late_store = stagger && wave >= NW / 2;
        if (st_h && !late_store) store_h(h, hbuf, g, lane);
        __builtin_amdgcn_sched_barrier(0);
        XFrag<T16, SPLIT> nf[4];
        ln_frags<T16, SPLIT>(nf, h);
        DC_STAMP(19);
        if constexpr (WGR) {
            // Workgroup record in one pass over each image.  While the value image is still landing in buf1: keys of all
            // four feature tiles (pairs of MFMA chains), exponentiated against THIS WAVE's column maxima and kept as
            // operand fragments (32 registers); the maxima go to LDS.  One barrier (maxima visible, values landed).  Then
            // values, exp(K-m_w)^T V, and the rescale exp2(m_w - M) to the workgroup maximum applied to the fp32 blocks,
            // staged per wave in the idle FiLM rings; barrier; summed in wave order and written by wg_write_record.
            // split formats (no FiLM rings; clip-aligned units, so no wave spans two clips and xp stays unused): the staged blocks go
            // to buf0 once its key image is consumed (behind the barrier below), the rescale factors to the upper half of AF
            float* mx = reinterpret_cast<float*>(lds + OFF_AF);
            f32x8* pst = reinterpret_cast<f32x8*>(SPLIT ? bufK : lds + OFF_ER);
            f32x8* xp = reinterpret_cast<f32x8*>(lds + OFF_AF + 8192);
            float* ss = reinterpret_cast<float*>(lds + OFF_SS);
            float* scw = reinterpret_cast<float*>(SPLIT ? lds + OFF_AF + 8192 : bufK) + wave * 2 * 4 * 32;   // this wave's rescale factors
                                                   // [2 slots][4 oc][32 cols]; (buf0's key image is consumed before the barrier below)
            const RowRange vr0 = valid_rows_clip(cx, ub0, B, M, T, length, active);
            const RowRange vr1 = valid_rows_clip(cx, ub0 + 1, B, M, T, length, active);
            const int s0 = cx.b0 - ub0;
            const RowRange vr_own = s0 ? vr1 : vr0;
            const bool strad = !SPLIT && active && cx.straddle;      // (split formats: clip-aligned units)
            XFrag<T16, SPLIT> efA[4], efB[SPLIT ? 1 : 4];
            float ssA[4], ssB[4], mA[4], mB[4];
            auto keys_of = [&](const f32x16& K, const RowRange& rr, XFrag<T16, SPLIT>& ef, float& ssum, float& mcol) {
                float m = -INFINITY;
                const bool full = __builtin_amdgcn_readfirstlane(rr.span) == 32u;
                if (full) {
                    m = max16(K);
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
                    sacc = exp_rows(Ee, K, mz);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        Ee[r] = row_ok(rr, r) ? exp2f_fast(K[r] - mz) : 0.f;
                        sacc += Ee[r];
                    }
                }
                ssum = xhalf_sum(sacc);
                make_frag<T16, SPLIT>(Ee, ef);
            };
            f32x16 Kp[4] = {splat(cK[cx.c]), splat(cK[32 + cx.c]), splat(cK[64 + cx.c]), splat(cK[96 + cx.c])};
            mmb_oc_quad<4, 4, T16, SPLIT>(Kp[0], Kp[1], Kp[2], Kp[3], wK, nf, lane);
            sprio<2, 2>();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = q;
                if (q == 1) sprio<1, 2>();
                if (q == 3) sprio<0, 2>();
                keys_of(Kp[q], vr_own, efA[oc], ssA[oc], mA[oc]);
                mB[oc] = -INFINITY;
                if constexpr (!SPLIT)
                    if (strad) keys_of(Kp[q], vr1, efB[oc], ssB[oc], mB[oc]);
                if (cx.hh == 0) {
                    mx[((oc * 2 + s0) * 32 + cx.c) * NW + wave] = mA[oc];
                    mx[((oc * 2 + (s0 ^ 1)) * 32 + cx.c) * NW + wave] = s0 ? -INFINITY : mB[oc];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            DC_STAMP(20);
            if (st_h && late_store) {
                __builtin_amdgcn_sched_barrier(0);
                store_h(h, hbuf, g, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (st_h)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DC_STAMP(21);
            __syncthreads(); sprio<3>();
            __builtin_amdgcn_sched_barrier(0);
            DC_STAMP(12);
            // rescale factors of this wave's columns, through its own LDS strip (each lane needs 8 of them as row factors)
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                const float fa = mA[oc] == -INFINITY ? 0.f : exp2f_fast(mA[oc] - wg_colmax<NW>(mx, oc, s0, cx.c));
                ssA[oc] *= fa;
                if (cx.hh == 0) scw[(0 * 4 + oc) * 32 + cx.c] = fa;
                if (!SPLIT && strad) {
                    const float fb = mB[oc] == -INFINITY ? 0.f : exp2f_fast(mB[oc] - wg_colmax<NW>(mx, oc, 1, cx.c));
                    ssB[oc] *= fb;
                    if (cx.hh == 0) scw[(1 * 4 + oc) * 32 + cx.c] = fb;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int rowq = 16 * (cx.c >> 4) + 4 * cx.hh;            // kept value j <-> column (row of P) rowq + (j&3) + 8(j>>2)
            auto block_of = [&](const XFrag<T16, SPLIT>& ef, const f32x16& V, const RowRange& rr, const float* sc) {
                f32x16 Vm;
                if (__builtin_amdgcn_readfirstlane(rr.span) == 32u) {
                    Vm = V;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Vm[r] = row_ok(rr, r) ? V[r] : 0.f;
                }
                XFrag<T16, SPLIT> vf;
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
                f32x8 keep = keep_head_block(P, cx.c);
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(sc + rowq), f1 = *reinterpret_cast<const f32x4*>(sc + rowq + 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    keep[i] *= f0[i];
                    keep[4 + i] *= f1[i];
                }
                return keep;
            };
            f32x16 Vp[4] = {splat(cV[cx.c]), splat(cV[32 + cx.c]), splat(cV[64 + cx.c]), splat(cV[96 + cx.c])};
            mmb_oc_quad<4, 4, T16, SPLIT>(Vp[0], Vp[1], Vp[2], Vp[3], wV, nf, lane);
            sprio<2, 2>();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = q;
                if (q == 1) sprio<1, 2>();
                if (q == 3) sprio<0, 2>();
                pst[(wave * 4 + oc) * 64 + lane] = block_of(efA[oc], Vp[q], vr_own, scw + (0 * 4 + oc) * 32);
                if (cx.hh == 0) ss[(wave * 4 + oc) * 32 + cx.c] = ssA[oc];
                if constexpr (!SPLIT)
                    if (strad) {
                        xp[oc * 64 + lane] = block_of(efB[oc], Vp[q], vr1, scw + (1 * 4 + oc) * 32);
                        if (cx.hh == 0) ss[(NW * 4 + oc) * 32 + cx.c] = ssB[oc];
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            DC_STAMP(16);
            __syncthreads(); sprio<3>();
            DC_STAMP(17);
            wg_write_record<NW>(recs_out, mx, pst, xp, ss, wave, lane, ub0, wm.nact, wm.Mu, wm.Tu, wg);
        } else {
            f32x16 K[4];                  // keys from buf0 while the value image lands in buf1
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                K[oc] = splat(cK[32 * oc + cx.c]);
                mmb_oc<4, 4, T16, SPLIT>(K[oc], wK, oc, nf, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (st_h)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); sprio<3>();
            __builtin_amdgcn_sched_barrier(0);
            DC_STAMP(12);
            float* rec = recs + (size_t)cx.g * 2 * DC_REC_FLOATS;
            const RowRange valid0 = valid_rows(cx, 0, M, T, length);
            const RowRange valid1 = valid_rows(cx, cx.straddle ? 1 : 0, M, T, length);
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                f32x16 V = splat(cV[32 * oc + cx.c]);
                mmb_oc<4, 4, T16, SPLIT>(V, wV, oc, nf, lane);
                if (active) {
                    emit_partial<T16, SPLIT>(K[oc], V, oc, valid0, rec, cx);
                    if (cx.straddle) emit_partial<T16, SPLIT>(K[oc], V, oc, valid1, rec + DC_REC_FLOATS, cx);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DC_STAMP(13);
        if (STAMP && stamps && blockIdx.x == 3 && (threadIdx.x & 63) == 0 && l == 3) stamps[(threadIdx.x >> 6) * 32 + 27] = __builtin_amdgcn_s_memtime();
        DC_WGSTAMP(1);
        return;
    }
  }
}

// ====================================================================================
// `no_eff` variant: full T x T attention (TemporalSelfAttention / TemporalCrossAttention,
// transformer.py:198-287) instead of the linear form.  Non-split operand formats only.
//
// Work split: a workgroup = 8 query groups of ONE clip (grid = B x WPC); the flat 32-token groups that
// straddle a clip edge are processed by both neighbours, each for its own lanes only (everything per token is
// identical in both; only the attention differs, and stores are per lane).  Keys/values of a clip live in a
// per-clip array of key tiles (tile kt = flat group g_lo(b) + kt, 16 fragments of 1 KiB):
//   frag h      (h < 8): K of head h, A operand of S = K Q^T:  lane (key, kh), element j <-> d = 8(j>>2) + 4kh + (j&3)
//   frag 8+2t+s        : V^T of feature tile t, keys 16s..16s+15 of the tile in accumulator-chaining order
// i.e. exactly make_frag() of the FT-form K tile and of the TF-form V tile, written by the previous kernel's
// front half.  A key tile (16 KiB) reaches LDS by LDS-DMA, double-buffered, and serves all 8 waves.
// Per (key tile, head): one MFMA for the 32 x 32 scores (hd = 16 = one k-step), online softmax over keys
// (registers + lane-half), two MFMAs for P V with the value rows of the other head of the pair zeroed, so both
// heads of a pair accumulate into one FT tile.  The kernel is bound by v_exp_f32 (one per score), not by MFMA.
// Reference quirks kept: the self-attention mask is added along the QUERY axis (-1e5 on every score of a padded
// query row, transformer.py:224) - reproduced including its fp32 rounding; values are never masked; the 1/sqrt(16)
// is folded into the query projection on the host (exact).
// ====================================================================================
#ifndef DC_FULL_SUM_ADDS
#define DC_FULL_SUM_ADDS 2       // a tile's normaliser term: 2 = packed-f16 tree over the operand fragments (7 v_pk_add_f16, f16 mode; -3 % per
                                 // loop, goldens +1e-5), 1 = 16 f32 adds of the unrounded weights, 0 = 8 v_dot2c on the fragments (+3.8 %)
#endif
#ifdef DC_DIAG_FULL_MOVES        // diagnostic build (tools/noeff_moves.py): how often the key loop's lazily moved reference point moves
__device__ unsigned long long g_full_moves[2];      // [0]: (key tile > 0, head, wave) visits, [1]: visits in which the reference point moved
hipError_t dc_full_moves_read(unsigned long long* out, bool reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_full_moves), 16);
    if (e == hipSuccess && reset) {
        const unsigned long long z[2] = {0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_full_moves), z, 16);
    }
    return e;
}
#else
hipError_t dc_full_moves_read(unsigned long long* out, bool) {
    out[0] = out[1] = 0;
    return hipErrorNotSupported;
}
#endif
#define DC_FULL_ZOFF (32768 + 8 * 8192)          // LDS: key-tile double buffer | per-wave query fragments | 8 KiB of zeros
#define DC_FULL_LDS (DC_FULL_ZOFF + 8192)
struct ClipCtx {
    int b, g_lo, nkt, g, c, hh, tok, n;
    bool active, lane_ok;
};
DEV ClipCtx make_clip_ctx(int blk, int WPC, int wave, int lane, int T, int M) {
    ClipCtx x;
    x.b = blk / WPC;
    const int j = blk % WPC;
    x.g_lo = (x.b * T) / 32;
    const int g_hi = ((x.b + 1) * T - 1) / 32;
    x.nkt = g_hi - x.g_lo + 1;
    x.g = x.g_lo + j * 8 + wave;
    x.active = x.g <= g_hi;
    if (!x.active) x.g = g_hi;
    x.c = lane & 31;
    x.hh = lane >> 5;
    x.tok = 32 * x.g + x.c;
    x.n = x.tok - x.b * T;
    x.lane_ok = x.active && x.n >= 0 && x.n < T && x.tok < M;
    return x;
}
DEV void store_h_lanes(const f32x16 (&h)[4], float* __restrict__ hbuf, int g, int lane, bool ok) {
    if (ok) store_h(h, hbuf, g, lane);
}

// K (FT form) and V (TF form) of n = LN(h) -> the 16 fragments of this group's key tile
// SPLIT: the projections on split operands (hi + lo images, three MFMAs per product); the key tile's fragments stay plain 16-bit.
template <class T16, bool SPLIT = false>
DEV void front_full(const XFrag<T16, SPLIT> (&nf)[4], const v8<T16>* __restrict__ wk, const v8<T16>* __restrict__ wv,
                    v8<T16>* __restrict__ tile, int lane, int c, int hh) {
    constexpr int HL = SPLIT ? 2 : 1;
    const float* bk = reinterpret_cast<const float*>(wk + 32 * 64 * HL);     // plain bias[128] behind the 32 (split: 64) fragments
    const float* bv = reinterpret_cast<const float*>(wv + 32 * 64 * HL);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 K;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(bk + 32 * t + 8 * q + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) K[4 * q + i] = v[i];
        }
        mma_ot<4, 4, T16, SPLIT>(K, wk, t, nf, lane);
        XFrag<T16, false> kf;
        make_frag<T16, false>(K, kf);
        tile[(2 * t) * 64 + lane] = kf.hi[0];
        tile[(2 * t + 1) * 64 + lane] = kf.hi[1];
        f32x16 V = splat(bv[32 * t + c]);
        mmb_oc<4, 4, T16, SPLIT>(V, wv, t, nf, lane);
        XFrag<T16, false> vf;
        make_frag<T16, false>(V, vf);
        tile[(8 + 2 * t) * 64 + lane] = vf.hi[0];
        tile[(8 + 2 * t + 1) * 64 + lane] = vf.hi[1];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// y = softmax_keys(Q K^T) V per head for this wave's 32 queries against all keys of the clip.
// wq: query projection image (LayerNorm affine and the 1/4 folded in), bias ftvec behind it (global memory).
// kv: the clip's key-tile array; keys are valid for flat tokens in [key_lo, key_hi).
#ifndef DC_FULL_PRIO
#define DC_FULL_PRIO 1
#endif
#ifndef DC_FULL_PF
#define DC_FULL_PF 2       // key loop: 0 = fragment reads at their MFMAs (round 5), 1 = two heads ahead, 2 = + the common tile software-pipelined inside the wave (-3 % per loop, bit-identical; profiles/r06_ab_full_pf.txt)
#endif
template <class T16, bool SPLIT = false /* the query projection on split operands; scores, weights and values stay plain 16-bit */>
DEV void full_attend(ytile<SPLIT> (&y)[4], float& y_rstd, float& y_shift, const f32x16 (&h)[4],
                     const v8<T16>* __restrict__ wq, const v8<T16>* __restrict__ kv, int nkt, int tok0, int key_lo,
                     int key_hi, bool q_pad, bool any_pad, char* lds, bool active, int wave, int lane, int hh,
                     unsigned long long* stamps = nullptr /* diagnostic builds: two slots of this wave */) {
    constexpr float LOG2E = 1.4426950408889634f;
    XFrag<T16, false> qf[4];
    {
        f32x16 q[4];
        XFrag<T16, SPLIT> nf[4];
        ln_frags<T16, SPLIT>(nf, h);
        const float* bq = reinterpret_cast<const float*>(wq + 32 * 64 * (SPLIT ? 2 : 1));
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] = ld_ft(bq, t, hh);
        gemm_wa<4, 4, T16, SPLIT>(q, wq, nf, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) make_frag<T16, false>(q[t], qf[t]);
    }
    // the 8 query fragments live in this wave's own 8 KiB of LDS during the key loop (32 VGPRs the loop cannot afford:
    // kept in registers they were spilled to scratch and reloaded once per head and key tile)
    v8<T16>* qs = reinterpret_cast<v8<T16>*>(lds + 32768 + wave * 8192);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        qs[(2 * t) * 64 + lane] = qf[t].hi[0];
        qs[(2 * t + 1) * 64 + lane] = qf[t].hi[1];
    }
    f32x16 Y[4];
    float mx[8], ls[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) Y[t] = splat(0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mx[i] = -1e30f;
        ls[i] = 0.f;
    }
    const float qshift = q_pad ? -100000.f : 0.f;
    // the -1e5 of a padded QUERY row only matters to waves that hold such a row (wave-uniform)
    const bool wpad = any_pad && __builtin_amdgcn_ballot_w64(q_pad) != 0;
    constexpr float INV_LOG2E = 0.6931471805599453f;
    auto issue = [&](int kt) {       // 16 fragments of key tile kt -> buffer kt & 1, two per wave
        const v8<T16>* src = kv + (size_t)kt * 16 * 64;
        char* dst = lds + (kt & 1) * 16384;
        lds_dma16(src + (size_t)wave * 64 + lane, dst + wave * 1024);
        lds_dma16(src + (size_t)(8 + wave) * 64 + lane, dst + (8 + wave) * 1024);
    };
    if (stamps && lane == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();       // query projection done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // nobody still reads the buffers (previous attention of this kernel)
    // 8 KiB of zeros: what the lanes of the OTHER head of a pair read in place of their value rows (all of them the same address
    // per fragment: a broadcast), so both heads of a pair accumulate into one FT tile without a select per operand register
    reinterpret_cast<f32x4*>(lds + DC_FULL_ZOFF)[threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0);
    const bool head0_lane = ((lane & 31) >> 4) == 0;
    const char* zb = lds + DC_FULL_ZOFF - 8192;          // + the value fragments' offsets inside a key tile (8192 ..)
    // The reference point m of a head's exponentials rides in the score MFMA itself: a second k-step whose key operand is the
    // constant (1, 0, ..) and whose query operand is (-m, 0, ..) makes the accumulators S - m, so the common case has no
    // subtraction per score (16 v_sub per head and tile against one MFMA + one v_mov).  m is kept as a T16 value for that
    // (any per-query constant cancels in the softmax).  Waves with padded query rows take the unfolded form: the reference's
    // -1e5 is added to the raw score there.
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const bool fold = !wpad;
    const unsigned short one16 = __builtin_bit_cast(unsigned short, (T16)1.f);
    const v8<T16> kone = __builtin_bit_cast(v8<T16>, u32x4{lane < 32 ? (unsigned)one16 : 0u, 0u, 0u, 0u});
    unsigned negm[8];                                      // (-m as T16 in k-slot 0 of lanes 0-31 | zero), per head
#pragma unroll
    for (int i = 0; i < 8; ++i) negm[i] = 0u;
    auto scores = [&](const v8<T16>* fr, int hd) {
        f32x16 S0 = mfma(fr[hd * 64 + lane], qs[hd * 64 + lane], splat(0.f));
        if (fold) S0 = mfma(kone, __builtin_bit_cast(v8<T16>, u32x4{negm[hd], 0u, 0u, 0u}), S0);
        return S0;
    };
    auto scores_fr = [&](const v8<T16>& kf, const v8<T16>& qf, int hd) {      // the same from fragments already in registers
        f32x16 S0 = mfma(kf, qf, splat(0.f));
        if (fold) S0 = mfma(kone, __builtin_bit_cast(v8<T16>, u32x4{negm[hd], 0u, 0u, 0u}), S0);
        return S0;
    };
#ifdef DC_DIAG_FULL_NKT               // diagnostic builds (results invalid): the key loop cut short, to split a layer's time
    nkt = min(nkt, DC_DIAG_FULL_NKT);
#endif
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) {
            issue(kt + 1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifndef DC_DIAG_FULL_NOSYNC           // diagnostic build (timing only, results invalid): the key loop without its two barriers per key tile
        __syncthreads();
#endif
        if (active) {
            const char* frc = lds + (kt & 1) * 16384;
            const v8<T16>* fr = reinterpret_cast<const v8<T16>*>(frc);
            const char* vb[2] = {head0_lane ? frc + lane * 16 : zb, head0_lane ? zb : frc + lane * 16};
            const int k0 = tok0 + 32 * kt;                    // flat token of the tile's row 0
            const bool edge = k0 < key_lo || k0 + 32 > key_hi;
            // Scores arrive in log2 units (log2(e)/4 is folded into the query projection).
            // One head ahead: the next head's score MFMAs are in the pipe while this head's exponentials issue.
            // COMMON (compile time): not the first tile, no clip edge in it, no padded query row in the wave - two copies of the
            // head loop instead of three wave-uniform branches per head.
            auto tile_body = [&](auto common_tag) {
                constexpr bool COMMON = decltype(common_tag)::value;
#if DC_FULL_PF
                // fragment reads two heads ahead of their MFMAs (key / query fragments of head hd + 2 and this head's two value
                // fragments are requested in front of this head's exponentials): the LDS round trip that sat in front of every pair of
                // MFMAs (ds_read, s_waitcnt, MFMA) now runs under the 16 exponentials - 16 more live registers inside the loop
                v8<T16> kfr[2] = {fr[lane], fr[64 + lane]}, qfr[2] = {qs[lane], qs[64 + lane]};
                f32x16 Snext = scores_fr(kfr[0], qfr[0], 0);
#else
                f32x16 Snext = scores(fr, 0);
#endif
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // progress priority across the tile's eight heads (dc_dev.h, sprio): the waves of a workgroup meet at a barrier per
                    // key tile, and under oldest-first arbitration the older wave of a SIMD ran ahead and idled there - same box,
                    // -4.2 % per loop (171.8 -> 164.7 ms, profiles/r04_ab_stage_prio.txt); -DDC_FULL_PRIO=0 builds without it
#if DC_FULL_PRIO
                    if (t == 0) __builtin_amdgcn_s_setprio(3);
                    if (t == 1) __builtin_amdgcn_s_setprio(2);
                    if (t == 2) __builtin_amdgcn_s_setprio(1);
                    if (t == 3) __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
                    for (int sh = 0; sh < 2; ++sh) {
                        const int hd = 2 * t + sh;
                        f32x16 S = Snext;
#if DC_FULL_PF
                        if (hd < 7) Snext = scores_fr(kfr[(hd + 1) & 1], qfr[(hd + 1) & 1], hd + 1);
                        if (hd < 6) {
                            kfr[hd & 1] = fr[(hd + 2) * 64 + lane];
                            qfr[hd & 1] = qs[(hd + 2) * 64 + lane];
                        }
                        const v8<T16> vf0 = *reinterpret_cast<const v8<T16>*>(vb[sh] + (8 + 2 * t) * 1024);
                        const v8<T16> vf1 = *reinterpret_cast<const v8<T16>*>(vb[sh] + (8 + 2 * t + 1) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
#else
                        if (hd < 7) {
                            Snext = scores(fr, hd + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#endif
                        if constexpr (!COMMON) {
                            if (edge) {
                                int kb = k0 + 4 * hh;
                                asm volatile("" : "+v"(kb));          // keeps the 15 key indices of this rare path out of the common one
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int key = kb + (r & 3) + 8 * (r >> 2);
                                    if (key < key_lo || key >= key_hi) S[r] = -1e30f;
                                }
                            }
                            if (wpad) {      // transformer.py:224 in its own units: fl(score + (-1e5)) on the padded query rows
#pragma unroll
                                for (int r = 0; r < 16; ++r) S[r] = fmaf(S[r], INV_LOG2E, qshift) * LOG2E;
                            }
                        }
                        // m only has to keep the exponentials in range, so it is not the exact running maximum: tile 0 fixes it,
                        // and afterwards it moves only when a tile's weights sum to more than 64 in some lane (then a score exceeds
                        // it by up to 6 bits - the weights stay below 2^6 * 16, far inside f16/bf16) - one compare on the sum the
                        // normaliser needs anyway instead of a 16-way maximum per head and tile.
                        XFrag<T16, false> pf;
                        float tsum;
                        auto weights = [&](float off, bool sub) {      // exp2(S - off) as operand fragments + their sum (f32 adds:
                            f32x16 Pw;                                  // v_dot2c on the fragments costs more issue than it saves)
                            if (sub) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) Pw[r] = exp2f_fast(S[r] - off);
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) Pw[r] = exp2f_fast(S[r]);
                            }
                            make_frag<T16, false>(Pw, pf);
#if DC_FULL_SUM_ADDS == 2
                            if constexpr (std::is_same<T16, _Float16>::value)
                                tsum = sum16_pk(pf.hi[0], pf.hi[1]);
                            else {
                                tsum = 0.f;
#pragma unroll
                                for (int r = 0; r < 16; ++r) tsum += Pw[r];
                            }
#elif DC_FULL_SUM_ADDS
                            tsum = 0.f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) tsum += Pw[r];
#else
                            tsum = sum8(pf.hi[1], sum8(pf.hi[0], 0.f));
#endif
                        };
                        if constexpr (COMMON)
                            weights(0.f, false);
                        else if (kt > 0)
                            weights(mx[hd], !fold);
#ifdef DC_DIAG_FULL_MOVES
                        if (kt > 0 && lane == 0) {
                            atomicAdd(&g_full_moves[0], 1ull);
                            if (__builtin_amdgcn_ballot_w64(!(tsum <= 64.f)) != 0) atomicAdd(&g_full_moves[1], 1ull);
                        }
#endif
                        if ((!COMMON && kt == 0) || __builtin_amdgcn_ballot_w64(!(tsum <= 64.f)) != 0) {      // (wave-uniform; NaN/inf land here too)
                            const float base = (fold && kt > 0) ? mx[hd] : 0.f;             // what S already has taken off
                            float mt = S[0];
#pragma unroll
                            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, S[r]);
                            mt = xhalf_max(mt) + base;
                            float mn = fmaxf(mx[hd], mt);
                            if (fold) {
                                const T16 m16 = (T16)mn;
                                mn = (float)m16;
                                negm[hd] = lane < 32 ? (unsigned)__builtin_bit_cast(unsigned short, (T16)(-mn)) : 0u;
                            }
                            const float alpha = exp2f_fast(mx[hd] - mn);
                            mx[hd] = mn;
                            ls[hd] *= alpha;
#pragma unroll
                            for (int r = 0; r < 8; ++r) Y[t][8 * sh + r] *= alpha;
                            weights(mn - base, true);
                        }
                        ls[hd] += tsum;
#if DC_FULL_PF
                        Y[t] = mfma(vf0, pf.hi[0], Y[t]);
                        Y[t] = mfma(vf1, pf.hi[1], Y[t]);
#else
                        Y[t] = mfma(*reinterpret_cast<const v8<T16>*>(vb[sh] + (8 + 2 * t) * 1024), pf.hi[0], Y[t]);
                        Y[t] = mfma(*reinterpret_cast<const v8<T16>*>(vb[sh] + (8 + 2 * t + 1) * 1024), pf.hi[1], Y[t]);
#endif
                        __builtin_amdgcn_sched_barrier(0);      // one head at a time: bounds the fragment-read lookahead
                    }
                }
            };
#if DC_FULL_PF == 2
            // The common tile, software-pipelined inside the wave: a head's four MFMAs (the next head's score and reference-point
            // MFMAs, the PREVIOUS head's two P V MFMAs) sit between the quarters of its 16 exponentials instead of in front of them -
            // four exponentials are the 32 cycles the matrix pipe needs per MFMA, so the wave no longer waits out a burst of four MFMAs
            // (in-order issue: ~100 cycles per head) before its vector work.  Same products in the same order per accumulator as
            // tile_body: bit-identical results.
            auto tile_body_pipe = [&]() {
                v8<T16> kfr[2] = {fr[lane], fr[64 + lane]}, qfr[2] = {qs[lane], qs[64 + lane]};
                f32x16 Snext = scores_fr(kfr[0], qfr[0], 0);
                XFrag<T16, false> pfp;
                v8<T16> vfp0, vfp1;
                u32x4 fz = {0u, 0u, 0u, 0u};      // the reference-point operand: one live register group, only its first word changes per head
                asm volatile("" : "+v"(fz));       // (opaque: or its zeros are re-materialised in front of every head's MFMA)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#if DC_FULL_PRIO
                    if (t == 0) __builtin_amdgcn_s_setprio(3);
                    if (t == 1) __builtin_amdgcn_s_setprio(2);
                    if (t == 2) __builtin_amdgcn_s_setprio(1);
                    if (t == 3) __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
                    for (int sh = 0; sh < 2; ++sh) {
                        const int hd = 2 * t + sh;
                        const int tp = (hd - 1) >> 1;                 // the previous head's feature tile
                        const f32x16 S = Snext;
                        f32x16 Pw;
                        if (hd < 7) Snext = mfma(kfr[(hd + 1) & 1], qfr[(hd + 1) & 1], splat(0.f));
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) Pw[r] = exp2f_fast(S[r]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (hd < 7) {
                            fz[0] = negm[hd + 1];
                            Snext = mfma(kone, __builtin_bit_cast(v8<T16>, fz), Snext);
                        }
                        if (hd < 6) {
                            kfr[hd & 1] = fr[(hd + 2) * 64 + lane];
                            qfr[hd & 1] = qs[(hd + 2) * 64 + lane];
                        }
                        const v8<T16> vf0 = *reinterpret_cast<const v8<T16>*>(vb[sh] + (8 + 2 * t) * 1024);
                        const v8<T16> vf1 = *reinterpret_cast<const v8<T16>*>(vb[sh] + (8 + 2 * t + 1) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 4; r < 8; ++r) Pw[r] = exp2f_fast(S[r]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (hd > 0) Y[tp] = mfma(vfp0, pfp.hi[0], Y[tp]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 8; r < 12; ++r) Pw[r] = exp2f_fast(S[r]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (hd > 0) Y[tp] = mfma(vfp1, pfp.hi[1], Y[tp]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 12; r < 16; ++r) Pw[r] = exp2f_fast(S[r]);
                        XFrag<T16, false> pf;
                        float tsum;
                        auto finish = [&](const f32x16& P_) {
                            make_frag<T16, false>(P_, pf);
                            if constexpr (DC_FULL_SUM_ADDS == 2 && std::is_same<T16, _Float16>::value)
                                tsum = sum16_pk(pf.hi[0], pf.hi[1]);
                            else {
                                tsum = 0.f;
#pragma unroll
                                for (int r = 0; r < 16; ++r) tsum += P_[r];
                            }
                        };
                        finish(Pw);
                        if (__builtin_amdgcn_ballot_w64(!(tsum <= 64.f)) != 0) {      // the reference point moves (tile_body's rare path)
                            const float base = mx[hd];
                            float mt = S[0];
#pragma unroll
                            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, S[r]);
                            mt = xhalf_max(mt) + base;
                            const T16 m16 = (T16)fmaxf(mx[hd], mt);
                            const float mn = (float)m16;
                            negm[hd] = lane < 32 ? (unsigned)__builtin_bit_cast(unsigned short, (T16)(-mn)) : 0u;
                            const float alpha = exp2f_fast(mx[hd] - mn);
                            mx[hd] = mn;
                            ls[hd] *= alpha;
#pragma unroll
                            for (int r = 0; r < 8; ++r) Y[t][8 * sh + r] *= alpha;
                            const float off = mn - base;
#pragma unroll
                            for (int r = 0; r < 16; ++r) Pw[r] = exp2f_fast(S[r] - off);
                            finish(Pw);
                        }
                        ls[hd] += tsum;
                        pfp = pf;
                        vfp0 = vf0;
                        vfp1 = vf1;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                Y[3] = mfma(vfp0, pfp.hi[0], Y[3]);
                Y[3] = mfma(vfp1, pfp.hi[1], Y[3]);
            };
            if (kt > 0 && !edge && fold)
                tile_body_pipe();
            else
                tile_body(std::false_type{});
#else
            if (kt > 0 && !edge && fold)
                tile_body(std::true_type{});
            else
                tile_body(std::false_type{});
#endif
        }
#ifndef DC_DIAG_FULL_NOSYNC
        __syncthreads();             // the buffer just read is the next-but-one tile's target
#endif
    }
    if (stamps && lane == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();       // key loop done
    RowStats st;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            const float inv = fast_rcp(xhalf_sum(ls[2 * t + sh]));
#pragma unroll
            for (int r = 0; r < 8; ++r) Y[t][8 * sh + r] *= inv;
        }
        st.add(Y[t]);
        put_y<SPLIT>(y[t], Y[t]);
    }
    st.finish(y_rstd, y_shift);
}

template <class T16, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void k_embed_front_full(const DcModel* __restrict__ dm, const float* __restrict__ x,
                                                             float* __restrict__ hbuf, v8<T16>* __restrict__ kv_next,
                                                             int M, int T, int KT, int WPC) {
    using W = v8<T16>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ClipCtx cx = make_clip_ctx(wg_index(), WPC, wave, lane, T, M);
    if (!cx.active) return;
    const int P = dm->input_feats;
    const bool live = cx.tok < M;
    const int n = live ? cx.tok % T : 0;
    f32x16 h[4];
    {
        XFrag<T16, true> xf[1];
        f32x16 xv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            xv[r] = (live && f < P) ? x[(size_t)cx.tok * P + f] : 0.f;
        }
        make_frag<T16, true>(xv, xf[0]);
        const W* img = reinterpret_cast<const W*>(dm->img_je);
        const float* je_b = reinterpret_cast<const float*>(img + 16 * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = ld_ft(je_b, t, cx.hh);
        gemm_wa<4, 1, T16, true>(h, img, xf, lane);
    }
    const float* se = dm->seq_emb + (size_t)n * DC_D;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(se + 32 * t + 8 * q + 4 * cx.hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) h[t][4 * q + i] += v[i];
        }
    store_h_lanes(h, hbuf, cx.g, lane, cx.lane_ok);
    const DcLayer& L = dm->layer[0];
    XFrag<T16, SPLIT> nf[4];
    ln_frags<T16, SPLIT>(nf, h);
    front_full<T16, SPLIT>(nf, reinterpret_cast<const W*>(L.img_sa_k), reinterpret_cast<const W*>(L.img_sa_v),
                           kv_next + ((size_t)cx.b * KT + (cx.g - cx.g_lo)) * 16 * 64, lane, cx.c, cx.hh);
}

#ifdef DC_FULL_STAMPS          // diagnostic build: s_memrealtime of workgroup 3's waves at the block boundaries of k_layer_full
#define FSTAMP(i)                                                                                                       \
    do {                                                                                                                \
        if (upd.stamps && blockIdx.x == 3 && (threadIdx.x & 63) == 0)                                               \
            upd.stamps[(threadIdx.x >> 6) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define FSTAMP_PTR(i) ((upd.stamps && blockIdx.x == 3) ? upd.stamps + (threadIdx.x >> 6) * 32 + (i) : nullptr)
#else
#define FSTAMP(i) do {} while (0)
#define FSTAMP_PTR(i) nullptr
#endif
// SPLIT (the precise tail's evaluations, EPSILON loops; dc_ddim.h): every 128-wide GEMM - query / key / value projections, the three
// stylization out-projections, the FFN - on split operands through the model record's split stage images; the attention itself
// (scores, weights, values: activations, whose rounding averages out over the keys) stays plain 16-bit.  Same register bound: what does
// not fit goes to scratch - this instantiation runs a few evaluations per loop, or the loops of an EPSILON model.
template <class T16, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void k_layer_full(const DcModel* __restrict__ dm, int l, float* __restrict__ hbuf,
                                                       const f16x16* __restrict__ E, int NT, const v8<T16>* __restrict__ kv_cur,
                                                       v8<T16>* __restrict__ kv_next, const v8<T16>* __restrict__ kv_ca,
                                                       const int* __restrict__ length, const float* __restrict__ xin,
                                                       float* __restrict__ xout, int out_mode, const float* __restrict__ coef_cur,
                                                       const int* __restrict__ snap_cur, float* __restrict__ snaps, int M, int T,
                                                       int B, int KT, int WPC, int stop_after, const DcUpdate upd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using W = v8<T16>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ClipCtx cx = make_clip_ctx(wg_index(), WPC, wave, lane, T, M);
    if ((wg_index() % WPC) * 8 >= cx.nkt) return;          // no query group of the clip falls into this workgroup
    const DcLayer& L = dm->layer[l];
    const bool last = l + 1 >= dm->num_layers;
    const int len = length[cx.b];
    const bool any_pad = len < T;
    const bool q_pad = cx.n >= len;
    const int key_lo = cx.b * T, key_hi = min((cx.b + 1) * T, M);
    const f16x8* Eg = reinterpret_cast<const f16x8*>(E) + ((size_t)cx.g * NT + (size_t)l * 24) * 128;
    constexpr int HL = SPLIT ? 2 : 1;
    auto consts = [](const bf16x8* img) { return reinterpret_cast<const float*>(reinterpret_cast<const W*>(img) + 32 * 64 * HL); };
    f32x16 h[4];
    ytile<SPLIT> y[4];
    float y_rstd, y_shift;
    // ---- self-attention
    FSTAMP(0);
    load_h(h, hbuf, cx.g, lane);
    full_attend<T16, SPLIT>(y, y_rstd, y_shift, h, reinterpret_cast<const W*>(L.img_sa_q), kv_cur + (size_t)cx.b * KT * 16 * 64, cx.nkt,
                     cx.g_lo * 32, key_lo, key_hi, q_pad, any_pad, lds, cx.active, wave, lane, cx.hh, FSTAMP_PTR(8));
    FSTAMP(1);
    load_h(h, hbuf, cx.g, lane);           // not kept live across the key loop
    styl_accumulate<T16, SPLIT>(h, y, y_rstd, y_shift, Eg, consts(L.img_sa_o), reinterpret_cast<const W*>(L.img_sa_o), lane,
                                cx.hh);
    if (stop_after == 1) { store_h_lanes(h, hbuf, cx.g, lane, cx.lane_ok); return; }
    // ---- cross-attention (no mask, transformer.py:244-264)
    FSTAMP(2);
    store_h_lanes(h, hbuf, cx.g, lane, cx.lane_ok);
    full_attend<T16, SPLIT>(y, y_rstd, y_shift, h, reinterpret_cast<const W*>(L.img_ca_q), kv_ca + ((size_t)l * B + cx.b) * KT * 16 * 64,
                     cx.nkt, cx.g_lo * 32, key_lo, key_hi, false, false, lds, cx.active, wave, lane, cx.hh, FSTAMP_PTR(10));
    FSTAMP(3);
    load_h(h, hbuf, cx.g, lane);
    styl_accumulate<T16, SPLIT>(h, y, y_rstd, y_shift, Eg + 8 * 128, consts(L.img_ca_o), reinterpret_cast<const W*>(L.img_ca_o), lane,
                                cx.hh);
    if (stop_after == 2) { store_h_lanes(h, hbuf, cx.g, lane, cx.lane_ok); return; }
    // ---- FFN
    FSTAMP(4);
    {
        const W* w1 = reinterpret_cast<const W*>(L.img_ffn_w1);
        const W* w2 = reinterpret_cast<const W*>(L.img_ffn_w2);
        const float* cb = reinterpret_cast<const float*>(w2 + 16 * 64 * HL);          // b1 ftvec[2] | b2 ftvec[4]
        f32x16 u[2];
        {
            XFrag<T16, SPLIT> hf[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) make_frag<T16, SPLIT>(h[kt], hf[kt]);
#pragma unroll
            for (int t = 0; t < 2; ++t) u[t] = ld_ft(cb, t, cx.hh);
            gemm_wa<2, 4, T16, SPLIT>(u, w1, hf, lane);
        }
        XFrag<T16, SPLIT> uf[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f32x2 gg = gelu_erf_pair(u[kt][2 * r], u[kt][2 * r + 1]);
                u[kt][2 * r] = gg.x;
                u[kt][2 * r + 1] = gg.y;
            }
            make_frag<T16, SPLIT>(u[kt], uf[kt]);
        }
        RowStats st;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 yf = ld_ft(cb + 64, t, cx.hh);
            mma_ot<4, 2, T16, SPLIT>(yf, w2, t, uf, lane);
            st.add(yf);
            put_y<SPLIT>(y[t], yf);
        }
        st.finish(y_rstd, y_shift);
    }
    FSTAMP(5);
    styl_accumulate<T16, SPLIT>(h, y, y_rstd, y_shift, Eg + 16 * 128, consts(L.img_ffn_o), reinterpret_cast<const W*>(L.img_ffn_o), lane,
                                cx.hh);
    FSTAMP(6);
    if (!cx.active) return;
    if (!last || stop_after == 3) {
        store_h_lanes(h, hbuf, cx.g, lane, cx.lane_ok);
        if (last) return;
        XFrag<T16, SPLIT> nf[4];
        ln_frags<T16, SPLIT>(nf, h);
        const DcLayer& Ln = dm->layer[l + 1];
        front_full<T16, SPLIT>(nf, reinterpret_cast<const W*>(Ln.img_sa_k), reinterpret_cast<const W*>(Ln.img_sa_v),
                        kv_next + ((size_t)cx.b * KT + (cx.g - cx.g_lo)) * 16 * 64, lane, cx.c, cx.hh);
        FSTAMP(7);
        return;
    }
    // ---- output projection (always split) + DDIM update (gaussian_diffusion.py:812-830 collapsed)
    f32x16 x0[1];
    {
        const W* wo = reinterpret_cast<const W*>(dm->img_out);
        XFrag<T16, true> hf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) make_frag<T16, true>(h[kt], hf[kt]);
        x0[0] = ld_ft(reinterpret_cast<const float*>(wo + 16 * 64), 0, cx.hh);
        gemm_wa<1, 4, T16, true>(x0, wo, hf, lane);
    }
    if (!cx.lane_ok) return;
    const int P = dm->input_feats;
    bool bad = false;
    if (out_mode == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            if (f < P) {
                xout[(size_t)cx.tok * P + f] = x0[0][r];
                bad = bad || !(fabsf(x0[0][r]) <= 3.0e38f);
            }
        }
    } else {
        const int snap = snap_cur[0];
        const bool noisy = (upd.flags & DC_UPD_NOISY) != 0;
        const float* zrow = nullptr;                                                      // (k_begin_step runs every step on this path)
        if (noisy) zrow = *upd.zslot + ((upd.flags & DC_UPD_ZSTEP) ? (size_t)0 : (size_t)snap_cur[1] * M * P);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = tile_row(r, cx.hh);
            if (f < P) {
                const size_t o = (size_t)cx.tok * P + f;
                const float xn = ddim_update(x0[0][r], xin[o], coef_cur, upd.flags, noisy, noisy ? zrow[o] : 0.f, bad);
                xout[o] = xn;
                if (snap >= 0) snaps[(size_t)snap * M * P + o] = xn;
            }
        }
    }
    if (bad && upd.status) atomicOr(upd.status, DC_STATUS_NONFINITE);
}

// Cross-attention keys/values of every layer as key tiles (one-time per batch): the linear-attention pre-pass's
// projections (split-bf16 from the normalised conditioning operands) emitted as fragments instead of records.
// grid (ceil(G/4), L); a group that straddles a clip edge is written into both clips' arrays.
template <class T16>
__global__ __launch_bounds__(256) void k_cond_ca_kv(const DcModel* __restrict__ dm, const bf16x8* __restrict__ nh_hi,
                                                    const bf16x8* __restrict__ nh_lo, v8<T16>* __restrict__ kv_ca, int M, int T,
                                                    int G, int B, int KT) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.y;
    const DcLayer& L = dm->layer[l];
    const GroupCtx cx = make_ctx(g, lane, M, T);
    f32x16 K[4], V[4];
#pragma unroll
    for (int oc = 0; oc < 4; ++oc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) K[oc][r] = L.ca_bk[32 * oc + tile_row(r, cx.hh)];
        V[oc] = splat(L.ca_bv[32 * oc + cx.c]);
    }
    constexpr int NF = 4 * DC_KS_E;
    for (int ks = 0; ks < DC_KS_E; ++ks) {
        const bf16x8 a = nh_hi[((size_t)g * DC_KS_E + ks) * 64 + lane];
        const bf16x8 al = nh_lo[((size_t)g * DC_KS_E + ks) * 64 + lane];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            const int fi = oc * DC_KS_E + ks;
            const bf16x8 wk = L.ca_wk[fi * 64 + lane];
            const bf16x8 wv = L.ca_wv[fi * 64 + lane];
            K[oc] = mfma(wk, a, K[oc]);                         // FT form: features on registers
            V[oc] = mfma(a, wv, V[oc]);                         // TF form: tokens on registers
            K[oc] = mfma(wk, al, K[oc]);
            V[oc] = mfma(al, wv, V[oc]);
            K[oc] = mfma(L.ca_wk[(NF + fi) * 64 + lane], a, K[oc]);
            V[oc] = mfma(a, L.ca_wv[(NF + fi) * 64 + lane], V[oc]);
        }
    }
    for (int b = cx.b0; b <= cx.b1; ++b) {
        const int g_lo = (b * T) / 32;
        v8<T16>* tile = kv_ca + (((size_t)l * B + b) * KT + (g - g_lo)) * 16 * 64;
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            XFrag<T16, false> kf, vf;
            make_frag<T16, false>(K[oc], kf);
            make_frag<T16, false>(V[oc], vf);
            tile[(2 * oc) * 64 + lane] = kf.hi[0];
            tile[(2 * oc + 1) * 64 + lane] = kf.hi[1];
            tile[(8 + 2 * oc) * 64 + lane] = vf.hi[0];
            tile[(8 + 2 * oc + 1) * 64 + lane] = vf.hi[1];
        }
    }
}

// ------------------------------------------------------------------------------------
// host-callable launchers (declared in dc_launch.h).  fmt: 0 = bf16, 1 = f16.
// ------------------------------------------------------------------------------------
#include "dc_launch.h"

#define LAUNCH_CHECK() (hipGetLastError())
#define DISPATCH(fmt, split, CALL)                       \
    do {                                                 \
        if ((fmt) == 1) {                                \
            using T16 = _Float16;                        \
            if (split) { constexpr bool SP = true; CALL; } else { constexpr bool SP = false; CALL; } \
        } else {                                         \
            using T16 = __bf16;                          \
            if (split) { constexpr bool SP = true; CALL; } else { constexpr bool SP = false; CALL; } \
        }                                                \
    } while (0)

// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute is
// per DEVICE: `done` is a per-kernel bit mask of the devices that already have it (a process may hold samplers on several GPUs).
static hipError_t lds_optin(const void* fn, int bytes, unsigned long long& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev < 64) done |= 1ull << dev;
    return e;
}
static int cu_count() {     // of the current device
    static int n[64] = {0};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!n[dev]) hipDeviceGetAttribute(&n[dev], hipDeviceAttributeMultiprocessorCount, dev);
    return n[dev];
}

// Diagnosis pass behind a DC_STATUS_NONFINITE report: do the FiLM tiles (fp16) hold an inf / nan?  n8 = number of 16-byte pieces.
__global__ __launch_bounds__(256) void k_scan_f16_nonfinite(const u32x8* __restrict__ e, size_t n32, int* __restrict__ status) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n32; i += (size_t)gridDim.x * 256) {
        const u32x8 v = e[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc |= ((v[k] & 0x7c007c00u) + 0x04000400u) & 0x80008000u;      // a half's exponent all ones
    }
    if (acc) atomicOr(status, DC_STATUS_F16_SAT);
}
hipError_t dc_launch_scan_f16(hipStream_t st, const void* e, size_t bytes, int* status) {
    k_scan_f16_nonfinite<<<dim3(2048), dim3(256), 0, st>>>(reinterpret_cast<const u32x8*>(e), bytes / 32, status);
    return hipGetLastError();
}

hipError_t dc_launch_advance_iter(hipStream_t st, int* iter, int k) {
    k_advance_iter<<<1, 1, 0, st>>>(iter, k);
    return hipGetLastError();
}
hipError_t dc_launch_set_ptr(hipStream_t st, const float** slot, const float* p, unsigned long long seed, unsigned long long first) {
    k_set_ptr<<<1, 1, 0, st>>>(slot, p, seed, first);
    return hipGetLastError();
}
hipError_t dc_launch_step_noise(hipStream_t st, float* z, size_t n, unsigned long long seed, const unsigned long long* seed_slot, const int* iter_base,
                                int step, const int* snap_cur, unsigned long long first) {
    const size_t quads = (n + 3) / 4 + 1;             // (a `first` that is not a multiple of 4 touches one quad more)
    const size_t nb = (quads + 255) / 256;
    const unsigned grid = (unsigned)(nb < 4096 ? nb : 4096);
    k_step_noise<<<dim3(grid ? grid : 1), dim3(256), 0, st>>>(z, n, seed, seed_slot, iter_base, step, snap_cur, first);
    return hipGetLastError();
}
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

hipError_t dc_launch_cond_embed(hipStream_t st, int mode, const float* xf, const float* wt, const float* b, float* out_f32,
                                void* out_hi, void* out_lo, int M, int G, int T, int Tx) {
    if (mode == 0)
        hipLaunchKernelGGL(k_cond_embed<0>, dim3(G), dim3(256), 0, st, xf, wt, b, out_f32, (bf16x8*)out_hi, (bf16x8*)out_lo, M, T, Tx);
    else
        hipLaunchKernelGGL(k_cond_embed<1>, dim3(G), dim3(256), 0, st, xf, wt, b, out_f32, (bf16x8*)out_hi, (bf16x8*)out_lo, M, T, Tx);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_ca_partials(hipStream_t st, const DcModel* dm, const void* nh_hi, const void* nh_lo,
                                 float* recs, int M, int T, int G, int L, int Tx) {
    hipLaunchKernelGGL(k_cond_ca_partials, dim3((G + 7) / 8, L), dim3(512), 0, st, dm, (const bf16x8*)nh_hi,
                       (const bf16x8*)nh_lo, recs, M, T, G, Tx);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_cond_pp64(hipStream_t st, const float* xf, const void* wpack, const float* b, float* out_f32, int M, int G, int T, int Tx) {
    static unsigned long long optin_done = 0;
    if (hipError_t e = lds_optin((const void*)k_cond_pp64, 131072, optin_done)) return e;
    hipLaunchKernelGGL(k_cond_pp64, dim3((G + 7) / 8), dim3(512), 131072, st, xf, (const bf16x8*)wpack, b, out_f32, M, T, G, Tx);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_ca_partials64(hipStream_t st, const DcModel* dm, const float* xf, const float* gram, float* rstd, float* recs, int M, int T, int G,
                                   int L, int Tx) {
    const int ntok = G * 32;
    hipLaunchKernelGGL(k_cond_rstd, dim3((ntok + 255) / 256), dim3(256), 0, st, xf, gram, rstd, M, T, Tx, ntok);
    if (hipError_t e = hipGetLastError()) return e;
    static unsigned long long optin_done = 0;
    if (hipError_t e = lds_optin((const void*)k_cond_ca_partials64, 65536, optin_done)) return e;
    hipLaunchKernelGGL(k_cond_ca_partials64, dim3((G + 7) / 8, L), dim3(512), 65536, st, dm, xf, rstd, recs, M, T, G, Tx);
    return LAUNCH_CHECK();
}

hipError_t dc_launch_attn_combine(hipStream_t st, int fmt, const float* recs, void* afrag, int T, int NU, int B, int nset,
                                  int gran) {
    const int ng_max = T / gran + 2;
    if (ng_max > 128) return hipErrorInvalidValue;    // combine holds <= 4 groups per thread (T <= 4032)
    const size_t shm = (size_t)(ng_max * 32 + 32 + 1024 + 4096) * sizeof(float);   // w, z, red, pacc
    if (fmt == 1)
        hipLaunchKernelGGL(k_attn_combine<_Float16>, dim3(B, 4, nset), dim3(1024), shm, st, recs, (f16x8*)afrag, T, NU, B, gran);
    else
        hipLaunchKernelGGL(k_attn_combine<__bf16>, dim3(B, 4, nset), dim3(1024), shm, st, recs, (bf16x8*)afrag, T, NU, B, gran);
    return LAUNCH_CHECK();
}

template <class T16, bool SP>
static void launch_silu_t(hipStream_t st, const float* pp, const float* temb, const int* t_clip, void* s_hi, void* s_lo,
                          int G, int T, int B, const int* iter_base) {
    const size_t n = (size_t)G * 32 * 64;
    k_silu_emb<T16, SP><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(pp, temb, t_clip, (v8<T16>*)s_hi,
                                                                                 (v8<T16>*)s_lo, G, T, B, iter_base);
}
hipError_t dc_launch_silu_emb(hipStream_t st, int fmt, bool split, const float* pp, const float* temb, const int* t_clip,
                              void* s_hi, void* s_lo, int G, int T, int B, const int* iter_base) {
    DISPATCH(fmt, split, (launch_silu_t<T16, SP>(st, pp, temb, t_clip, s_hi, s_lo, G, T, B, iter_base)));
    return LAUNCH_CHECK();
}

template <class T16, bool SP>
static void launch_film_t(hipStream_t st, const void* W, const float* bias_ft,
                          const void* s_hi, const void* s_lo, void* E, int G, int NT, int* status) {
    k_film_gemm<T16, SP><<<dim3(NT / 8, (G + 3) / 4), dim3(256), 0, st>>>((const v8<T16>*)W, bias_ft,
                                                                         (const v8<T16>*)s_hi, (const v8<T16>*)s_lo,
                                                                         (f16x16*)E, G, NT, status);
}
// GEMM workgroups of a FiLM launch that also carries `ne` embedding workgroups behind them (DcEmbedArgs::extra; small batches).  While
// both fit the chip and the GEMM keeps its rounds, the GEMM gives up CUs (every workgroup on a CU of its own).  With the chip full the
// GEMM keeps all its workgroups and the embedding's are dispatched as the first GEMM workgroups retire - into the ragged end of the
// GEMM's equal shares where there is one (bs=4 at T=1800: 84 of 256 workgroups hold 2 units, the others 3: -4.9 % per loop), and
// otherwise behind it, which still saves the kernel boundary (same box, bs = 3 ... 16: never slower than two launches,
// profiles/r04_small_batch_fused.txt).
static int film_extra_workgroups(long long nunit, int nblk, int ne, int ncu) {
    const int room = ncu - ne;
    const long long alone = nunit < ncu ? nunit : ncu;
    if (room >= nblk && room >= 1) {
        const int nf = (int)(nunit < room ? nunit : room);
        if ((nunit + nf - 1) / nf == (nunit + alone - 1) / alone) return nf;
    }
    return (int)alone;
}
template <class T16>
static hipError_t launch_film3_t(hipStream_t st, const void* W16, const float* bias16, void* E, int G, int NT, int round0, int nround,
                                 const float* pp, const float* temb, const int* t_clip, int T, int B, unsigned long long* clk,
                                 const float* rate_in, float* rate_out, const int* iter_base, const DcEmbedArgs* ea, int* status) {
    const size_t shm = 4 * DC_KS_E * 1024 + 64;        // slab + the pair counter
    static unsigned long long optin_done = 0;
    if (hipError_t e = lds_optin((const void*)k_film_gemm3<T16>, (int)shm, optin_done)) return e;
    const int ncu = cu_count();
    // One workgroup per CU for big batches; for small ones (the reference samples ONE clip per call: 15 token blocks) the
    // (token block x round) units are spread over more workgroups than there are token blocks, down to one unit each - several
    // workgroups then build the same slab, but a slab fill costs ~6 us and a block's 12 rounds ~60 us when one workgroup sweeps them
    // alone (B=1: 49 -> 36 us per step with one unit per workgroup instead of four).
    const int nblk = (G + 3) / 4;
    const long long nunit = (long long)nblk * nround;
    int nwg = (int)(nunit < ncu ? nunit : ncu);
    if (nwg < nblk) nwg = nblk < ncu ? nblk : ncu;
    if (nwg < 1) nwg = 1;
    if (ea && ea->x && ea->extra) {              // small batches: the embedding's narrow units as extra workgroups of this launch (k_film_embed)
        if (ea->split_bf16 || ea->upc <= 0) return hipErrorInvalidValue;
        const int nf = film_extra_workgroups(nunit, nblk, ea->ne, ncu);
        if (nf <= 0) return hipErrorInvalidValue;
        static unsigned long long optin4 = 0;
        if (hipError_t e = lds_optin((const void*)k_film_embed<T16>, (int)shm, optin4)) return e;
        k_film_embed<T16><<<dim3(nf + ea->ne), dim3(512), shm, st>>>((const v8<T16>*)W16, bias16, (f16x16*)E, G, NT, round0, nround, pp, temb, t_clip, T, B,
                                                                     clk, rate_in, rate_out, iter_base, *ea, status);
        return hipGetLastError();
    }
    if (ea && ea->x && nwg >= ea->ne) {          // fused with k_embed_front (its LDS image is smaller than the slab)
        if (ea->split_bf16) {                    // "mixed": split-bf16 embedding (two 65-KiB images: more LDS than the slab) beside the f16 GEMM
            if (!std::is_same<T16, _Float16>::value || ea->upc <= 0) return hipErrorInvalidValue;
            const size_t shm2 = 8192 + 65 * 1024 + 9 * 4 * 32 * 4 + 65 * 1024;
            static unsigned long long optin3 = 0;
            if (hipError_t e = lds_optin((const void*)k_film_embed<T16, __bf16, true>, (int)shm2, optin3)) return e;
            k_film_embed<T16, __bf16, true><<<dim3(nwg), dim3(512), shm2, st>>>((const v8<T16>*)W16, bias16, (f16x16*)E, G, NT, round0, nround, pp, temb,
                                                                                t_clip, T, B, clk, rate_in, rate_out, iter_base, *ea, status);
            return hipGetLastError();
        }
        static unsigned long long optin2 = 0;
        if (hipError_t e = lds_optin((const void*)k_film_embed<T16>, (int)shm, optin2)) return e;
        k_film_embed<T16><<<dim3(nwg), dim3(512), shm, st>>>((const v8<T16>*)W16, bias16, (f16x16*)E, G, NT, round0, nround, pp, temb, t_clip, T, B,
                                                             clk, rate_in, rate_out, iter_base, *ea, status);
        return hipGetLastError();
    }
    if (ea && ea->x) return hipErrorInvalidValue;      // the caller skipped k_embed_front relying on the fused launch: never drop it silently
    k_film_gemm3<T16><<<dim3(nwg), dim3(512), shm, st>>>((const v8<T16>*)W16, bias16, (f16x16*)E, G, NT, round0, nround,
                                                                           pp, temb, t_clip, T, B, clk, rate_in, rate_out, iter_base, status);
    return hipGetLastError();
}
hipError_t dc_launch_film_gemm(hipStream_t st, int fmt, bool split, const void* W, const float* bias_ft, const void* s_hi, const void* s_lo, void* E, int G, int NT, int round0,
                               int nround, const float* pp, const float* temb, const int* t_clip, int T, int B, unsigned long long* clk,
                               const float* rate_in, float* rate_out, const int* iter_base, const void* W16, const float* bias16,
                               const DcEmbedArgs* embed, int* status) {
    // non-split formats with the fp32 emb image at hand: the S-stationary 16x16x32 kernel builds its operand itself; the split
    // formats (and the test hooks, which read the 16-bit operand image back) use the plain tiled kernel on S_hi / S_lo
    if (!split && pp && W16)
        return fmt == 1 ? launch_film3_t<_Float16>(st, W16, bias16, E, G, NT, round0, nround, pp, temb, t_clip, T, B, clk, rate_in, rate_out, iter_base, embed, status)
                        : launch_film3_t<__bf16>(st, W16, bias16, E, G, NT, round0, nround, pp, temb, t_clip, T, B, clk, rate_in, rate_out, iter_base, embed, status);
    if (embed && embed->x) return hipErrorInvalidValue;      // only the S-stationary form can carry the embedding
    if (round0 != 0) return hipSuccess;        // the plain kernel computes all rounds in its first launch
    DISPATCH(fmt, split, (launch_film_t<T16, SP>(st, W, bias_ft, s_hi, s_lo, E, G, NT, status)));
    return LAUNCH_CHECK();
}

template <class T16, bool SP>
static hipError_t launch_front_from_h_t(hipStream_t st, const DcModel* dm, float* hbuf, float* recs, const int* length,
                                        int M, int T, int G, int B, int l0) {
    constexpr int NW = SP ? 4 : 8;
    k_embed_front<T16, SP, false, true><<<dim3((G + NW - 1) / NW), dim3(NW * 64), 0, st>>>(dm, nullptr, hbuf, recs, length, M, T, G, B, nullptr, l0, T, 0);
    return hipGetLastError();
}
hipError_t dc_launch_front_from_h(hipStream_t st, int fmt, bool split, const DcModel* dm, float* hbuf, float* recs, const int* length,
                                  int M, int T, int G, int B, int l0) {
    hipError_t e = hipSuccess;
    DISPATCH(fmt, split, (e = launch_front_from_h_t<T16, SP>(st, dm, hbuf, recs, length, M, T, G, B, l0)));
    return e;
}

template <class T16, bool SP, bool WGR, bool NARROW = false>
static hipError_t launch_embed_t(hipStream_t st, const DcModel* dm, const float* x, float* hbuf, float* recs, const int* length,
                                 int M, int T, int G, int B, unsigned long long* clk, int Tx, int upc) {
    constexpr int NW = (NARROW || (SP && !WGR)) ? 4 : 8;
    const size_t shm = !WGR ? 0 : SP ? 8192 + 65 * 1024 + 9 * 4 * 32 * 4 + 65 * 1024 : 8192 + 65536 + 8192 + 9 * 4 * 32 * 4 + 33 * 1024;
    if (WGR) {
        static unsigned long long optin_done = 0;
        if (hipError_t e = lds_optin((const void*)k_embed_front<T16, SP, WGR, false, NARROW>, (int)shm, optin_done)) return e;
    }
    k_embed_front<T16, SP, WGR, false, NARROW><<<dim3((WGR && upc) ? B * upc : (G + NW - 1) / NW), dim3(NW * 64), shm, st>>>(dm, x, hbuf, recs, length, M, T, G, B,
                                                                                                                         clk, 0, Tx, WGR ? upc : 0);
    return hipGetLastError();
}
hipError_t dc_launch_embed_front(hipStream_t st, int fmt, bool split, bool wgr, const DcModel* dm, const float* x, float* hbuf,
                                 float* recs, const int* length, int M, int T, int G, int B, unsigned long long* clk, bool narrow, int Tx, int upc) {
    hipError_t e = hipSuccess;
    if (wgr && !split && narrow)
        return fmt == 1 ? launch_embed_t<_Float16, false, true, true>(st, dm, x, hbuf, recs, length, M, T, G, B, clk, Tx, upc)
                        : launch_embed_t<__bf16, false, true, true>(st, dm, x, hbuf, recs, length, M, T, G, B, clk, Tx, upc);
    if (wgr && !split) {
        e = fmt == 1 ? launch_embed_t<_Float16, false, true>(st, dm, x, hbuf, recs, length, M, T, G, B, clk, Tx, upc)
                     : launch_embed_t<__bf16, false, true>(st, dm, x, hbuf, recs, length, M, T, G, B, clk, Tx, upc);
        return e;
    }
    if (wgr) {            // split formats: workgroup records on clip-aligned units only (one clip per workgroup)
        if (upc <= 0) return hipErrorInvalidValue;
        return fmt == 1 ? launch_embed_t<_Float16, true, true>(st, dm, x, hbuf, recs, length, M, T, G, B, clk, Tx, upc)
                        : launch_embed_t<__bf16, true, true>(st, dm, x, hbuf, recs, length, M, T, G, B, clk, Tx, upc);
    }
    DISPATCH(fmt, split, (e = launch_embed_t<T16, SP, false>(st, dm, x, hbuf, recs, length, M, T, G, B, clk, Tx, upc)));
    return e;
}

template <class T16, bool SP, bool DBG, bool STAMP, bool WGR, bool NARROW = false, bool G1 = false>
static hipError_t launch_layer_t(hipStream_t st, const DcModel* dm, int l, float* hbuf, const void* E, int NT,
                                 const void* a_sa, const void* a_ca, float* recs, const int* length, const float* xin,
                                 float* xout, int out_mode, const float* coef_cur, const int* snap_cur, float* snaps,
                                 int M, int T, int G, int B, int dbg, unsigned long long* stamps, size_t rec_stride,
                                 const int* iter_base, int Tx, int upc, const DcUpdate& upd) {
    constexpr int NW = NARROW ? 4 : (SP ? DC_SPLIT_NW : 8);
    // two stage images (+1 KiB constants each); non-split adds the attention-frag region and the FiLM rings
    const size_t shm = SP ? 2 * 65 * 1024 + (WGR ? 16384 + 6144 : 0) : 2 * 33 * 1024 + 16384 + 8 * 8192 + 6144;
    static unsigned long long optin_done = 0;   // > 64 KiB of dynamic LDS needs the opt-in
    if (hipError_t e = lds_optin((const void*)k_layer<T16, SP, DBG, STAMP, WGR, NARROW, G1>, (int)shm, optin_done)) return e;
    k_layer<T16, SP, DBG, STAMP, WGR, NARROW, G1><<<dim3((WGR && upc) ? B * upc : (G + NW - 1) / NW), dim3(NW * 64), shm, st>>>(
        dm, l, hbuf, (const f16x16*)E, NT, (const v8<T16>*)a_sa, (const v8<T16>*)a_ca, recs, length, xin, xout, out_mode, coef_cur, snap_cur,
        snaps, M, T, G, B, dbg, stamps, rec_stride, iter_base, Tx, WGR ? upc : 0, upd);
    return hipGetLastError();
}

hipError_t dc_launch_layer(hipStream_t st, int fmt, bool split, bool wgr, const DcModel* dm, int l, float* hbuf, const void* E, int NT,
                           const void* a_sa, const void* a_ca, float* recs, const int* length, const float* xin,
                           float* xout, int out_mode, const float* coef_cur, const int* snap_cur, float* snaps,
                           int M, int T, int G, int B, int dbg, unsigned long long* stamps, size_t rec_stride, const int* iter_base,
                           bool narrow, int Tx, int upc, const DcUpdate& upd, bool g1) {
    hipError_t e = hipSuccess;
#define LAYER_ARGS st, dm, l, hbuf, E, NT, a_sa, a_ca, recs, length, xin, xout, out_mode, coef_cur, snap_cur, snaps, M, T, G, B, dbg, stamps, \
                   rec_stride, iter_base, Tx, upc, upd
    // (g1: the FiLM scale tiles of this evaluation hold G' - the two production forms of the plain-operand kernel only)
    if (g1 && !(wgr && !split && dbg == 0 && stamps == nullptr)) return hipErrorInvalidValue;
    if (wgr && !split && narrow && dbg == 0 && stamps == nullptr) {      // narrow workgroups: production build only
        if (g1)
            return fmt == 1 ? launch_layer_t<_Float16, false, false, false, true, true, true>(LAYER_ARGS)
                            : launch_layer_t<__bf16, false, false, false, true, true, true>(LAYER_ARGS);
        return fmt == 1 ? launch_layer_t<_Float16, false, false, false, true, true>(LAYER_ARGS)
                        : launch_layer_t<__bf16, false, false, false, true, true>(LAYER_ARGS);
    }
    if (wgr && !split && g1)
        return fmt == 1 ? launch_layer_t<_Float16, false, false, false, true, false, true>(LAYER_ARGS)
                        : launch_layer_t<__bf16, false, false, false, true, false, true>(LAYER_ARGS);
    if (wgr && !split) {        // workgroup-level records + in-kernel combine (non-split formats, T >= 256)
        if (dbg != 0)
            e = fmt == 1 ? launch_layer_t<_Float16, false, true, false, true>(LAYER_ARGS) : launch_layer_t<__bf16, false, true, false, true>(LAYER_ARGS);
        else if (stamps != nullptr)
            e = fmt == 1 ? launch_layer_t<_Float16, false, false, true, true>(LAYER_ARGS) : launch_layer_t<__bf16, false, false, true, true>(LAYER_ARGS);
        else
            e = fmt == 1 ? launch_layer_t<_Float16, false, false, false, true>(LAYER_ARGS) : launch_layer_t<__bf16, false, false, false, true>(LAYER_ARGS);
        return e;
    }
    if (wgr) {            // split formats: workgroup records + in-kernel combine on clip-aligned units (production build only)
        if (upc <= 0 || dbg != 0) return hipErrorInvalidValue;
        if (stamps != nullptr)
            return fmt == 1 ? launch_layer_t<_Float16, true, false, true, true>(LAYER_ARGS) : launch_layer_t<__bf16, true, false, true, true>(LAYER_ARGS);
        return fmt == 1 ? launch_layer_t<_Float16, true, false, false, true>(LAYER_ARGS) : launch_layer_t<__bf16, true, false, false, true>(LAYER_ARGS);
    }
    if (dbg != 0) {
        DISPATCH(fmt, split, (e = launch_layer_t<T16, SP, true, false, false>(LAYER_ARGS)));
    } else if (stamps != nullptr) {
        DISPATCH(fmt, split, (e = launch_layer_t<T16, SP, false, true, false>(LAYER_ARGS)));
    } else {
        DISPATCH(fmt, split, (e = launch_layer_t<T16, SP, false, false, false>(LAYER_ARGS)));
    }
#undef LAYER_ARGS
    return e;
}

// ---- no_eff variant ------------------------------------------------------------------
template <class T16, bool SP>
static hipError_t launch_full_t(hipStream_t st, int which, const DcModel* dm, int l, const float* x, float* hbuf, const void* E,
                                int NT, const void* kv_cur, void* kv_next, const void* kv_ca, const int* length,
                                const float* xin, float* xout, int out_mode, const float* coef_cur, const int* snap_cur,
                                float* snaps, int M, int T, int B, int KT, int stop_after, const DcUpdate& upd) {
    const int WPC = (KT + 7) / 8;
    static unsigned long long optin_done = 0;   // key-tile double buffer (32 KiB) + per-wave query fragments (64 KiB) > 64 KiB of dynamic LDS
    if (hipError_t e = lds_optin((const void*)k_layer_full<T16, SP>, DC_FULL_LDS, optin_done)) return e;
    if (which == 0)
        k_embed_front_full<T16, SP><<<dim3(B * WPC), dim3(512), 0, st>>>(dm, x, hbuf, (v8<T16>*)kv_next, M, T, KT, WPC);
    else
        k_layer_full<T16, SP><<<dim3(B * WPC), dim3(512), DC_FULL_LDS, st>>>(dm, l, hbuf, (const f16x16*)E, NT, (const v8<T16>*)kv_cur,
                                                                       (v8<T16>*)kv_next, (const v8<T16>*)kv_ca, length, xin, xout,
                                                                       out_mode, coef_cur, snap_cur, snaps, M, T, B, KT, WPC,
                                                                       stop_after, upd);
    return hipGetLastError();
}
// (fp16 only: with bf16 scores, weights and values x0 sits 1 - 2e-3 from the reference - dc_sampler_create refuses the combination)
hipError_t dc_launch_embed_front_full(hipStream_t st, int fmt, bool split, const DcModel* dm, const float* x, float* hbuf, void* kv_next,
                                      int M, int T, int B, int KT) {
    if (fmt != 1) return hipErrorInvalidValue;
    if (split)
        return launch_full_t<_Float16, true>(st, 0, dm, 0, x, hbuf, nullptr, 0, nullptr, kv_next, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                             nullptr, M, T, B, KT, 0, DcUpdate{});
    return launch_full_t<_Float16, false>(st, 0, dm, 0, x, hbuf, nullptr, 0, nullptr, kv_next, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                          nullptr, M, T, B, KT, 0, DcUpdate{});
}
hipError_t dc_launch_layer_full(hipStream_t st, int fmt, bool split, const DcModel* dm, int l, float* hbuf, const void* E, int NT,
                                const void* kv_cur, void* kv_next, const void* kv_ca, const int* length, const float* xin,
                                float* xout, int out_mode, const float* coef_cur, const int* snap_cur, float* snaps, int M,
                                int T, int B, int KT, int stop_after, const DcUpdate& upd) {
    if (fmt != 1) return hipErrorInvalidValue;
    if (split)
        return launch_full_t<_Float16, true>(st, 1, dm, l, nullptr, hbuf, E, NT, kv_cur, kv_next, kv_ca, length, xin, xout, out_mode, coef_cur, snap_cur,
                                             snaps, M, T, B, KT, stop_after, upd);
    return launch_full_t<_Float16, false>(st, 1, dm, l, nullptr, hbuf, E, NT, kv_cur, kv_next, kv_ca, length, xin, xout, out_mode, coef_cur, snap_cur,
                                          snaps, M, T, B, KT, stop_after, upd);
}
hipError_t dc_launch_ca_kv(hipStream_t st, int fmt, const DcModel* dm, const void* nh_hi, const void* nh_lo, void* kv_ca,
                           int M, int T, int G, int B, int KT, int L) {
    const dim3 grid((G + 3) / 4, L);
    if (fmt != 1) return hipErrorInvalidValue;
    k_cond_ca_kv<_Float16><<<grid, dim3(256), 0, st>>>(dm, (const bf16x8*)nh_hi, (const bf16x8*)nh_lo, (f16x8*)kv_ca, M, T, G, B, KT);
    return hipGetLastError();
}

// ---- post-processing: Savitzky-Golay smoothing of the sampled poses (tools/visualization.py:20-26, applied with
// kernel=19, order=5 at :126; scipy.signal.savgol_filter, mode="interp") --------------------------------------------------
// Along time, per clip and pose channel: interior frames are a fixed `win`-tap FIR; the first / last win/2 frames are
// the least-squares polynomial of the first / last `win` frames evaluated at their own positions, i.e. rows of the
// same hat matrix H = A pinv(A).  coef: H row-major [win][win] fp32 (row win/2 = the FIR).  One thread per output value;
// consecutive threads walk the channel axis, so every tap is a coalesced row read.  HBM-bound: 2 x 4 bytes per value.
__global__ __launch_bounds__(256) void k_savgol(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ coef,
                                                int B, int T, int P, int win) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * T * P) return;
    const int p = (int)(idx % P);
    const int t = (int)((idx / P) % T);
    const int b = (int)(idx / ((long long)P * T));
    const int h = win >> 1;
    int row, t0;                        // hat-matrix row to apply and first frame of the window it applies to
    if (t < h) {
        row = t;
        t0 = 0;
    } else if (t >= T - h) {
        row = win - (T - t);
        t0 = T - win;
    } else {
        row = h;
        t0 = t - h;
    }
    const float* c = coef + row * win;
    const float* src = x + ((size_t)b * T + t0) * P + p;
    float acc = 0.f;
    for (int k = 0; k < win; ++k) acc = fmaf(c[k], src[(size_t)k * P], acc);
    y[idx] = acc;
}
hipError_t dc_launch_savgol(hipStream_t st, const float* x, float* y, const float* coef, int B, int T, int P, int win) {
    const long long n = (long long)B * T * P;
    k_savgol<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(x, y, coef, B, T, P, win);
    return hipGetLastError();
}
