// dc_common.h - shared host/device definitions for libdc_ddim.so (gfx950 only).
//
// Vocabulary
//   token      one motion frame of one clip; tokens are numbered flat, tok = b*T + n
//   group      32 consecutive tokens = the work of one wavefront (MFMA 32x32 tile width)
//   FT tile    a 32-feature x 32-token fp32 tile in the v_mfma_f32_32x32x16 C/D register
//              layout: lane l, reg r  <->  feature (r&3)+8*(r>>2)+4*(l>>5), token l&31
//   TF tile    the same layout with tokens on rows and features on columns
//   frag       one MFMA A/B operand: 64 lanes x 8 bf16 (1 KiB), stored lane-major so a
//              wave reads it with one coalesced 16-B-per-lane load
#pragma once
#include <stdint.h>

#define DC_D 128        // latent_dim
#define DC_H 8          // heads
#define DC_HD 16        // head dim
#define DC_F 64         // ffn dim
#define DC_E 512        // time_embed_dim == music_latent_dim
#define DC_C 64         // music feature channels
#define DC_PMAX 32      // input_feats padded to one MFMA tile
#define DC_MAX_LAYERS 16
#define DC_FILM_TILES_PER_BLOCK 8   // 256 FiLM outputs = 4 scale tiles + 4 shift tiles
#define DC_KS_E (DC_E / 16)         // 32 k-steps of 16 over the 512-wide embedding

// one partial record of the linear-attention K-softmax / K^T V reduction
// (per group, per clip slot): column max (log2 units), column sum of exp2, and for each of the 4 feature
// tiles the two diagonal 16x16 head blocks of exp(K-m)^T V: lane (c, hh) keeps the 8 accumulator registers
// 8*(c>>4) .. +7 of the 32x32 tile (the other 8 are cross-head products nobody reads).
#define DC_REC_FLOATS (128 + 128 + 4 * 64 * 8)

#ifdef __HIPCC__
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) _Float16 f16x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
#else
struct bf16x8 { uint16_t v[8]; };
#endif

// Stage images: what one GEMM stage of k_layer consumes, laid out exactly as it sits in LDS so that one
// run of LDS-DMA copies brings it in: operand fragments ([hi frags][lo frags], lo always present in the
// file, staged only in split modes) followed by ONE 1-KiB block of fp32 constants (biases).
//   128x128 projection: 32 frags per half, chained pack [kt][ot][s];   constants: bias as ftvec [4][2][16]
//   (K/V projections: plain bias[128], indexed by output feature);    ffn: W1 (16 frags/half) | W2 (16) | b1 ftvec[2], b2 ftvec[4]
// LayerNorm affines preceding a projection are folded into it on the host:
//   W (g*n + b) + c = (W diag(g)) n + (W b + c),  n = (x - mean) * rstd
// and each StylizationBlock's LayerNorm affine is folded into the FiLM GEMM's epilogue.
struct DcLayer {
    const bf16x8 *img_sa_q, *img_sa_o, *img_ca_q, *img_ca_o, *img_ffn_o, *img_sa_k, *img_sa_v;   // 32 frags/half + consts
    const bf16x8 *img_ffn_w1, *img_ffn_w2;     // 16 frags/half each; consts (b1 | b2) follow img_ffn_w2
    const bf16x8 *ca_wk, *ca_wv;               // conditioning pre-pass: natural-k pack [ot][ks], bf16 hi+lo (text_norm folded in)
    const float *ca_bk, *ca_bv;                // plain [128]
    // ... and the same projections composed with `linear` (k_cond_ca_partials64): K = rstd (A x + d) + b on the 64 music features x
    const bf16x8 *ca_ak, *ca_av;               // A = W' Wc  [128][64], natural-k pack [ot 4][ks 4], bf16 hi + lo
    const float *ca_dk, *ca_dv;                // d = W' bc  [128]
};

// Stage images of the 16-token layer kernel for small batches (dc_layer16.hip; non-split formats, linear attention): fragments
// of v_mfma_f32_16x16x32 operands, frag (m, rb) at index m * RB + rb, lane l, element j =
// W[16 rb + (l & 15)][32 m + 16 (j >> 2) + 4 (l >> 4) + (j & 3)]; 32 fragments + 1 KiB of fp32 constants (plain per-feature
// vectors).  Same folded matrices as DcLayer's images.
struct DcLayer16 {
    const bf16x8 *sa_q, *sa_o, *ca_q, *ca_o, *ffn_o, *sa_k, *sa_v;
    const bf16x8* ffn_w;       // W1 (16 fragments) | W2 (16) | b1[64], b2[128]
};

struct DcModel {
    DcLayer layer[DC_MAX_LAYERS];
    DcLayer16 l16[DC_MAX_LAYERS];
    const bf16x8* out16;     // out: 8 hi + 8 lo fragments (2 row blocks x 4 k-steps) + bias[32]
    const bf16x8* img_je;    // joint_embed: chained pack OT=4 KT=1 (8 frags/half, always used split) + bias ftvec[4]
    const float* seq_emb;    // row-major [num_frames][128]
    const bf16x8* img_out;   // out: chained pack OT=1 KT=4 (8 frags/half, always used split) + bias ftvec[1]
    // FiLM operands with the StylizationBlock LayerNorm affine (g, beta) and the emb_layers bias folded in on the host, so the
    // GEMM produces the E tiles directly:  G'-1 = (g (.) W_scale) S + [g (1 + b_scale) - 1],
    //                               log2(e) H'   = log2(e) { (beta (.) W_scale + W_shift) S + [beta (1 + b_scale) + b_shift] }
    const bf16x8* film_w;    // natural-k pack [3*L*8 tiles][32 ks] (tiles interleaved G'0, H'0, G'1, H'1, ...), hi then lo; bf16 or f16 bits
    const float* film_b;     // ftvec [3*L*8 tiles]: the bracketed constants (accumulator initial values), same tile order
    // the same operands for the 16x16x32-MFMA form of the GEMM (k_film_gemm3): [tile][ks32][fb][64][8], lane l =
    // row 16 fb + pi(l & 15), k = 32 ks32 + 8 (l >> 4) + j with pi = (0..3, 8..11, 4..7, 12..15); constants [tile][fb][l >> 4][4]
    const bf16x8* film_w16;
    const float* film_b16;
    const bf16x8* film_w16_tail;                // bf16 precision: the same image in fp16, the FiLM GEMM operand of the precise tail's evaluations (dc_ddim.h)
    const float *film_b_g1, *film_b16_g1;     // the same constants with the scale tiles holding G' instead of G' - 1 (plain-operand consumers)
    const float* lin_wt;     // `linear` weight transposed [64][512]
    const bf16x8* lin_pack;  // `linear` weight [512][64] as natural-k fragments [ot 16][ks 4], bf16 hi + lo (k_cond_pp64)
    const float* lin_gram;   // LayerNorm variance of linear(x) as a quadratic form of x: [64][64] Gc = Wc^T Wc / 512, then gv[64] = Wc^T bc / 512, then c = |bc|^2 / 512
    const float* lin_b;      // [512]
    const float* temb;       // [max_timesteps][512]
    int num_layers;
    int input_feats;
    int num_frames;
    int max_timesteps;
};

// Options of the DDIM update fused into the last layer kernel (gaussian_diffusion.py:503-521, 812-830).  The per-step scalars
// come as 8 floats per timestep: sqrt(1/abar), sqrt(1/abar - 1), sqrt(abar_prev), sqrt(1 - abar_prev - sigma^2), sigma, 0, 0, 0.
#define DC_COEF 8
#define DC_UPD_CLIP 1        // clip_denoised: pred_xstart.clamp(-1, 1)  (:506-507)
#define DC_UPD_EPS 2         // ModelMeanType.EPSILON: pred_xstart = sqrt(1/abar) x_t - sqrt(1/abar - 1) model_out  (:516-521, 539-544)
#define DC_UPD_NOISY 4       // (internal) eta > 0: sigma z is added; the draws are read from *zslot
#define DC_UPD_ZSTEP 8       // (internal) *zslot holds ONE iteration's draws [B][Tx][P], refilled by k_step_noise at the head of every step
#define DC_UPD_TEST_DROP_SLICE 16   // (test hook, DC_L16_TEST_DROP_SLICE=1) workgroup 0 of k_layer16 publishes nothing: its neighbours' bounded wait must end in DC_STATUS_SYNC_TIMEOUT
#define DC_UPD_EMBED_NEXT 32  // (internal) the last layer of this step also embeds x_{t-1} and runs layer 0's self-attention front half for the NEXT step (k_layer, wide non-split production form)
#define DC_STATUS_NONFINITE 1    // a predicted x0 was inf / nan
#define DC_STATUS_F16_SAT 2      // a FiLM modulation value exceeded the fp16 range when stored
#define DC_STATUS_SYNC_TIMEOUT 4 // a workgroup of the small-batch layer kernel gave up waiting for its clip's combine slices (GPU shared?)
struct DcUpdate {
    const float* const* zslot;   // DC_UPD_NOISY: device slot holding the base of the per-iteration noise (the reference's th.randn_like(x)
                         // draws, :822): the caller's [S][B][Tx][P] tensor, or (DC_UPD_ZSTEP) the library's one-iteration buffer.  A slot,
                         // not the address itself, so that a captured graph stays valid when the caller hands in another tensor
    int* status;         // device status word (DC_STATUS_* bits are OR-ed in), or nullptr
    int flags;           // DC_UPD_*
    int step;            // captured loop: step number inside the graph (iteration = step + *iter_base); eager: -1 (iteration = snap_cur[1])
    unsigned long long* stamps;   // diagnostic builds (-DDC_FULL_STAMPS, tools/stage_stamps_full.py): stage stamps of k_layer_full, else nullptr
};

// k_embed_front's arguments when it rides in the FiLM GEMM's launch (the first `ne` workgroups embed one 256-token unit each, flat units)
struct DcEmbedArgs {
    const DcModel* dm;
    const float* x;
    float *hbuf, *recs;
    const int* length;
    int M, Tx, ne;
    int upc;         // 0: flat 256-token units (non-split formats); > 0: clip-aligned units, `upc` workgroups per clip (split formats)
    int split_bf16;  // the embedding runs in the split-bf16 format of the "mixed" mode (FiLM GEMM f16, 128-wide GEMMs bf16x3)
    int extra;       // small batches: the `ne` narrow (128-token, clip-aligned) units are EXTRA workgroups behind the GEMM's
};
