/*
 * dc_ddim.h - C ABI of libdc_ddim.so: the MI355X (gfx950) DDIM sampler of
 * Diffusion-Conductor's Diffusion_Stage.
 *
 * The reference has no FFI layer: its boundary is three Python call surfaces
 * (SURVEY.md section 8b).  These entry points are what a binding for that path binds
 * *below* those surfaces; each one names the reference code it replaces (paths
 * relative to Diffusion_Stage/).  Plain pointers and sizes only, no torch types.
 *
 * Conventions
 *   - every function returns DC_OK (0) or a negative dc_status; dc_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - `d_` arguments are DEVICE pointers (caller-owned, e.g. tensor.data_ptr());
 *     `h_` arguments are HOST pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Work is
 *     ordered after everything already enqueued on `stream` and later work on `stream`
 *     is ordered after it; calls are asynchronous unless stated otherwise;
 *   - a dc_sampler is not thread-safe; use one per host thread / per GPU;
 *   - there is NO CPU fallback anywhere in this library.
 */
#ifndef DC_DDIM_H
#define DC_DDIM_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table
 * (tests/test_host_logic.py compares `nm -D` with this header). */
#define DC_EXPORT __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dc_sampler dc_sampler;

typedef enum dc_status {
    DC_OK = 0,
    DC_ERR_INVALID = -1,      /* bad argument / wrong call order          */
    DC_ERR_NO_DEVICE = -2,    /* no HIP device visible                    */
    DC_ERR_HIP = -3,          /* a HIP runtime call failed                */
    DC_ERR_PARAM = -4,        /* unknown / missing / mis-sized parameter  */
    DC_ERR_UNSUPPORTED = -5   /* configuration outside the built path     */
} dc_status;

/* MFMA operand precision of the denoiser (accumulation is always fp32; LayerNorm, softmax, SiLU/GELU,
 * FiLM and the DDIM update are fp32).  f16 and bf16 MFMA run at the same rate on gfx950; "split" means
 * x = hi + lo and three MFMAs per product (hi*hi + lo*hi + hi*lo), ~fp32 accurate.  In every mode the
 * two tiny pose projections (joint_embed 26->128, out 128->26) run split, and the one-time
 * cross-attention pre-pass runs split-bf16.  Measured rel-L2 on x0, DDIM-50: see DESIGN.md. */
typedef enum dc_precision {
    DC_PREC_BF16 = 0,   /* every GEMM plain bf16; meets the 1e-3 bound through its precise tail (below) */
    DC_PREC_MIXED = 1,  /* 128-wide GEMMs (Q/K/V, attention, out-proj, FFN) split-bf16; FiLM GEMM f16 */
    DC_PREC_BF16X3 = 2, /* split-bf16 everywhere (validation mode)                                   */
    DC_PREC_FP16 = 3    /* default: every GEMM plain f16, one MFMA per product                       */
} dc_precision;

/* Mirrors the constructor arguments the sampler path consumes:
 * MotionTransformer.__init__ (models/transformer.py:360-374) and
 * tools/visualization.py:169-178 (build_models). */
typedef struct dc_config {
    int32_t input_feats;   /* 26  (dim_pose, 13 joints x 2)            */
    int32_t num_frames;    /* 1800 rows of sequence_embedding          */
    int32_t latent_dim;    /* 128 (only value the reference supports)  */
    int32_t ff_size;       /* 64                                       */
    int32_t num_layers;    /* 8                                        */
    int32_t num_heads;     /* 8                                        */
    int32_t no_eff;        /* 0 = linear attention (default), 1 = full T x T attention (DC_PREC_FP16 only: bf16 attention operands are outside the parity bound) */
    int32_t precision;     /* dc_precision                             */
    int32_t max_timesteps; /* size of the timestep-embedding table (>= diffusion_steps), e.g. 1000 */
    int32_t device;        /* HIP device ordinal                       */
} dc_config;

DC_EXPORT const char* dc_last_error(void);
DC_EXPORT const char* dc_version(void);

/* ------------------------------------------------------------------------------------
 * Host-only helpers (no GPU needed; usable and tested on a CPU-only box).
 * ---------------------------------------------------------------------------------- */

/* get_named_beta_schedule('linear', n) (models/gaussian_diffusion.py:228-245) and the
 * tables GaussianDiffusion.__init__ derives from it (:342-361), all fp64, each [n]. */
DC_EXPORT int dc_linear_beta_schedule(int32_t num_steps, double* h_betas, double* h_alphas_cumprod,
                            double* h_alphas_cumprod_prev, double* h_sqrt_recip_alphas_cumprod,
                            double* h_sqrt_recipm1_alphas_cumprod);

/* The four fp32 scalars ddim_sample (models/gaussian_diffusion.py:812-830) uses at
 * timestep t with eta = 0, from fp64 alphas_cumprod[n]:
 *   h_coef[t*4 + 0] = (float)sqrt(1/abar_t)            (sqrt_recip_alphas_cumprod)
 *   h_coef[t*4 + 1] = (float)sqrt(1/abar_t - 1)        (sqrt_recipm1_alphas_cumprod)
 *   h_coef[t*4 + 2] = sqrtf((float)abar_{t-1})         (coefficient of pred_xstart)
 *   h_coef[t*4 + 3] = sqrtf(1 - (float)abar_{t-1})     (coefficient of eps)          */
DC_EXPORT int dc_ddim_coefficients(int32_t num_steps, const double* h_alphas_cumprod, float* h_coef);

/* The same for any eta >= 0 (models/gaussian_diffusion.py:814-826), eight floats per timestep, fp32 arithmetic on the
 * fp32-rounded table entries in the reference's own order:
 *   h_coef8[t*8 + 0..2] as above;  sigma = eta sqrt((1-abar_{t-1})/(1-abar_t)) sqrt(1-abar_t/abar_{t-1})
 *   h_coef8[t*8 + 3] = sqrtf(1 - abar_{t-1} - sigma^2)   (coefficient of eps)
 *   h_coef8[t*8 + 4] = sigma                             (coefficient of the noise draw; 0 at t = 0 = the reference's nonzero_mask)
 *   h_coef8[t*8 + 5..7] = 0 */
DC_EXPORT int dc_ddim_coefficients_ex(int32_t num_steps, const double* h_alphas_cumprod, float eta, float* h_coef8);

/* Test hook: pack a row-major Linear weight W[n_out][k_in] (torch layout) into the
 * MFMA fragment-major bf16 image the kernels read (see DESIGN.md "weight image").
 * `chained` != 0 uses the accumulator-as-operand k order, 0 the natural k order.
 * h_hi / h_lo receive ceil(n_out/32)*ceil(k_in/32)*2*64*8 uint16 (bf16 bits) each. */
DC_EXPORT int dc_pack_weight(const float* h_w, int32_t n_out, int32_t k_in, int32_t chained,
                   uint16_t* h_hi, uint16_t* h_lo);

/* ------------------------------------------------------------------------------------
 * Sampler object
 * ---------------------------------------------------------------------------------- */

/* Replaces constructing MotionTransformer (models/transformer.py:360-445) +
 * GaussianDiffusion (models/gaussian_diffusion.py:328-379) for sampling. */
DC_EXPORT int dc_sampler_create(const dc_config* cfg, dc_sampler** out);
DC_EXPORT void dc_sampler_destroy(dc_sampler* s);

/* Replaces nn.Module.load_state_dict for the entries of state['encoder']
 * (trainers/ddpm_trainer.py:303-319): call once per float tensor with the reference's
 * own key (e.g. "temporal_decoder_blocks.3.sa_block.query.weight") and its contiguous
 * fp32 data, then dc_sampler_finalize_params.  Unknown keys -> DC_ERR_PARAM. */
DC_EXPORT int dc_sampler_set_param(dc_sampler* s, const char* name, const float* h_data, int64_t numel);

/* Packs all parameters into the device weight image and builds the
 * timestep_embedding + time_embed table (models/transformer.py:8-25, 410-414, 482).
 * Missing entries -> DC_ERR_PARAM.  Synchronous. */
DC_EXPORT int dc_sampler_finalize_params(dc_sampler* s);

/* Step-invariant part of MotionTransformer.forward (models/transformer.py:479-482 and
 * the K/V/attention half of LinearTemporalCrossAttention.forward :149-155): applies
 * `linear` to xf_proj / xf_out, and builds the per-clip cross-attention matrices for
 * all layers.  d_xf_proj, d_xf_out: fp32 [B, T, 64] (the pair encode_music returns);
 * h_length: int32 [B] (model_kwargs['length'], 1 <= length <= T), NULL = all T.  1 <= T <= min(num_frames, 4032); full attention
 * (`no_eff`) needs T >= 32 (DC_ERR_UNSUPPORTED below that).
 * Must be called before dc_sampler_denoise / dc_sampler_ddim_loop; (re)allocates the
 * workspace for (B, T). */
DC_EXPORT int dc_sampler_set_conditioning(dc_sampler* s, const float* d_xf_proj, const float* d_xf_out,
                                const int32_t* h_length, int32_t B, int32_t T, void* stream);

/* MotionTransformer.encode_music in eval mode (models/transformer.py:447-459) with the MusicEncoder conv stack
 * (:289-340) as MFMA kernels: d_mel fp32 [B, Tm, n_mels=128] (the tensor generate_music_motion builds at
 * trainers/ddpm_trainer.py:186-189) -> d_xf_out = music_encoder(mel) and d_xf_proj = proj(d_xf_out), both fp32
 * [B, (Tm-1)/3+1, 64], caller-allocated; Tm >= 4 (the reference's reflection padding raises below that).  Needs the `music_encoder.*` and `proj.*` state_dict entries
 * (optional as a group in dc_sampler_set_param; DC_ERR_PARAM here when they were not supplied). */
DC_EXPORT int dc_sampler_encode_music(dc_sampler* s, const float* d_mel, int32_t B, int32_t Tm, int32_t n_mels,
                            float* d_xf_proj, float* d_xf_out, void* stream);

/* Activation format of the MusicEncoder kernels (no counterpart in the reference, whose encoder is fp32 PyTorch):
 *   DC_ME_SPLIT (0): two bf16 planes per activation and three MFMAs per product - 6e-6 relative at the encoder's output;
 *   DC_ME_FP16  (1): one fp16 plane, one MFMA per product, half the bytes - 3.7e-4 at the encoder's output (every activation is
 *                    rounded to 11 bits once per layer), 1.3e-4 of x0 after DDIM-50, about half the time (2.1 vs 4.2 ms per 32 clips).
 * Default: DC_ME_FP16 for the precisions whose denoiser rounds these features to 16-bit operands anyway (DC_PREC_FP16, DC_PREC_BF16),
 * DC_ME_SPLIT for the split-operand precisions.  The environment variable DC_ME_PREC=f16|split (read per call) overrides both. */
/* Precise tail of the sampling loops (DC_PREC_FP16, DC_PREC_BF16; no counterpart in the reference, which computes in fp32): the last
 * `steps` model evaluations of a loop run their 128-wide GEMMs on SPLIT operands (hi + lo halves of the same 16-bit weight images, three
 * MFMAs per product) instead of plain ones.  What a 16-bit mode loses against the reference is almost entirely the WEIGHTS' rounding
 * in the final evaluations (DDIM's last step returns the model's own prediction of x0).  Golden DDIM-50, rel-L2 of x0:
 *   fp16:  5.0e-4 with steps = 0,  2.3e-4 with 1,  1.6e-4 with 2,  1.2e-4 with 4   (+0.45 % of the loop per step at bs = 32)
 *   bf16:  3.1e-3 with steps = 0,  9.5e-4 with 2,  6.8e-4 with 4,  5.4e-4 with 8   - the bf16-operand mode that meets the 1e-3 bound
 *          (round 5's figures, the tail's FiLM GEMM on bf16 operands).  Since round 6 the bf16 precision's split evaluations take the FiLM
 *          GEMM's operands in fp16 - they are "mixed"-precision evaluations: 1.42e-3 with 1, 8.5e-4 with 2, 5.6e-4 with 4, 3.5e-4 with 8;
 *          default 6 (4 left 1.1e-3 on one short ragged batch of tools/fuzz_shapes.py, 6: 8.6e-4, 8: 7.3e-4)
 * Default (steps never set): 1 for fp16, 6 for bf16 (dc_precise_tail_default).  (Clip strides of whole 32-frame groups run the split evaluations in the
 * workgroup-record form on clip-aligned units, others - T = 900 x 128 unpadded, short clips - in the per-group record form: 5.2e-4 ->
 * 2.6e-4 there at no measurable cost.)  Loops of an EPSILON model (DC_UPDATE_EPSILON) run EVERY evaluation on split operands unless a
 * number was set here: their final sample carries what the plain evaluations left in x_t (fp16, eta = 0: 1.4 - 1.8e-3 with any shorter
 * tail, 1.5e-4 all split; with full attention such a loop is REFUSED at eta = 0 - DC_ERR_UNSUPPORTED: 2.3e-4 ... 1.26e-3 over randomized loops even
 * with every GEMM split, the attention's scores, weights and values being plain fp16 - and runs at eta > 0, <= 2e-4), and so do loops of the bf16 precision over clips of fewer than 100 frames (a clip's error is then a norm over
 * few numbers and the worst clip of a batch of dozens reached 1.27e-3 with the default tail: 39 clips of 36 frames, tools/fuzz_shapes.py;
 * such loops are launch-bound, the split form costs them little).  Applies to linear and full attention alike (`no_eff`, fp16: its split instantiation keeps scores, weights and
 * values plain 16-bit); ignored for the split precisions.  DC_PRECISE_TAIL=k in the environment
 * overrides it.  (In a loop that has a tail the plain
 * evaluations read FiLM scale tiles that hold G' itself - one mixed-precision FMA per element instead of two -, the split ones G' - 1;
 * loops without a tail keep G' - 1 everywhere.) */
DC_EXPORT int dc_sampler_set_precise_tail(dc_sampler* s, int32_t steps);      /* steps = -1: back to the precision's default */
DC_EXPORT int32_t dc_precise_tail_default(int32_t precision);                    /* DC_PREC_FP16: 1, DC_PREC_BF16: see above, others 0 */

/* dc_sampler_denoise (MotionTransformer.forward, transformer.py:469-497) on split operands - the precise tail's evaluation form - when
 * on != 0; default off (plain operands of the precision).  For callers that step a sampler THROUGH single evaluations and whose update
 * keeps the evaluations' error (an EPSILON or PREVIOUS_X model, cond_fn: gaussian_diffusion.py:510-520, 581-603): the Python sampler
 * switches it on for exactly those loops.  The bf16 precision's split evaluations also take their FiLM GEMM operands in fp16 (they are
 * then the evaluations of the "mixed" precision).  Ignored for the split precisions (already split). */
DC_EXPORT int dc_sampler_set_precise_forward(dc_sampler* s, int32_t on);

#define DC_ME_SPLIT 0
#define DC_ME_FP16 1
DC_EXPORT int dc_sampler_set_encoder_format(dc_sampler* s, int32_t format);

/* One MotionTransformer.forward (models/transformer.py:469-497) on the conditioning set
 * above: d_x fp32 [B, T, input_feats], h_timesteps int32 [B] -> d_out fp32 [B, T, input_feats]. */
DC_EXPORT int dc_sampler_denoise(dc_sampler* s, const float* d_x, const int32_t* h_timesteps,
                       float* d_out, void* stream);

/* GaussianDiffusion.ddim_sample_loop (models/gaussian_diffusion.py:871-965) with
 * model_mean_type=START_X, clip_denoised=False, eta=0, cond_fn=denoised_fn=None, as
 * DDPMTrainer.generate_music_motion calls it (trainers/ddpm_trainer.py:190-200).
 *   d_noise  fp32 [B,T,P]  x_T (the `noise=` argument; the caller draws it)
 *   d_out    fp32 [B,T,P]  final sample (== pred_xstart of the last step)
 *   num_steps              diffusion_steps S; timesteps run S-1 .. 0
 *   h_coef   fp32 [S,4]    per-timestep scalars, see dc_ddim_coefficients
 *   h_snap_iters int32[n_snap], d_snaps fp32 [n_snap,B,T,P]: `idxs` - the sample after
 *                          iteration i (0 = after the first step) is also stored; may be NULL/0.
 * The step sequence is captured once into a hipGraph per (B,T,S) and replayed. */
DC_EXPORT int dc_sampler_ddim_loop(dc_sampler* s, const float* d_noise, float* d_out, int32_t num_steps,
                         const float* h_coef, const int32_t* h_snap_iters, int32_t n_snap,
                         float* d_snaps, void* stream);

/* The same loop with the other branches of the reference's sampler on the device (replaces the rest of p_mean_variance /
 * ddim_sample, models/gaussian_diffusion.py:503-521, 812-830; ddim_sample_loop's own default is clip_denoised=True, :876):
 *   h_coef8  fp32 [S,8]    per-timestep scalars from dc_ddim_coefficients_ex (eta folded in: sigma, sqrt(1-abar_prev-sigma^2))
 *   flags                  DC_UPDATE_CLIP_DENOISED: pred_xstart.clamp(-1, 1) (:506-507);
 *                          DC_UPDATE_EPSILON: the denoiser predicts epsilon, pred_xstart = sqrt(1/abar) x_t - sqrt(1/abar-1) out
 *                          (ModelMeanType.EPSILON, :516-521, 539-544)
 *   d_step_noise fp32 [S,B,T,P]  the draws the reference takes with th.randn_like(x) at iteration i = 0 .. S-1 (:822); used when
 *                          any sigma != 0 (eta > 0), ignored (may be NULL) otherwise.  NULL with sigma != 0: the library generates
 *                          the draws step by step (dc_sampler_set_step_noise_seed must have been called).  The tensor's address is
 *                          read through a device slot: another tensor on the next call does not re-capture the hipGraph.
 * Everything else as dc_sampler_ddim_loop; flags = 0 with an eta = 0 table is that call.  denoised_fn / cond_fn are host
 * callbacks and stay on the caller's side of the ABI (per-step dc_sampler_denoise). */
#define DC_UPDATE_CLIP_DENOISED 1
#define DC_UPDATE_EPSILON 2
DC_EXPORT int dc_sampler_ddim_loop_ex(dc_sampler* s, const float* d_noise, float* d_out, int32_t num_steps, const float* h_coef8,
                            int32_t flags, const float* d_step_noise, const int32_t* h_snap_iters, int32_t n_snap,
                            float* d_snaps, void* stream);

/* eta > 0 without a noise tensor.  After this call a dc_sampler_ddim_loop_ex with sigma != 0 and d_step_noise == NULL
 * generates the draws of each iteration itself at the head of the step that consumes them (one [B,T,P] buffer instead of the
 * [S,B,T,P] tensor: 6 GB at S = 1000, bs = 32): Philox4x32-10 keyed by `seed`, counter = (element, iteration), Box-Muller.
 * Draw (seed, iteration, element) does not depend on the batch layout or launch form.  The reference draws with
 * th.randn_like on its own device generator (:822), which no other implementation can reproduce - parity tests pass the
 * draws explicitly (d_step_noise); dc_step_noise_fill writes the library's draws of one iteration into a caller buffer
 * (n = B*T*P elements), so a [S,B,T,P] tensor that reproduces a seeded run can be assembled.
 * A seed serves ONE loop: the dc_sampler_ddim_loop_ex that uses it consumes it, and a later loop with sigma != 0, no tensor and no
 * new seed fails with DC_ERR_INVALID instead of replaying the same draws.
 * dc_sampler_set_step_noise_seed_at: the same for a sampler that holds clips [lo, hi) of a larger batch (one rank of a sharded
 * run, sharding.py): first_element = lo*T*P is the index of its first element in the whole batch's [B,T,P] draw, so that every
 * rank seeded alike draws exactly the rows the unsharded run (gaussian_diffusion.py:822: ONE th.randn_like over the batch) gives
 * its clips - never the same rows on every rank. */
DC_EXPORT int dc_sampler_set_step_noise_seed(dc_sampler* s, uint64_t seed);
DC_EXPORT int dc_sampler_set_step_noise_seed_at(dc_sampler* s, uint64_t seed, uint64_t first_element);
DC_EXPORT int dc_step_noise_fill(float* d_out, int64_t n, uint64_t seed, int32_t iteration, void* stream);

/* Numeric health of the sampler's LAST sampling loop (the word is reset when a loop starts), plus whatever a
 * dc_sampler_denoise since then added (no reference counterpart: the reference computes in fp32).  Waits for the sampler's
 * work, then returns the OR of
 *   DC_STATUS_NONFINITE    a predicted x0 (the denoiser's output) was inf or nan;
 *   DC_STATUS_F16_SATURATED  a FiLM modulation value (StylizationBlock scale / shift, transformer.py:74-78) left the fp16 range
 *                          in which every precision mode stores it: the checkpoint is outside what the library supports.
 *                          Sources: the range check in the epilogue of the split-path FiLM GEMM (bf16x3 / bf16 modes: the bit
 *                          can then appear without NONFINITE), and - in every mode, behind a NONFINITE report - a scan of the
 *                          FiLM tiles of every timestep of the last loop (the GEMM is re-run per timestep on this failure path).
 *                          NONFINITE without this bit in the fp16 mode means an fp16 OPERAND overflowed: use precision "mixed"
 *                          (bf16-range operands; Python: MotionTransformer(precision="auto") switches and re-runs by itself).
 *   DC_STATUS_TIMEOUT      small batches only (B * ceil(T / 64) <= CUs): a workgroup gave up waiting for the slices of its clip's attention
 *                          combine, which the clip's workgroups exchange inside a layer launch - they are co-resident by construction unless
 *                          the GPU is shared with other work; the wait is bounded (~0.1 s) and the loop's results are then invalid.
 *                          One timeout per loop at most: once the bit is set, the loop's remaining launches stop waiting.  Reading
 *                          the bit through dc_sampler_status LATCHES the form without the exchange on this sampler
 *                          (dc_sampler_set_combine_exchange(s, 0)), so re-running the loop gives a valid result; callers that share a
 *                          GPU, or run several small-batch samplers on streams of one process, should select that form up front.
 *                          DC_L16_OWN_COMBINE=1 in the environment selects it for every sampler of the process.
 * clear != 0 resets the word. */
#define DC_STATUS_NONFINITE 1
#define DC_STATUS_F16_SATURATED 2
#define DC_STATUS_TIMEOUT 4
DC_EXPORT int dc_sampler_status(dc_sampler* s, int32_t* h_status, int32_t clear);

/* Batch invariance (the reference's key softmax is per clip, transformer.py:111): on clip-aligned units - no workgroup's tokens span two
 * clips - a clip's result is bit-identical whatever the batch around it, and equal to sampling it alone in the same launch form.  Small
 * batches and the split-operand evaluations always run them.  The wide form (the chip full: bs = 32 x 1800) runs them by default
 * whenever they cost no extra round of workgroups over the chip (mode -1, the library's rule; +1.6 ... +2.1 % per loop at bs = 32 x 1800,
 * 256 instead of 228 workgroups); otherwise flat 256-token units, where a unit that contains a clip edge exponentiates both clips' keys
 * against one maximum and a clip depends on its neighbours at the rounding level (up to 4e-4, inside the parity bound).
 * mode 1: clip-aligned units always; 0: flat units (the throughput form); -1: the rule.  Needs a clip stride of whole 32-frame groups
 * (dc_sampler_clip_stride) and T >= 256; ignored otherwise. */
DC_EXPORT int dc_sampler_set_clip_aligned(dc_sampler* s, int32_t mode);

/* Small batches (B * ceil(T / 64) <= CUs): on != 0 (default) lets a clip's workgroups share the combine of the attention unit records
 * inside a layer launch (-8 % per layer launch at one clip per call; needs the launch's workgroups co-resident, see DC_STATUS_TIMEOUT);
 * on == 0 makes every workgroup combine alone - no in-launch wait, safe beside other work on the GPU.  No reference counterpart
 * (the reference's attention is one einsum, transformer.py:113-116). */
DC_EXPORT int dc_sampler_set_combine_exchange(dc_sampler* s, int32_t on);

/* Timing hook for bench.py: device time (ms, HIP events on the library's own stream)
 * of the last dc_sampler_ddim_loop and the summed duration + launch count of its
 * dominant kernel are not observable from outside a graph, so the library can run the
 * same loop eagerly with per-kernel events.  Fills h_ms[kernel_id] with the total ms
 * and h_count[kernel_id] with launches for each kernel id < n (see dc_kernel_name). */
DC_EXPORT int dc_sampler_profile_loop(dc_sampler* s, const float* d_noise, float* d_out, int32_t num_steps,
                            const float* h_coef, float* h_ms, int32_t* h_count, int32_t n, void* stream);
DC_EXPORT const char* dc_kernel_name(int32_t kernel_id);
DC_EXPORT int32_t dc_kernel_count(void);

/* Test hooks (tests/ only).  dc_sampler_debug_denoise runs dc_sampler_denoise but stops after
 * `n_layers` decoder layers, the last one cut after stage 1 = self-attention, 2 = cross-attention,
 * 3 = FFN (0 = whole layer), leaving the residual stream in the internal buffer "h".
 * dc_sampler_debug_read copies an internal device buffer to the host (synchronous); names:
 * "h" "pp" "s_hi" "s_lo" "E" "recs" "a_sa" "a_ca" "temb" (layouts: DESIGN.md). */
DC_EXPORT int dc_sampler_debug_denoise(dc_sampler* s, const float* d_x, const int32_t* h_timesteps, float* d_out,
                             int32_t n_layers, int32_t stage, void* stream);
DC_EXPORT int dc_sampler_debug_read(dc_sampler* s, const char* what, void* h_out, int64_t nbytes);
/* Test hook: decoder layer `layer` alone on a GIVEN residual stream - h_h is host fp32 [B*T][128] row-major, the `h` a
 * LinearTemporalDiffusionTransformerDecoderLayer.forward receives (models/transformer.py:192-196); emb comes from
 * h_timesteps and the conditioning set before.  Blocks first_stage .. last_stage of the layer run (1 = sa_block,
 * 2 = ca_block, 3 = ffn; 1..3 = the whole layer); the result is left in the internal buffer "h" (dc_sampler_debug_read).
 * Lets the block-level known answers of tests/golden/g3_blocks.npz gate the kernels directly. */
DC_EXPORT int dc_sampler_debug_layer(dc_sampler* s, const float* h_h, const int32_t* h_timesteps, int32_t layer, int32_t first_stage,
                           int32_t last_stage, void* stream);

/* Post-processing of the sampled poses as tools/visualization.py applies it (smooth_motion :20-26, called with kernel=19,
 * order 5 at :126): scipy.signal.savgol_filter(mode="interp") along time for every pose channel.
 * dc_savgol_coefficients: host only; h_coef receives the [window][window] hat matrix of the polynomial fit (row window/2 =
 * the FIR taps, rows 0..window/2-1 and window/2+1.. the edge frames).  dc_savgol_filter: d_in, d_out fp32 [B, T, P] device
 * pointers (distinct), T >= window. */
DC_EXPORT int dc_savgol_coefficients(int32_t window, int32_t order, float* h_coef);
/* The same filter as part of the sampling loop: after this call dc_sampler_ddim_loop / _ex write
 * the SMOOTHED x0 to d_out - the filter reads the loop's final x0 and writes the caller's tensor in place of the plain copy, so
 * smoothing costs no pass of its own.  window = 0 switches it off again; snapshots (`idxs`) stay unsmoothed. */
DC_EXPORT int dc_sampler_set_smoothing(dc_sampler* s, int32_t window, int32_t order);
DC_EXPORT int dc_savgol_filter(const float* d_in, float* d_out, int32_t B, int32_t T, int32_t P, int32_t window, int32_t order,
                     void* stream);

/* Introspection used by tests: bytes of device workspace currently held; frames per clip of the sampler's internal token space
 * (the T of dc_sampler_set_conditioning, padded to whole 32-frame groups where the clip-aligned kernels run): the layout of the
 * buffers dc_sampler_debug_read returns. */
DC_EXPORT int64_t dc_sampler_workspace_bytes(const dc_sampler* s);
DC_EXPORT int32_t dc_sampler_clip_stride(const dc_sampler* s);

#ifdef __cplusplus
}
#endif
#endif /* DC_DDIM_H */
