// dc_launch.h - host-callable launchers implemented in dc_kernels.hip.  fmt: 0 = bf16 operands, 1 = f16.
#pragma once
#include <hip/hip_runtime.h>
#include "dc_common.h"

hipError_t dc_launch_begin_step(hipStream_t st, int* iter, const int* t_of_iter, const float* coef_of_t,
                                const int* snap_of_iter, int* t_clip, float* coef_cur, int* snap_cur, int B);
hipError_t dc_launch_temb_table(hipStream_t st, const float* freqs, const float* w0t, const float* b0,
                                const float* w2t, const float* b2, float* temb, int nt);
// `self.linear` of one conditioning tensor [M][64] into a fragment-major operand image: mode 0 = fp32 image of emb's
// step-invariant term (out_f32), mode 1 = text_norm'ed bf16 hi / lo images (out_hi, out_lo) for the cross-attention pre-pass
hipError_t dc_launch_cond_embed(hipStream_t st, int mode, const float* xf, const float* wt, const float* b, float* out_f32,
                                void* out_hi, void* out_lo, int M, int G, int T /* clip stride of the images */,
                                int Tx /* frames per clip of xf (<= T; the rest of a clip's stride is padding) */);
hipError_t dc_launch_ca_partials(hipStream_t st, const DcModel* dm, const void* nh_hi, const void* nh_lo,
                                 float* recs, int M, int T, int G, int L, int Tx /* frames per clip (<= the clip stride T) */);
// MODE 0 of dc_launch_cond_embed (the fp32 image of linear(xf_proj)) on split-bf16 MFMAs
hipError_t dc_launch_cond_pp64(hipStream_t st, const float* xf /*[B][Tx][64]*/, const void* wpack, const float* b, float* out_f32, int M, int G, int T, int Tx);
// the same records from the 64 music features (K = rstd (A x + d) + b'): k_cond_rstd -> rstd [G * 32], then k_cond_ca_partials64
hipError_t dc_launch_ca_partials64(hipStream_t st, const DcModel* dm, const float* xf /*[B][Tx][64]*/, const float* gram, float* rstd, float* recs,
                                   int M, int T, int G, int L, int Tx);
hipError_t dc_launch_attn_combine(hipStream_t st, int fmt, const float* recs, void* afrag, int T, int NU, int B, int nset,
                                  int gran);
hipError_t dc_launch_silu_emb(hipStream_t st, int fmt, bool split, const float* pp, const float* temb, const int* t_clip,
                              void* s_hi, void* s_lo, int G, int T, int B,
                              const int* iter_base = nullptr /* captured loop: t_clip = &t_of_iter[step], indexed by *iter_base */);
hipError_t dc_launch_film_gemm(hipStream_t st, int fmt, bool split, const void* W, const float* bias_ft, const void* s_hi, const void* s_lo, void* E, int G, int NT, int round0,
                               int nround, const float* pp, const float* temb, const int* t_clip, int T, int B,
                               unsigned long long* clk /* diagnostic clock stamps or nullptr */,
                               const float* rate_in, float* rate_out /* per-workgroup speeds of the previous / this launch (num_cu floats) or nullptr */,
                               const int* iter_base /* captured loop: t_clip = &t_of_iter[step], indexed by *iter_base; else nullptr */,
                               const void* W16, const float* bias16 /* operands of the 16x16x32-MFMA form (used with pp, non-split) */,
                               const DcEmbedArgs* embed = nullptr /* S-stationary form only: fuse k_embed_front into this launch;
                                                                     an error if the launch cannot carry it */,
                               int* status = nullptr /* device status word: DC_STATUS_F16_SAT is OR-ed in when a tile leaves the fp16 range */);
// pp != nullptr (non-split formats): the FiLM GEMM builds its operand SiLU(temb[t_clip] + pp) itself and s_hi is not read
// wgr: workgroup-level partial records, combined by the consuming layer kernel itself (non-split formats and T >= 256 only;
// no dc_launch_attn_combine between the layers then)
hipError_t dc_launch_embed_front(hipStream_t st, int fmt, bool split, bool wgr, const DcModel* dm, const float* x, float* hbuf, float* recs,
                                 const int* length, int M, int T, int G, int B,
                                 unsigned long long* clk /* diagnostic stamps (8 slots) or nullptr */,
                                 bool narrow /* wgr, non-split: 4-wave workgroups = 128-token units (small batches) */,
                                 int Tx /* frames per clip of x (<= the clip stride T) */,
                                 int upc /* wgr: workgroups per clip (clip-aligned units, WgMap in dc_dev.h), grid = B * upc; else 0 */);
// test hook: front half of layer l0 from the residual stream as it stands in hbuf (per-group records)
hipError_t dc_launch_front_from_h(hipStream_t st, int fmt, bool split, const DcModel* dm, float* hbuf, float* recs, const int* length,
                                  int M, int T, int G, int B, int l0);
hipError_t dc_launch_layer(hipStream_t st, int fmt, bool split, bool wgr, const DcModel* dm, int l, float* hbuf, const void* E, int NT,
                           const void* a_sa, const void* a_ca, float* recs, const int* length, const float* xin,
                           float* xout, int out_mode, const float* coef_cur, const int* snap_cur, float* snaps,
                           int M, int T, int G, int B, int dbg, unsigned long long* stamps, size_t rec_stride,
                           const int* iter_base /* captured loop: coef_cur / snap_cur = this step's slots of the per-iteration tables,
                                                   indexed by *iter_base; else nullptr (scalars prepared by k_begin_step) */,
                           bool narrow /* wgr, non-split, dbg == 0: 4-wave workgroups; recs / rec_stride then count 128-token units */,
                           int Tx /* frames per clip of xin / xout / snaps */, int upc /* as dc_launch_embed_front */,
                           const DcUpdate& upd /* options of the fused DDIM update + the status word (dc_common.h) */,
                           bool g1 = false /* the FiLM scale tiles hold G' (film_affine in dc_dev.h; plain-operand production forms only) */);
// The same layer for SMALL batches on 16-token waves (dc_layer16.hip): non-split formats, clip-aligned 64-token units (grid = B * upc,
// upc = ceil(T / 64), T = clip stride, a multiple of 32), one unit record per workgroup.  a_ca16 = the cross-attention fragments in
// that kernel's form (dc_launch_cond_af16, once per conditioning).  nu_in / stride_in: unit records per clip and floats per unit of
// the records this layer combines (layer 0: k_embed_front's narrow 128-token units, 2 * DC_REC_FLOATS apart; later layers: upc
// units DC_REC_FLOATS apart).  At most dc_layer16_max_units() records per clip.
hipError_t dc_launch_layer16(hipStream_t st, int fmt, const DcModel* dm, int l, float* hbuf, const void* E, int NT, const void* a_ca16,
                             float* recs, const int* length, const float* xin, float* xout, int out_mode, const float* coef_cur,
                             const int* snap_cur, float* snaps, int M, int T, int B, int upc, size_t rec_stride, int nu_in, size_t stride_in,
                             const int* iter_base, int Tx, const DcUpdate& upd,
                             unsigned long long* gran /* [B][1024] granules: the clip's workgroups share the combine inside the launch (nullptr: each alone) */,
                             unsigned tag_base /* the launch's tag = tag_base + 16 * (*iter_base) + l + 1: must differ between consecutive launches */,
                             bool g1 = false /* the FiLM scale tiles hold G' */);
hipError_t dc_launch_cond_af16(hipStream_t st, int fmt, const void* a_ca, void* a_ca16, int n_matrices);
int dc_layer16_max_units(void);
hipError_t dc_launch_advance_iter(hipStream_t st, int* iter, int k);
hipError_t dc_launch_set_ptr(hipStream_t st, const float** slot /* 24 bytes: base, seed, first element */, const float* p, unsigned long long seed,
                             unsigned long long first);
// N(0, 1) draws of one DDIM iteration (Philox keyed by *seed_slot when given, else seed; iteration = step + *iter_base, else snap_cur[1],
// else step) into z[0..n); z[0] is element `first` (seed_slot[1] when a slot is given) of the whole batch's draw
hipError_t dc_launch_step_noise(hipStream_t st, float* z, size_t n, unsigned long long seed, const unsigned long long* seed_slot, const int* iter_base,
                                int step, const int* snap_cur, unsigned long long first);
// diagnosis: OR DC_STATUS_F16_SAT into *status when the fp16 buffer e holds an inf / nan
hipError_t dc_launch_scan_f16(hipStream_t st, const void* e, size_t bytes, int* status);
// rec_stride: floats between the two alternating unit-record buffers (0 = single buffer, non-wgr)

// ---- no_eff variant (full T x T attention).  KT = key tiles per clip array.  split: the 128-wide GEMMs on split operands (`dm` is then
// the model record with the split stage images); the attention's own operands stay plain 16-bit.
hipError_t dc_launch_ca_kv(hipStream_t st, int fmt, const DcModel* dm, const void* nh_hi, const void* nh_lo, void* kv_ca,
                           int M, int T, int G, int B, int KT, int L);
hipError_t dc_launch_embed_front_full(hipStream_t st, int fmt, bool split, const DcModel* dm, const float* x, float* hbuf, void* kv_next,
                                      int M, int T, int B, int KT);
hipError_t dc_launch_layer_full(hipStream_t st, int fmt, bool split, const DcModel* dm, int l, float* hbuf, const void* E, int NT,
                                const void* kv_cur, void* kv_next, const void* kv_ca, const int* length, const float* xin,
                                float* xout, int out_mode, const float* coef_cur, const int* snap_cur, float* snaps, int M,
                                int T, int B, int KT, int stop_after, const DcUpdate& upd);

// diagnostic builds (-DDC_DIAG_FULL_MOVES): visits / moves of the no_eff key loop's reference point; hipErrorNotSupported otherwise
hipError_t dc_full_moves_read(unsigned long long* out /* [2] */, bool reset);

// Savitzky-Golay smoothing along time of [B][T][P] fp32 (coef: hat matrix [win][win]); y != x
hipError_t dc_launch_savgol(hipStream_t st, const float* x, float* y, const float* coef, int B, int T, int P, int win);
