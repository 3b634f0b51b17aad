// dc_music.h - internal interface of the MusicEncoder / encode_music kernels (dc_music.hip).
// Reference: Diffusion_Stage/models/transformer.py:289-340 (Conv2dResLayer, MusicEncoder), :447-459 (encode_music).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

struct dc_music;   // device-resident folded weights + ping-pong activation planes

// Builds the encoder from reference state_dict entries (`music_encoder.*`, `proj.*`); returns nullptr and sets
// *err when a key is missing or has the wrong size.  BatchNorm (eval mode, running statistics) is folded into the
// convolution in front of it.
dc_music* dc_music_build(const std::map<std::string, std::vector<float>>& params, int music_dim, std::string* err);
void dc_music_destroy(dc_music* m);
// names/sizes of the entries dc_music_build consumes
std::vector<std::pair<std::string, size_t>> dc_music_required(int music_dim);

// mel [B][Tm][128] fp32 (device) -> xf_out [B][T][64], xf_proj [B][T][64] fp32 (device), T = (Tm - 1) / 3 + 1.
// Work is enqueued on `st`; clips are processed in chunks so the activation planes stay bounded.
hipError_t dc_music_encode(dc_music* m, const float* d_mel, int B, int Tm, float* d_xf_proj, float* d_xf_out, hipStream_t st,
                           std::string* err);
// plane format of this encoder's activations: 0 = two bf16 planes (hi + lo, three MFMAs per product: ~6e-6 at the output), 1 = one
// fp16 plane (one MFMA per product, half the bytes: ~4e-4 at the output, 1.3e-4 of x0 after DDIM-50).  DC_ME_PREC=f16|split overrides.
void dc_music_set_format(dc_music* m, int single_fp16);
int dc_music_format(const dc_music* m);
int dc_music_frames(int Tm);
long long dc_music_workspace_bytes(const dc_music* m);
