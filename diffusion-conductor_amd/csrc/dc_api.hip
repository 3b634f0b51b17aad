// dc_api.hip - host side of libdc_ddim.so: parameter packing, workspace, step enqueue,
// hipGraph capture/replay and the C ABI declared in include/dc_ddim.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dc_ddim.h"
#include "dc_common.h"
#include "dc_launch.h"
#include "dc_music.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(DC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- bf16 helpers (round to nearest even; inputs are finite weights) -------------------
inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline uint16_t f2h(float f) {   // fp32 -> fp16 bits, round to nearest even (compiler's conversion)
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

inline int tile_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Weight image [kt][ot][s][64][8] (kt-major).  "chained" k order: frag (ot, kt, s), lane (i = l&31, hh = l>>5), element j
//   = W[32ot + i][32kt + 16s + 8(j>>2) + 4hh + (j&3)]
// i.e. the k order in which an accumulator tile, converted in registers, presents its rows.
// "natural" k order: frag (ot, ks): element j = W[32ot + i][16ks + 8hh + j].
void pack_weight(const float* w, int n_out, int k_in, bool chained, uint16_t* hi, uint16_t* lo, bool f16 = false) {
    const int OT = cdiv(n_out, 32), KT = cdiv(k_in, 32);
    for (int ot = 0; ot < OT; ++ot)
        for (int kt = 0; kt < KT; ++kt)
            for (int s = 0; s < 2; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int i = l & 31, hh = l >> 5;
                        const int row = 32 * ot + i;
                        const int col = chained ? 32 * kt + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)
                                                : 32 * kt + 16 * s + 8 * hh + j;
                        const float v = (row < n_out && col < k_in) ? w[(size_t)row * k_in + col] : 0.f;
                        const size_t o = chained ? ((((size_t)kt * OT + ot) * 2 + s) * 64 + l) * 8 + j     // [kt][ot][s]: k-outer sweeps
                                                 : ((((size_t)ot * KT + kt) * 2 + s) * 64 + l) * 8 + j;    // [ot][ks]: streamed per tile
                        const uint16_t h = f16 ? f2h(v) : f2bf(v);
                        hi[o] = h;
                        lo[o] = f16 ? f2h(v - h2f(h)) : f2bf(v - bf2f(h));
                    }
}
size_t packed_elems(int n_out, int k_in) { return (size_t)cdiv(n_out, 32) * cdiv(k_in, 32) * 2 * 64 * 8; }
// v_mfma_f32_16x16x32 operand image of the 16-token layer kernel (DcLayer16 in dc_common.h): [m][rb][64][8]
void pack_weight16(const float* w, int n_out, int k_in, uint16_t* hi, uint16_t* lo, bool f16) {
    const int RB = cdiv(n_out, 16), KM = cdiv(k_in, 32);
    for (int m = 0; m < KM; ++m)
        for (int rb = 0; rb < RB; ++rb)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int row = 16 * rb + (l & 15), col = 32 * m + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
                    const float v = (row < n_out && col < k_in) ? w[(size_t)row * k_in + col] : 0.f;
                    const size_t o = (((size_t)m * RB + rb) * 64 + l) * 8 + j;
                    const uint16_t h = f16 ? f2h(v) : f2bf(v);
                    hi[o] = h;
                    if (lo) lo[o] = f16 ? f2h(v - h2f(h)) : f2bf(v - bf2f(h));
                }
}
size_t packed_elems16(int n_out, int k_in) { return (size_t)cdiv(n_out, 16) * cdiv(k_in, 32) * 64 * 8; }

// per-feature vector in FT register order: out[(t*2+hh)*16 + r] = v[32t + tile_row(r,hh)]
void pack_ftvec(const float* v, int n, int NT, float* out) {
    for (int t = 0; t < NT; ++t)
        for (int hh = 0; hh < 2; ++hh)
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * t + tile_row(r, hh);
                out[(t * 2 + hh) * 16 + r] = f < n ? v[f] : 0.f;
            }
}

// ---- device arena builder --------------------------------------------------------------
struct Arena {
    std::vector<uint8_t> host;
    size_t add(const void* p, size_t bytes) {
        size_t off = (host.size() + 255) & ~(size_t)255;
        host.resize(off + bytes);
        memcpy(host.data() + off, p, bytes);
        return off;
    }
};

enum KernelId { K_BEGIN = 0, K_SILU, K_FILM, K_EMBED, K_COMBINE, K_LAYER, K_NOISE, K_COUNT };
const char* kKernelNames[K_COUNT] = {"k_begin_step", "k_silu_emb", "k_film_gemm", "k_embed_front", "k_attn_combine", "k_layer", "k_step_noise"};

struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;     // pairs
    std::vector<int> ids;
};

}  // namespace

struct dc_sampler {
    dc_config cfg{};
    // operand formats (0 = bf16, 1 = f16) and split flags of the 128-wide GEMMs and of the FiLM GEMM
    int small_fmt = 0, film_fmt = 0;
    bool split_small = false, split_film = false;
    std::map<std::string, std::vector<float>> params;
    bool finalized = false;

    uint8_t* d_arena = nullptr;
    size_t arena_bytes = 0;
    DcModel* d_model = nullptr;
    DcModel h_model{};
    int NT = 0;   // FiLM feature tiles = 3 * L * 8
    int gran = 32;   // tokens per partial-record unit: 32 (per group) or waves-per-workgroup * 32 (T permitting)

    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;

    // workspace (capacity-tracked)
    int B = 0, T = 0, M = 0, G = 0;          // T: clip stride of the token space (the caller's frames per clip, padded: below)
    int Tx = 0;                              // the caller's frames per clip: x, xf, snapshots are [B][Tx][..]
    size_t cap_G = 0, cap_B = 0, cap_MP = 0, cap_steps = 0, cap_snap = 0, cap_kv = 0;
    size_t cap_rec_floats = 0;      // floats behind d_recs
    std::vector<int> len_dev;       // the clip lengths d_length holds (dc_sampler_set_conditioning), valid while len_dev_ptr == d_length
    const int* len_dev_ptr = nullptr;
    int* d_length = nullptr;
    float* d_pp = nullptr;
    void *d_s_hi = nullptr, *d_s_lo = nullptr;
    void* d_E = nullptr;
    float* d_h = nullptr;
    float* d_recs = nullptr;
    void *d_a_sa = nullptr, *d_a_ca = nullptr;
    void* d_a_ca16 = nullptr;             // cross-attention fragments in the 16-token layer kernel's form (small batches)
    unsigned long long* d_gran = nullptr; // small batches: granules of the combine the clip's workgroups share inside a launch, [B][1024] (dc_layer16.hip)
    unsigned l16_seq = 0;                 // eager launches of k_layer16: tag sequence (tags must differ between consecutive launches)
    bool l16_own = false;                 // every workgroup of k_layer16 combines alone (no in-launch exchange): dc_sampler_set_combine_exchange(s, 0),
                                          // or latched by dc_sampler_status after a DC_STATUS_TIMEOUT (the GPU is shared: co-residency cannot be assumed)
    void *d_kv_sa[2] = {nullptr, nullptr}, *d_kv_ca = nullptr;   // no_eff: key-tile arrays (dc_kernels.hip, full attention)
    int KT = 0;                                                   // key tiles per clip array
    float* d_x = nullptr;
    float* d_snaps = nullptr;
    // conditioning temporaries
    float* d_recs_ca = nullptr;
    void *d_nh_hi = nullptr, *d_nh_lo = nullptr;
    // step state
    unsigned long long* d_stamps = nullptr;
    float* d_film_rate = nullptr;  // FiLM GEMM: per-workgroup speeds measured by the previous launches, two buffers of 1024 (ping-pong)
    int film_rate_parity = 0;
    bool graph_folded = false;     // the captured steps look their timestep up through *d_iter (no k_begin_step launches)
    int num_cu = 0;
    int *d_iter = nullptr, *d_t_clip = nullptr, *d_snap_cur = nullptr, *d_t_of_iter = nullptr, *d_snap_of_iter = nullptr;
    float *d_coef_cur = nullptr, *d_coef_of_t = nullptr, *d_coef_of_iter = nullptr;   // DDIM scalars by timestep / by iteration
    bool cond_set = false;
    int64_t ws_bytes = 0;
    std::map<void*, size_t> alloc_bytes;   // live workspace allocations (ws_bytes = their sum)
    // host copies of the per-iteration tables currently on the device (tables_S == 0: none)
    int tables_S = 0;
    std::vector<int> tab_t, tab_snap;
    std::vector<float> tab_coef, tab_coef_iter;

    // graph cache: one per (B, T, steps_per_graph, launch form, update options)
    hipGraphExec_t graph = nullptr;
    int graph_B = 0, graph_T = 0, graph_Tx = 0, graph_K = 0;
    unsigned long long graph_form = 0;     // form_key() of the captured launches
    // ... and up to three more for other batch shapes (a dataset's last, smaller batch; a service whose batch size varies): a shape
    // seen before replays its graph instead of paying a capture (tens of ms) every time the shape changes.  The workspace's
    // addresses are baked into every captured graph, so whatever re-allocates a buffer drops them all (drop_graph).
    struct GraphSlot {
        hipGraphExec_t exec;
        int B, T, Tx, K;
        unsigned long long form;
        bool folded;
    };
    std::vector<GraphSlot> graph_park;
    // DDIM update options of the loop being enqueued (dc_sampler_ddim_loop_ex) and the device status word
    int upd_flags = 0;              // DC_UPD_* of the loop being enqueued (incl. the internal NOISY / ZSTEP bits)
    const float** d_zslot = nullptr;    // device slot holding the base address of the per-iteration noise (DcUpdate::zslot)
    float* d_zstep = nullptr;       // library-generated draws of one iteration [B][Tx][P] (dc_sampler_set_step_noise_seed)
    size_t cap_zstep = 0;
    unsigned long long noise_seed = 0, noise_first = 0;      // Philox key; index of this sampler's first element in the whole batch's draw
    bool noise_seed_set = false;                             // a seed is consumed by the loop that uses it
    int* d_status = nullptr;
    // Savitzky-Golay smoothing applied by the loop's final write (dc_sampler_set_smoothing; window 0 = off)
    int smooth_window = 0, smooth_order = 0, smooth_table_window = 0;     // (table_window: the hat matrix d_smooth_coef holds)
    float* d_smooth_coef = nullptr;

    DcModel h_model_split{};     // fp16 precision: the same model with the layer stage images in their split form ([hi][lo][consts]): the loop's precise tail
    DcModel* d_model_split = nullptr;
    dc_music* music = nullptr;   // MusicEncoder (built when its parameters were supplied)
    int me_format = -1;          // dc_sampler_set_encoder_format (-1: by precision)
    int clip_aligned = -1;         // dc_sampler_set_clip_aligned: 1 clip-aligned units in the wide form too, 0 flat units, -1 the library's rule
    bool precise_forward = false;  // dc_sampler_set_precise_forward: dc_sampler_denoise on split operands (the precise tail's evaluation form)
    int tail_split = -1;         // dc_sampler_set_precise_tail: the loop's last evaluations with split operands (-1: by precision - fp16 1, bf16 8)
    bool host_only = false;      // -DDC_HOST_SANITIZE builds without a device: the host half only (tests/test_host_sanitize.py)

    Prof prof;
    int dbg_layers = -1, dbg_stage = 0;   // test hooks (dc_sampler_debug_denoise)
    int dbg_first = -1;                   // test hook (dc_sampler_debug_layer): start at this layer from the residual stream in d_h
    bool embedded_by_prev = false;        // enqueue_step: the step just enqueued also did the next step's front work (DC_UPD_EMBED_NEXT)
    bool diag_film_done = false;          // DC_DIAG_SKIP_FILM (diagnostic): the FiLM GEMM has been launched once on this sampler
};

namespace {

// (Re)allocates one workspace buffer.  The caller's capacity field is only raised after every buffer of its group was
// allocated (ensure_workspace resets it to 0 first), so a failed hipMalloc leaves "no capacity", never a stale one.
template <class P>
int dev_alloc(dc_sampler* s, P*& p, size_t bytes) {
    if (p) {
        auto it = s->alloc_bytes.find((void*)p);
        if (it != s->alloc_bytes.end()) {
            s->ws_bytes -= (int64_t)it->second;
            s->alloc_bytes.erase(it);
        }
        void* old = (void*)p;
        p = nullptr;
        HIP_TRY(hipFree(old));
    }
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, bytes));
    p = (P*)q;
    s->alloc_bytes[q] = bytes;
    s->ws_bytes += (int64_t)bytes;
    return DC_OK;
}

void drop_graph(dc_sampler* s) {
    for (auto& g : s->graph_park) hipGraphExecDestroy(g.exec);
    s->graph_park.clear();
    if (s->graph) {
        hipGraphExecDestroy(s->graph);
        s->graph = nullptr;
    }
}

const std::vector<float>* find(dc_sampler* s, const std::string& n) {
    auto it = s->params.find(n);
    return it == s->params.end() ? nullptr : &it->second;
}

struct ParamReq {
    std::string name;
    size_t numel;
};

std::vector<ParamReq> required_params(const dc_config& c) {
    std::vector<ParamReq> r;
    const size_t D = c.latent_dim, E = 4 * D, L = 512, F = c.ff_size, P = c.input_feats;
    r.push_back({"sequence_embedding", (size_t)c.num_frames * D});
    r.push_back({"linear.weight", L * 64});
    r.push_back({"linear.bias", L});
    r.push_back({"joint_embed.weight", D * P});
    r.push_back({"joint_embed.bias", D});
    r.push_back({"time_embed.0.weight", E * D});
    r.push_back({"time_embed.0.bias", E});
    r.push_back({"time_embed.2.weight", E * E});
    r.push_back({"time_embed.2.bias", E});
    auto styl = [&](const std::string& p) {
        r.push_back({p + ".emb_layers.1.weight", 2 * D * E});
        r.push_back({p + ".emb_layers.1.bias", 2 * D});
        r.push_back({p + ".norm.weight", D});
        r.push_back({p + ".norm.bias", D});
        r.push_back({p + ".out_layers.2.weight", D * D});
        r.push_back({p + ".out_layers.2.bias", D});
    };
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = "temporal_decoder_blocks." + std::to_string(i);
        r.push_back({p + ".sa_block.norm.weight", D});
        r.push_back({p + ".sa_block.norm.bias", D});
        for (const char* n : {"query", "key", "value"}) {
            r.push_back({p + ".sa_block." + n + ".weight", D * D});
            r.push_back({p + ".sa_block." + n + ".bias", D});
        }
        styl(p + ".sa_block.proj_out");
        r.push_back({p + ".ca_block.norm.weight", D});
        r.push_back({p + ".ca_block.norm.bias", D});
        r.push_back({p + ".ca_block.text_norm.weight", L});
        r.push_back({p + ".ca_block.text_norm.bias", L});
        r.push_back({p + ".ca_block.query.weight", D * D});
        r.push_back({p + ".ca_block.query.bias", D});
        for (const char* n : {"key", "value"}) {
            r.push_back({p + ".ca_block." + n + ".weight", D * L});
            r.push_back({p + ".ca_block." + n + ".bias", D});
        }
        styl(p + ".ca_block.proj_out");
        r.push_back({p + ".ffn.linear1.weight", F * D});
        r.push_back({p + ".ffn.linear1.bias", F});
        r.push_back({p + ".ffn.linear2.weight", D * F});
        r.push_back({p + ".ffn.linear2.bias", D});
        styl(p + ".ffn.proj_out");
    }
    r.push_back({"out.weight", P * D});
    r.push_back({"out.bias", P});
    return r;
}

// MusicEncoder / proj entries are optional as a group: a sampler fed x_proj/x_out from elsewhere never needs them.
// Entries of the group the kernels do not consume (num_batches_tracked counters) are accepted and dropped.
bool music_param(const std::string& n) {
    return n.rfind("music_encoder.", 0) == 0 || n == "proj.weight" || n == "proj.bias";
}

struct Offsets {   // arena offsets mirrored into DcModel after upload
    std::vector<std::pair<const void**, size_t>> fix;
};

int build_model(dc_sampler* s) {
    const dc_config& c = s->cfg;
    const int D = DC_D, L = c.num_layers, P = c.input_feats;
    Arena A;
    Offsets O;
    DcModel& m = s->h_model;
    memset(&m, 0, sizeof m);
    auto P_ = [&](const std::string& n) -> const float* { return find(s, n)->data(); };

    const bool sf16 = s->small_fmt == 1;
    auto add_packed = [&](const bf16x8** dst, const float* w, int n_out, int k_in, bool chained, bool f16) {
        const size_t ne = packed_elems(n_out, k_in);
        std::vector<uint16_t> buf(2 * ne);
        pack_weight(w, n_out, k_in, chained, buf.data(), buf.data() + ne, f16);
        O.fix.push_back({(const void**)dst, A.add(buf.data(), buf.size() * 2)});
    };
    auto add_ft = [&](const float** dst, const float* v, int n, int NT) {
        std::vector<float> buf((size_t)NT * 32);
        pack_ftvec(v, n, NT, buf.data());
        O.fix.push_back({(const void**)dst, A.add(buf.data(), buf.size() * 4)});
    };
    auto add_raw = [&](const float** dst, const float* v, size_t n) {
        O.fix.push_back({(const void**)dst, A.add(v, n * 4)});
    };
    // stage image: [hi frags][lo frags if `with_lo`][1 KiB of fp32 constants if `consts`] (dc_common.h)
    // fp16 precision: every layer stage image is also kept in its split form (the fp16 lo halves exist anyway): the loop's last
    // evaluations can then run on split operands (dc_sampler_set_precise_tail) through h_model_split, a copy of the model record
    // whose image pointers are these twins
    const bool want_twins = (c.precision == DC_PREC_FP16 || c.precision == DC_PREC_BF16) && !s->split_small;
    std::vector<std::pair<size_t, size_t>> twins;          // (offset of the pointer inside DcModel, arena offset of the split image)
    auto add_image = [&](const bf16x8** dst, const float* w, int n_out, int k_in, bool with_lo, const float* consts,
                         size_t n_consts) {
        const size_t ne = packed_elems(n_out, k_in);
        std::vector<uint16_t> hi(ne), lo(ne);
        pack_weight(w, n_out, k_in, true, hi.data(), lo.data(), sf16);
        auto blob_of = [&](bool lo_too) {
            std::vector<uint8_t> blob(ne * 2 * (lo_too ? 2 : 1) + (consts ? 1024 : 0), 0);
            memcpy(blob.data(), hi.data(), ne * 2);
            if (lo_too) memcpy(blob.data() + ne * 2, lo.data(), ne * 2);
            if (consts) memcpy(blob.data() + ne * 2 * (lo_too ? 2 : 1), consts, n_consts * 4);
            return blob;
        };
        const std::vector<uint8_t> blob = blob_of(with_lo);
        O.fix.push_back({(const void**)dst, A.add(blob.data(), blob.size())});
        const size_t off = (size_t)((const char*)dst - (const char*)&m);
        if (want_twins && !with_lo && off < sizeof(DcModel)) {
            const std::vector<uint8_t> b2 = blob_of(true);
            twins.push_back({off, A.add(b2.data(), b2.size())});
        }
    };
    auto ftvec = [&](const float* v, int n, int NT_) {
        std::vector<float> buf((size_t)NT_ * 32);
        pack_ftvec(v, n, NT_, buf.data());
        return buf;
    };
    const bool ssp = s->split_small;
    // 16-token layer kernel (small batches; non-split formats, linear attention): up to two matrices + constants per stage image
    const bool want16 = !ssp && !c.no_eff;
    auto add_image16 = [&](const bf16x8** dst, const float* wa, int na_out, int ka, const float* wb, int nb_out, int kb, bool with_lo,
                           const std::vector<float>& consts) {
        if (!want16) return;
        const size_t ea = packed_elems16(na_out, ka), eb = wb ? packed_elems16(nb_out, kb) : 0;
        std::vector<uint16_t> hi(ea + eb), lo(ea + eb);
        pack_weight16(wa, na_out, ka, hi.data(), lo.data(), sf16);
        if (wb) pack_weight16(wb, nb_out, kb, hi.data() + ea, lo.data() + ea, sf16);
        std::vector<uint8_t> blob((ea + eb) * 2 * (with_lo ? 2 : 1) + 1024, 0);
        memcpy(blob.data(), hi.data(), (ea + eb) * 2);
        if (with_lo) memcpy(blob.data() + (ea + eb) * 2, lo.data(), (ea + eb) * 2);
        memcpy(blob.data() + (ea + eb) * 2 * (with_lo ? 2 : 1), consts.data(), consts.size() * 4);
        O.fix.push_back({(const void**)dst, A.add(blob.data(), blob.size())});
    };
    auto vec = [](const float* p, size_t n) { return std::vector<float>(p, p + n); };
    auto add_styl = [&](const bf16x8** dst, const std::string& p) -> std::vector<float> {
        const std::vector<float> bo = ftvec(P_(p + ".out_layers.2.bias"), D, 4);
        // the kernels hand over log2(e) * SiLU(.) (silu_l2_pair in dc_kernels.hip): ln 2 goes into the weights
        const float* w = P_(p + ".out_layers.2.weight");
        std::vector<float> ws((size_t)D * D);
        for (size_t i = 0; i < ws.size(); ++i) ws[i] = (float)((double)w[i] * 0.6931471805599453);
        add_image(dst, ws.data(), D, D, ssp, bo.data(), bo.size());
        return ws;
    };
    // W' = W diag(g), c' = c + W b  (LayerNorm affine folded into the projection that consumes it)
    // `scale` additionally multiplies the whole projection: log2(e) for the query / key projections, whose
    // outputs only ever feed exp() (softmax), so the kernels can use the native exp2.
    auto fold_ln = [&](const float* w, const float* c, const float* g, const float* b, int n_out, int k,
                       std::vector<float>& wf, std::vector<float>& cf, double scale = 1.0) {
        wf.resize((size_t)n_out * k);
        cf.resize(n_out);
        for (int o = 0; o < n_out; ++o) {
            double acc = c[o];
            for (int i = 0; i < k; ++i) {
                wf[(size_t)o * k + i] = (float)((double)w[(size_t)o * k + i] * g[i] * scale);
                acc += (double)w[(size_t)o * k + i] * b[i];
            }
            cf[o] = (float)(acc * scale);
        }
    };
    // `linear` with the mean over its 512 outputs taken off (the LayerNorm in front of the cross-attention K / V projections sees
    // linear(x) - mean = Wc x + bc), and that LayerNorm's variance as a quadratic form of the 64 inputs
    std::vector<double> lin_wc((size_t)512 * 64), lin_bc(512);
    std::vector<float> lin_gram(64 * 64 + 64 + 1);
    {
        const float* w = P_("linear.weight");   // [512][64]
        const float* b = P_("linear.bias");
        double bm = 0.0;
        for (int k = 0; k < 512; ++k) bm += b[k];
        bm /= 512.0;
        for (int k = 0; k < 512; ++k) lin_bc[k] = (double)b[k] - bm;
        for (int i = 0; i < 64; ++i) {
            double wm = 0.0;
            for (int k = 0; k < 512; ++k) wm += w[(size_t)k * 64 + i];
            wm /= 512.0;
            for (int k = 0; k < 512; ++k) lin_wc[(size_t)k * 64 + i] = (double)w[(size_t)k * 64 + i] - wm;
        }
        for (int i = 0; i < 64; ++i) {
            for (int j = 0; j < 64; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 512; ++k) acc += lin_wc[(size_t)k * 64 + i] * lin_wc[(size_t)k * 64 + j];
                lin_gram[i * 64 + j] = (float)(acc / 512.0);
            }
            double acc = 0.0;
            for (int k = 0; k < 512; ++k) acc += lin_wc[(size_t)k * 64 + i] * lin_bc[k];
            lin_gram[64 * 64 + i] = (float)(acc / 512.0);
        }
        double cc = 0.0;
        for (int k = 0; k < 512; ++k) cc += lin_bc[k] * lin_bc[k];
        lin_gram[64 * 64 + 64] = (float)(cc / 512.0);
    }
    const double LOG2E = 1.4426950408889634;
    // softmax inputs: the linear-attention kernels use exp2 on log2(e)-scaled queries/keys; the full-attention
    // (no_eff) kernels keep keys unscaled and fold log2(e) / sqrt(head_dim) into the queries (scores arrive as exp2 exponents)
    const bool full = c.no_eff != 0;
    const double QS = full ? 0.25 * LOG2E : LOG2E, KS = full ? 1.0 : LOG2E;
    // FiLM: all 3L blocks stacked along the output axis -> one [3L*256][512] GEMM operand; inside a block the
    // 32-row tiles are interleaved (scale0, shift0, scale1, shift1, ...) so one wave holds matching pairs
    const int NT = 3 * L * DC_FILM_TILES_PER_BLOCK;
    s->NT = NT;
    std::vector<float> film_w((size_t)NT * 32 * DC_E), film_b((size_t)NT * 32), film_b_g1((size_t)NT * 32);
    for (int i = 0; i < L; ++i) {
        const std::string p = "temporal_decoder_blocks." + std::to_string(i);
        DcLayer& y = m.layer[i];
        DcLayer16& y16 = m.l16[i];
        std::vector<float> wf, cf;
        const float* sg = P_(p + ".sa_block.norm.weight");
        const float* sb = P_(p + ".sa_block.norm.bias");
        fold_ln(P_(p + ".sa_block.query.weight"), P_(p + ".sa_block.query.bias"), sg, sb, D, D, wf, cf, QS);
        {
            const std::vector<float> c = ftvec(cf.data(), D, 4);
            add_image(&y.img_sa_q, wf.data(), D, D, ssp, c.data(), c.size());
            add_image16(&y16.sa_q, wf.data(), D, D, nullptr, 0, 0, false, cf);
        }
        fold_ln(P_(p + ".sa_block.key.weight"), P_(p + ".sa_block.key.bias"), sg, sb, D, D, wf, cf, KS);
        add_image(&y.img_sa_k, wf.data(), D, D, ssp, cf.data(), cf.size());          // plain bias[128]
        add_image16(&y16.sa_k, wf.data(), D, D, nullptr, 0, 0, false, cf);
        fold_ln(P_(p + ".sa_block.value.weight"), P_(p + ".sa_block.value.bias"), sg, sb, D, D, wf, cf);
        add_image(&y.img_sa_v, wf.data(), D, D, ssp, cf.data(), cf.size());
        add_image16(&y16.sa_v, wf.data(), D, D, nullptr, 0, 0, false, cf);
        {
            const std::vector<float> ws = add_styl(&y.img_sa_o, p + ".sa_block.proj_out");
            add_image16(&y16.sa_o, ws.data(), D, D, nullptr, 0, 0, false, vec(P_(p + ".sa_block.proj_out.out_layers.2.bias"), D));
        }
        fold_ln(P_(p + ".ca_block.query.weight"), P_(p + ".ca_block.query.bias"), P_(p + ".ca_block.norm.weight"),
                P_(p + ".ca_block.norm.bias"), D, D, wf, cf, QS);
        {
            const std::vector<float> c = ftvec(cf.data(), D, 4);
            add_image(&y.img_ca_q, wf.data(), D, D, ssp, c.data(), c.size());
            add_image16(&y16.ca_q, wf.data(), D, D, nullptr, 0, 0, false, cf);
        }
        // fold text_norm's affine (transformer.py:149,153) into the K/V projections:
        //   W (g*n + b) + c = (W*g) n + (W b + c)
        {
            const float* g = P_(p + ".ca_block.text_norm.weight");
            const float* bt = P_(p + ".ca_block.text_norm.bias");
            for (int kv = 0; kv < 2; ++kv) {
                const std::string nm = p + ".ca_block." + (kv ? "value" : "key");
                const float* w = P_(nm + ".weight");
                const float* bb = P_(nm + ".bias");
                std::vector<float> wf((size_t)D * DC_E), bf(D);
                const double sc = kv ? 1.0 : KS;         // keys feed exp2 in the partial records
                for (int o = 0; o < D; ++o) {
                    double acc = bb[o];
                    for (int k = 0; k < DC_E; ++k) {
                        wf[(size_t)o * DC_E + k] = (float)((double)w[(size_t)o * DC_E + k] * g[k] * sc);
                        acc += (double)w[(size_t)o * DC_E + k] * bt[k];
                    }
                    bf[o] = (float)(acc * sc);
                }
                add_packed(kv ? &y.ca_wv : &y.ca_wk, wf.data(), D, DC_E, false, false);   // conditioning pre-pass is always split-bf16
                add_raw(kv ? &y.ca_bv : &y.ca_bk, bf.data(), D);
                // The same projection composed with `linear` (transformer.py:479-480; 64 -> 512, shared by all layers): with y = W x + b,
                // n-hat = (y - mean(y)) rstd = rstd (Wc x + bc), Wc / bc = W / b with their mean over the 512 outputs taken off, so
                //   W' n-hat + b' = rstd (A x + d) + b',   A = W' Wc [128][64],  d = W' bc
                // - an eighth of the pre-pass GEMM's products (k_cond_ca_partials64), and no [tokens][512] image in between.
                std::vector<float> af((size_t)D * 64), df(D);
                for (int o = 0; o < D; ++o) {
                    double dacc = 0.0;
                    for (int k = 0; k < DC_E; ++k) dacc += (double)wf[(size_t)o * DC_E + k] * lin_bc[k];
                    df[o] = (float)dacc;
                    for (int i = 0; i < 64; ++i) {
                        double acc = 0.0;
                        for (int k = 0; k < DC_E; ++k) acc += (double)wf[(size_t)o * DC_E + k] * lin_wc[(size_t)k * 64 + i];
                        af[(size_t)o * 64 + i] = (float)acc;
                    }
                }
                add_packed(kv ? &y.ca_av : &y.ca_ak, af.data(), D, 64, false, false);
                add_raw(kv ? &y.ca_dv : &y.ca_dk, df.data(), D);
            }
        }
        {
            const std::vector<float> ws = add_styl(&y.img_ca_o, p + ".ca_block.proj_out");
            add_image16(&y16.ca_o, ws.data(), D, D, nullptr, 0, 0, false, vec(P_(p + ".ca_block.proj_out.out_layers.2.bias"), D));
        }
        add_image(&y.img_ffn_w1, P_(p + ".ffn.linear1.weight"), DC_F, D, ssp, nullptr, 0);
        {
            std::vector<float> c = ftvec(P_(p + ".ffn.linear1.bias"), DC_F, 2);       // 64 floats, then b2
            const std::vector<float> c2 = ftvec(P_(p + ".ffn.linear2.bias"), D, 4);
            c.insert(c.end(), c2.begin(), c2.end());
            add_image(&y.img_ffn_w2, P_(p + ".ffn.linear2.weight"), D, DC_F, ssp, c.data(), c.size());
            std::vector<float> pc = vec(P_(p + ".ffn.linear1.bias"), DC_F);
            const std::vector<float> pb2 = vec(P_(p + ".ffn.linear2.bias"), D);
            pc.insert(pc.end(), pb2.begin(), pb2.end());
            add_image16(&y16.ffn_w, P_(p + ".ffn.linear1.weight"), DC_F, D, P_(p + ".ffn.linear2.weight"), D, DC_F, false, pc);
        }
        {
            const std::vector<float> ws = add_styl(&y.img_ffn_o, p + ".ffn.proj_out");
            add_image16(&y16.ffn_o, ws.data(), D, D, nullptr, 0, 0, false, vec(P_(p + ".ffn.proj_out.out_layers.2.bias"), D));
        }
        const char* blk[3] = {".sa_block.proj_out", ".ca_block.proj_out", ".ffn.proj_out"};
        for (int j = 0; j < 3; ++j) {
            const size_t row0 = (size_t)(3 * i + j) * 256;
            const float* w = P_(p + blk[j] + ".emb_layers.1.weight");   // rows 0..127 scale, 128..255 shift
            const float* bb = P_(p + blk[j] + ".emb_layers.1.bias");
            const float* ng = P_(p + blk[j] + ".norm.weight");
            const float* nb = P_(p + blk[j] + ".norm.bias");
            // y = LN(h) (1 + scale) + shift with LN = g n + beta (transformer.py:74-78) becomes  y = n G' + H',
            //   G' = g (1 + scale), H' = beta (1 + scale) + shift, both affine in S = SiLU(emb): fold g / beta into the rows.
            // The H' tiles additionally carry log2(e): the kernels evaluate SiLU on log2(e)-scaled arguments (silu_l2_pair)
            // (tile 2t = G' - 1 of features 32t.., tile 2t+1 = H' of the same features, so one wave holds matching pairs)
            for (int t = 0; t < 4; ++t)
                for (int f = 0; f < 32; ++f) {
                    const int o = 32 * t + f;
                    const float* ws = w + (size_t)o * DC_E;
                    const float* wh = w + (size_t)(128 + o) * DC_E;
                    float* dg = &film_w[(row0 + (size_t)(2 * t) * 32 + f) * DC_E];
                    float* dh = &film_w[(row0 + (size_t)(2 * t + 1) * 32 + f) * DC_E];
                    for (int k = 0; k < DC_E; ++k) {
                        dg[k] = (float)((double)ng[o] * ws[k]);
                        dh[k] = (float)(((double)nb[o] * ws[k] + wh[k]) * LOG2E);
                    }
                    film_b[row0 + (size_t)(2 * t) * 32 + f] = (float)((double)ng[o] * (1.0 + bb[o]) - 1.0);
                    film_b[row0 + (size_t)(2 * t + 1) * 32 + f] = (float)(((double)nb[o] * (1.0 + bb[o]) + bb[128 + o]) * LOG2E);
                    film_b_g1[row0 + (size_t)(2 * t) * 32 + f] = (float)((double)ng[o] * (1.0 + bb[o]));          // (G' itself: see below)
                    film_b_g1[row0 + (size_t)(2 * t + 1) * 32 + f] = film_b[row0 + (size_t)(2 * t + 1) * 32 + f];
                }
        }
    }
    add_packed(&m.film_w, film_w.data(), NT * 32, DC_E, false, s->film_fmt == 1);
    add_ft(&m.film_b, film_b.data(), NT * 32, NT);
    // ... and with the scale tiles holding G' itself: the plain-operand layer kernels then form n-hat G' + H' in ONE mixed-precision FMA
    // instead of two (-192 vector instructions per wave and layer).  fp16 keeps 11 bits of a number near 1 there instead of 11 bits of
    // its small part - affordable where the loop's last evaluations run on split operands and G' - 1 tiles (precise tail, dc_ddim.h)
    add_ft(&m.film_b_g1, film_b_g1.data(), NT * 32, NT);
    {   // 16x16x32 operand order (dc_common.h)
        static const int pi[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};
        const bool f16 = s->film_fmt == 1;
        std::vector<uint16_t> w16((size_t)NT * 32 * DC_E);
        std::vector<float> b16((size_t)NT * 32);
        for (int ot = 0; ot < NT; ++ot)
            for (int ks = 0; ks < DC_E / 32; ++ks)
                for (int fb = 0; fb < 2; ++fb)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const float v = film_w[(size_t)(32 * ot + 16 * fb + pi[l & 15]) * DC_E + 32 * ks + 8 * (l >> 4) + j];
                            w16[((((size_t)ot * (DC_E / 32) + ks) * 2 + fb) * 64 + l) * 8 + j] = f16 ? f2h(v) : f2bf(v);
                        }
        for (int ot = 0; ot < NT; ++ot)
            for (int fb = 0; fb < 2; ++fb)
                for (int r = 0; r < 16; ++r) b16[((size_t)ot * 2 + fb) * 16 + r] = film_b[(size_t)32 * ot + 16 * fb + pi[r]];
        O.fix.push_back({(const void**)&m.film_w16, A.add(w16.data(), w16.size() * 2)});
        add_raw(&m.film_b16, b16.data(), b16.size());
        m.film_w16_tail = nullptr;
        if (want_twins && c.precision == DC_PREC_BF16) {
            // bf16 precision: the evaluations of the precise tail run the "mixed" form - split-bf16 128-wide GEMMs AND an f16 FiLM GEMM
            // (8 mantissa bits on the K = 512 operands were the tail's floor: 3.6 - 4.1e-4 with every evaluation split)
            for (size_t i = 0; i < w16.size(); ++i) w16[i] = 0;
            for (int ot = 0; ot < NT; ++ot)
                for (int ks = 0; ks < DC_E / 32; ++ks)
                    for (int fb = 0; fb < 2; ++fb)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j) {
                                const float v = film_w[(size_t)(32 * ot + 16 * fb + pi[l & 15]) * DC_E + 32 * ks + 8 * (l >> 4) + j];
                                w16[((((size_t)ot * (DC_E / 32) + ks) * 2 + fb) * 64 + l) * 8 + j] = f2h(v);
                            }
            O.fix.push_back({(const void**)&m.film_w16_tail, A.add(w16.data(), w16.size() * 2)});
        }
        for (int ot = 0; ot < NT; ++ot)
            for (int fb = 0; fb < 2; ++fb)
                for (int r = 0; r < 16; ++r) b16[((size_t)ot * 2 + fb) * 16 + r] = film_b_g1[(size_t)32 * ot + 16 * fb + pi[r]];
        add_raw(&m.film_b16_g1, b16.data(), b16.size());
    }
    {   // the two pose projections always run split: [hi][lo][bias]
        const std::vector<float> jb = ftvec(P_("joint_embed.bias"), D, 4);
        add_image(&m.img_je, P_("joint_embed.weight"), D, P, true, jb.data(), jb.size());
        const std::vector<float> ob = ftvec(P_("out.bias"), P, 1);
        add_image(&m.img_out, P_("out.weight"), P, D, true, ob.data(), ob.size());
        std::vector<float> ob16(32, 0.f);
        for (int i = 0; i < P; ++i) ob16[i] = P_("out.bias")[i];
        add_image16(&m.out16, P_("out.weight"), P, D, nullptr, 0, 0, true, ob16);
    }
    add_raw(&m.seq_emb, P_("sequence_embedding"), (size_t)c.num_frames * D);
    {
        const float* w = P_("linear.weight");   // [512][64] -> transposed [64][512]
        std::vector<float> wt((size_t)64 * 512);
        for (int k = 0; k < 512; ++k)
            for (int i = 0; i < 64; ++i) wt[(size_t)i * 512 + k] = w[(size_t)k * 64 + i];
        add_raw(&m.lin_wt, wt.data(), wt.size());
        add_raw(&m.lin_b, P_("linear.bias"), 512);
        add_raw(&m.lin_gram, lin_gram.data(), lin_gram.size());
        add_packed(&m.lin_pack, w, 512, 64, false, false);
    }
    // timestep table storage + MLP operands (transposed for coalesced reads)
    const int nt = c.max_timesteps;
    std::vector<float> zeros((size_t)nt * 512, 0.f);
    add_raw(&m.temb, zeros.data(), zeros.size());
    const float *d_freqs = nullptr, *d_w0t = nullptr, *d_b0 = nullptr, *d_w2t = nullptr, *d_b2 = nullptr;
    {
        std::vector<float> fr(64);
        for (int k = 0; k < 64; ++k) fr[k] = expf((float)(-std::log(10000.0)) * (float)k / 64.f);   // transformer.py:18-20 in fp32
        std::vector<float> w0t((size_t)128 * 512), w2t((size_t)512 * 512);
        const float* w0 = P_("time_embed.0.weight");
        const float* w2 = P_("time_embed.2.weight");
        for (int o = 0; o < 512; ++o) {
            for (int i = 0; i < 128; ++i) w0t[(size_t)i * 512 + o] = w0[(size_t)o * 128 + i];
            for (int i = 0; i < 512; ++i) w2t[(size_t)i * 512 + o] = w2[(size_t)o * 512 + i];
        }
        add_raw(&d_freqs, fr.data(), 64);
        add_raw(&d_w0t, w0t.data(), w0t.size());
        add_raw(&d_b0, P_("time_embed.0.bias"), 512);
        add_raw(&d_w2t, w2t.data(), w2t.size());
        add_raw(&d_b2, P_("time_embed.2.bias"), 512);
    }
    m.num_layers = L;
    m.input_feats = P;
    m.num_frames = c.num_frames;
    m.max_timesteps = nt;

    if (s->host_only) {          // sanitizer build without a device: the arena stays on the host, the model's pointers point into it
        free(s->d_arena);
        s->arena_bytes = A.host.size();
        s->d_arena = (uint8_t*)malloc(s->arena_bytes);
        memcpy(s->d_arena, A.host.data(), s->arena_bytes);
        for (auto& f : O.fix) *f.first = s->d_arena + f.second;
        s->h_model_split = m;
        for (auto& tw : twins) *(const void**)((char*)&s->h_model_split + tw.first) = s->d_arena + tw.second;
        return DC_OK;
    }
    if (s->d_arena) hipFree(s->d_arena);
    s->arena_bytes = A.host.size();
    HIP_TRY(hipMalloc((void**)&s->d_arena, s->arena_bytes));
    HIP_TRY(hipMemcpy(s->d_arena, A.host.data(), s->arena_bytes, hipMemcpyHostToDevice));
    for (auto& f : O.fix) *f.first = s->d_arena + f.second;
    if (!s->d_model) HIP_TRY(hipMalloc((void**)&s->d_model, sizeof(DcModel)));
    HIP_TRY(hipMemcpy(s->d_model, &m, sizeof(DcModel), hipMemcpyHostToDevice));
    s->h_model_split = m;
    for (auto& tw : twins) *(const void**)((char*)&s->h_model_split + tw.first) = s->d_arena + tw.second;
    if (!s->d_model_split) HIP_TRY(hipMalloc((void**)&s->d_model_split, sizeof(DcModel)));
    HIP_TRY(hipMemcpy(s->d_model_split, &s->h_model_split, sizeof(DcModel), hipMemcpyHostToDevice));
    HIP_TRY(dc_launch_temb_table(s->stream, d_freqs, d_w0t, d_b0, d_w2t, d_b2, (float*)m.temb, nt));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return DC_OK;
}

// Clip stride of the internal token space.  Where the workgroup-record kernels can run (non-split formats, linear attention,
// Tx >= 256) a clip may be padded to whole 32-token groups, so that no group spans two clips (the padding frames behave like
// frames past `length`).  Measured (same box): bs=32 x 1800 (+1.3 % tokens) -1.6 % per loop; bs=128 x 900
// (+3.1 %) +0.2 %; small batches, which then also run clip-aligned units (enqueue_step), -9 % at bs=4 x 1800.
int clip_stride(const dc_sampler* s, int B, int Tx) {
    // clips shorter than one 32-token group: one group per clip (a group's records name at most two clips)
    if (Tx < 32 && !s->cfg.no_eff && !getenv("DC_NO_PAD")) return 32;
    if (s->cfg.no_eff || Tx < 256 || Tx % 32 == 0 || getenv("DC_NO_PAD")) return Tx;
    const int Tp = (Tx + 31) / 32 * 32;
    if (s->split_small) return Tp;         // split formats: workgroup records exist on clip-aligned units only (one clip per workgroup)
    const bool small_batch = (long long)B * ((Tp + 127) / 128) <= s->num_cu;       // narrow, clip-aligned workgroups
    // ... unless the padding frames cost the layer launches a whole extra round of workgroups (256 tokens each, one per CU): 36 clips of
    // 1800 frames are 254 workgroups, of 1824 frames 257 - 38 vs 51 ms per loop (profiles/r05_big_batches.md)
    const auto rounds = [&](int T) { return (((long long)B * T + 255) / 256 + s->num_cu - 1) / s->num_cu; };
    if (!small_batch && rounds(Tp) > rounds(Tx)) return Tx;
    return (small_batch || (Tp - Tx) * 50 <= Tx) ? Tp : Tx;
}
int ensure_workspace(dc_sampler* s, int B, int Tx) {
    const int T = clip_stride(s, B, Tx);
    const int M = B * T, G = cdiv(M, 32), L = s->cfg.num_layers, P = s->cfg.input_feats;
    if ((size_t)G > s->cap_G) {
        drop_graph(s);
        const size_t g = (size_t)G;
        s->cap_G = 0;
        int rc;
        if ((rc = dev_alloc(s, s->d_pp, g * 32 * 64 * 8 * 4))) return rc;
        if ((rc = dev_alloc(s, s->d_s_hi, g * 32 * 64 * 16))) return rc;
        if ((rc = dev_alloc(s, s->d_s_lo, g * 32 * 64 * 16))) return rc;
        if ((rc = dev_alloc(s, s->d_E, g * s->NT * 64 * 32))) return rc;
        if ((rc = dev_alloc(s, s->d_h, g * 4 * 64 * 64))) return rc;
        HIP_TRY(hipMemset(s->d_h, 0, g * 4 * 64 * 64));     // rows past M are read (never written) by the full-attention front half
        // unit records: per-group form 2 slots per group; workgroup-record forms two alternating buffers of 2 slots per workgroup,
        // i.e. 4 * nwg records with nwg <= ceil(g / 4) (narrow flat units) or B * ceil(T / 128) (clip-aligned): both <= g for the
        // T >= 256 those forms need - sized for the larger of the two explicitly, and checked against the launch form in enqueue_step
        s->cap_rec_floats = std::max(g * 2, 4 * ((g + 3) / 4) + 8) * DC_REC_FLOATS;
        if ((rc = dev_alloc(s, s->d_recs, s->cap_rec_floats * 4))) return rc;
        if ((rc = dev_alloc(s, s->d_nh_hi, g * 32 * 64 * 16))) return rc;
        if ((rc = dev_alloc(s, s->d_nh_lo, g * 32 * 64 * 16))) return rc;
        if (!s->cfg.no_eff && (rc = dev_alloc(s, s->d_recs_ca, (size_t)L * g * 2 * DC_REC_FLOATS * 4))) return rc;
        s->cap_G = g;
    }
    if (s->cfg.no_eff) {     // key-tile arrays: [B][KT] tiles of 16 KiB; two for self-attention (layer parity), L for cross-attention
        const int KT = (T + 31) / 32 + 1;
        const size_t need = (size_t)B * KT;
        if (need > s->cap_kv) {
            drop_graph(s);
            s->cap_kv = 0;
            int rc;
            for (int i = 0; i < 2; ++i)
                if ((rc = dev_alloc(s, s->d_kv_sa[i], need * 16384))) return rc;
            if ((rc = dev_alloc(s, s->d_kv_ca, (size_t)L * need * 16384))) return rc;
            s->cap_kv = need;
        }
        s->KT = KT;
    }
    if ((size_t)B > s->cap_B) {
        drop_graph(s);
        s->cap_B = 0;
        int rc;
        if ((rc = dev_alloc(s, s->d_length, (size_t)B * 4))) return rc;
        if ((rc = dev_alloc(s, s->d_t_clip, (size_t)B * 4))) return rc;
        if ((rc = dev_alloc(s, s->d_a_sa, (size_t)B * 16 * 1024))) return rc;
        if ((rc = dev_alloc(s, s->d_a_ca, (size_t)L * B * 16 * 1024))) return rc;
        if ((rc = dev_alloc(s, s->d_a_ca16, (size_t)L * B * 8 * 1024))) return rc;
        if ((rc = dev_alloc(s, s->d_gran, (size_t)B * 1024 * 8))) return rc;
        HIP_TRY(hipMemset(s->d_gran, 0, (size_t)B * 1024 * 8));        // tag 0 is never a launch's
        s->cap_B = (size_t)B;
    }
    if ((size_t)M * P > s->cap_MP) {
        drop_graph(s);
        s->cap_MP = 0;
        int rc;
        if ((rc = dev_alloc(s, s->d_x, (size_t)M * P * 4))) return rc;
        s->cap_MP = (size_t)M * P;
    }
    if (!s->d_iter) {
        int rc;
        if ((rc = dev_alloc(s, s->d_stamps, (8 * 32 + 8 + 1024 + 1024 + 256 + 8 + 512) * 8))) return rc;
        if ((rc = dev_alloc(s, s->d_film_rate, 2 * 1024 * sizeof(float)))) return rc;
        HIP_TRY(hipMemset(s->d_film_rate, 0, 2 * 1024 * sizeof(float)));      // 0 = not measured yet: equal shares
        if ((rc = dev_alloc(s, s->d_status, 16))) return rc;
        HIP_TRY(hipMemset(s->d_status, 0, 16));
        if ((rc = dev_alloc(s, s->d_zslot, 32))) return rc;
        HIP_TRY(hipMemset(s->d_zslot, 0, 32));
        if ((rc = dev_alloc(s, s->d_iter, 16))) return rc;
        if ((rc = dev_alloc(s, s->d_snap_cur, 16))) return rc;
        if ((rc = dev_alloc(s, s->d_coef_cur, DC_COEF * 4))) return rc;
    }
    if ((s->B != B || s->T != T) && s->d_gran)       // another geometry: no granule of the shared combine may carry a tag a launch could expect
        HIP_TRY(hipMemsetAsync(s->d_gran, 0, s->cap_B * 1024 * 8, s->stream));
    s->B = B;
    s->T = T;
    s->Tx = Tx;
    s->M = M;
    s->G = G;
    {   // per-workgroup partial records when a workgroup (NW groups) cannot span more than two clips
        const int nw = s->split_small ? 4 : 8;
        (void)nw;
        s->gran = 32;   // per-group records (a workgroup-level LDS pre-reduction was measured slower overall)
    }
    return DC_OK;
}

int ensure_steps(dc_sampler* s, int S) {
    if ((size_t)S > s->cap_steps) {
        drop_graph(s);
        s->cap_steps = 0;
        s->tables_S = 0;
        int rc;
        if ((rc = dev_alloc(s, s->d_t_of_iter, (size_t)S * 4))) return rc;
        if ((rc = dev_alloc(s, s->d_snap_of_iter, (size_t)S * 4))) return rc;
        if ((rc = dev_alloc(s, s->d_coef_of_t, (size_t)S * DC_COEF * 4))) return rc;
        if ((rc = dev_alloc(s, s->d_coef_of_iter, (size_t)S * DC_COEF * 4))) return rc;
        s->cap_steps = (size_t)S;
    }
    return DC_OK;
}

struct Timed {   // RAII-less helper: wraps a launch with events when profiling
    dc_sampler* s;
    int id;
};

#define LAUNCH(id, expr)                                              \
    do {                                                              \
        if (s->prof.on) {                                             \
            hipEvent_t a_, b_;                                        \
            HIP_TRY(hipEventCreate(&a_));                             \
            HIP_TRY(hipEventCreate(&b_));                             \
            HIP_TRY(hipEventRecord(a_, st));                          \
            HIP_TRY(expr);                                            \
            HIP_TRY(hipEventRecord(b_, st));                          \
            s->prof.ev.push_back(a_);                                 \
            s->prof.ev.push_back(b_);                                 \
            s->prof.ids.push_back(id);                                \
        } else {                                                      \
            HIP_TRY(expr);                                            \
        }                                                             \
    } while (0)

// Everything a captured graph bakes in besides (B, T, K): the environment switches that pick the launch form (read per call, so
// that one process can A/B them: a change re-captures) and the update options of the loop.
unsigned long long form_key(const dc_sampler* s) {
    static const char* sw[] = {"DC_NO_WGREC", "DC_NO_NARROW", "DC_NO_ALIGN", "DC_ALIGN", "DC_NO_FUSE_EMBED", "DC_FILM_STATIC",
                               "DC_BEGIN_STEP", "DC_NO_PAD", "DC_NO_LAYER16", "DC_L16_OWN_COMBINE", "DC_L16_TEST_DROP_SLICE", "DC_TAIL_FILM_BF16", "DC_FLAT_UNITS",
                               "DC_NO_EMBED_NEXT"};
    unsigned long long k = 0;
    for (size_t i = 0; i < sizeof sw / sizeof *sw; ++i) k |= (getenv(sw[i]) ? 1ull : 0ull) << i;
    k |= (unsigned long long)(s->upd_flags & 0xff) << 16;     // (the noise tensor's address is not baked in: the kernels read it from d_zslot)
    k |= (s->l16_own ? 1ull : 0ull) << 24;
    k |= (unsigned long long)((s->clip_aligned + 1) & 3) << 25;
    return k;
}

// One denoiser evaluation (+ DDIM update when loop_mode) enqueued on st.
// graph_step >= 0: step number inside a graph being captured.  On the default path (fused SiLU fill, per-layer launches) the
// step's kernels then look the timestep / DDIM scalars up themselves - this step's slot of the per-iteration tables, offset by
// the iteration at which the replay began (*d_iter, advanced once per replay) - and the per-step bookkeeping launch
// (k_begin_step, 5 us + a launch gap) is dropped.
int enqueue_step(dc_sampler* s, hipStream_t st, bool loop_mode, const float* x_src, float* x_dst, int graph_step = -1,
                 bool split_step = false /* this evaluation's 128-wide GEMMs on split operands (the fp16 images' hi + lo halves) */,
                 bool g1_loop = false /* a loop with a precise tail: its plain-operand evaluations read G' scale tiles */,
                 bool next_plain = false /* loops: another step follows in this enqueue sequence (same graph) and it is a plain-operand evaluation */) {
    const int B = s->B, T = s->T, M = s->M, G = s->G, L = s->cfg.num_layers;
    const bool embedded = s->embedded_by_prev;      // the previous step's last layer has embedded x and run layer 0's front half for this step
    s->embedded_by_prev = false;
    const bool ss = s->split_small || split_step, sf = s->split_film;
    const DcModel* dmod = (split_step && !s->split_small) ? s->d_model_split : s->d_model;      // (the precise tail's split stage images)
    // (bf16 precision, split evaluations: the f16 FiLM image - the step is then exactly a "mixed" evaluation)
    const bool film_tail = split_step && !s->split_small && s->h_model.film_w16_tail != nullptr && !getenv("DC_TAIL_FILM_BF16");
    const int fs = s->small_fmt, ff = film_tail ? 1 : s->film_fmt;
    // non-split formats: the FiLM GEMM produces its own operand from pp + temb (no k_silu_emb pass); the separate pass
    // remains for the split formats, for the v1 kernel, and under the test hooks that read the operand image back
    const bool fuse_silu = !sf && s->dbg_layers < 0;
    const bool folded = loop_mode && graph_step >= 0 && fuse_silu && !s->cfg.no_eff && !getenv("DC_BEGIN_STEP") && s->dbg_stage == 0;
    const int* iter_base = folded ? s->d_iter : nullptr;
    const int* t_src = folded ? s->d_t_of_iter + graph_step : s->d_t_clip;
    const float* coef_src = folded ? s->d_coef_of_iter + DC_COEF * (size_t)graph_step : s->d_coef_cur;
    const int* snap_src = folded ? s->d_snap_of_iter + graph_step : s->d_snap_cur;
    if (graph_step >= 0) s->graph_folded = folded;
    if (loop_mode && !folded)
        LAUNCH(K_BEGIN, dc_launch_begin_step(st, s->d_iter, s->d_t_of_iter, s->d_coef_of_t, s->d_snap_of_iter,
                                             s->d_t_clip, s->d_coef_cur, s->d_snap_cur, B));
    if (loop_mode && (s->upd_flags & DC_UPD_ZSTEP))       // this iteration's draws (eta > 0, library-generated): consumed by the last layer's epilogue
        LAUNCH(K_NOISE, dc_launch_step_noise(st, s->d_zstep, (size_t)B * s->Tx * s->cfg.input_feats, 0, reinterpret_cast<const unsigned long long*>(s->d_zslot) + 1,
                                             iter_base, folded ? graph_step : 0,
                                             folded ? nullptr : s->d_snap_cur, 0));
    if (!fuse_silu)
        LAUNCH(K_SILU, dc_launch_silu_emb(st, ff, sf, s->d_pp, s->h_model.temb, s->d_t_clip, s->d_s_hi, s->d_s_lo, G, T, B));
    static const bool want_stamps_film = getenv("DC_STAMPS") != nullptr;       // clock stamps land in stamp slots 28..31 of wave 7
    // adaptive work shares of the persistent FiLM GEMM (dc_kernels.hip, film_shares); DC_FILM_STATIC=1 keeps equal shares
    const bool film_static = getenv("DC_FILM_STATIC") != nullptr;           // (read per call: the tests toggle it)
    const bool adapt = !film_static && s->num_cu <= 1024;
    // ---- form of the layer launches (linear attention) --------------------------------------------------------------------
    // workgroup-level records (no combine launches) whenever a workgroup's 256 tokens cannot touch more than two clips
    const bool no_wgr = getenv("DC_NO_WGREC") != nullptr;          // (read per call: the tests toggle it)
    // (split formats: on clip-aligned units only - the doubled weight images leave LDS for ONE clip's attention fragments - and in
    // the production build only: the test hooks keep the per-group form)
    const bool wgr = T >= 256 && !no_wgr && s->dbg_first < 0 && !s->cfg.no_eff &&
                     (!ss || (T % 32 == 0 && s->dbg_layers < 0 && s->dbg_stage == 0 && !getenv("DC_NO_ALIGN")));
    static const bool want_stamps = getenv("DC_STAMPS") != nullptr;
    // Narrow workgroups (4 waves = 128-token units, one wave per SIMD) while every unit still gets a CU of its own: the layer
    // kernel is bound by instruction issue, so a wave alone on its SIMD runs a layer in about half the time (DESIGN.md
    // section 4).  T <= 3840: the narrow combine holds 32 units per clip.  DC_NO_NARROW=1 keeps the 8-wave form (read per call).
    // Clip-aligned units (WgMap in dc_dev.h; needs a clip stride of whole groups): upc workgroups per clip, no workgroup spans two
    // clips.  Default for the narrow (small-batch) form; with the chip full (bs=32 x 1800: 256 workgroups instead of 228 flat
    // units) it measured 1.2 % slower than flat units - DC_ALIGN=1 forces it there.
    const bool can_align = wgr && T % 32 == 0 && !getenv("DC_NO_ALIGN");
    const int upc_wide = (T + 255) / 256, upc_narrow = (T + 127) / 128;
    const bool aligned_env = can_align && getenv("DC_ALIGN") != nullptr;
    const int nwg_narrow = can_align ? B * upc_narrow : (G + 3) / 4;
    const bool narrow = wgr && !ss && nwg_narrow <= s->num_cu && T <= 3840 && s->dbg_layers < 0 && s->dbg_stage == 0 &&
                        !getenv("DC_NO_NARROW") && !want_stamps;
    // ... and for the wide (chip-full) form whenever the clip-aligned launch needs no more rounds of workgroups over the chip than the flat
    // one (bs = 32 x 1800: 256 workgroups instead of 228, one round either way): a clip's result is then bit-identical whatever the
    // batch around it - the reference's semantics (transformer.py:111: the key softmax is per clip) - for +1.6 ... +2.1 % per loop
    // (profiles/r06_ab_align.txt).  Where it would cost a round (bs = 35 x 1800: 280 against 250 workgroups on 256 CUs) flat units stay -
    // a clip then depends on its neighbours at the rounding level (4e-4).  dc_sampler_set_clip_aligned: 1 forces
    // aligned units, 0 flat ones; DC_ALIGN=1 / DC_FLAT_UNITS=1 in the environment do the same per process.
    const int nwg_flat = (G + 7) / 8, ncu = s->num_cu > 0 ? s->num_cu : 256;
    const bool same_rounds = ((long long)B * upc_wide + ncu - 1) / ncu == ((long long)nwg_flat + ncu - 1) / ncu;
    const bool aligned_wide = s->clip_aligned > 0 || aligned_env || (s->clip_aligned < 0 && same_rounds && !getenv("DC_FLAT_UNITS"));
    const bool aligned = can_align && (narrow || ss || aligned_wide);
    // 16-token waves (dc_layer16.hip) while every clip-aligned 64-token unit still gets a CU of its own (bs <= 8 at T = 1800): in that
    // regime the layer is bound by the LENGTH of one wave's dependency chain, and a 16-token wave's is about half as long.  The
    // embedding stays the narrow 32-token form (its 128-token unit records feed layer 0).  DC_NO_LAYER16=1 keeps the 32-token form.
    const int upc16 = (T + 63) / 64;
    const bool layer16 = narrow && aligned && (long long)B * upc16 <= s->num_cu && upc16 <= dc_layer16_max_units() && s->dbg_first < 0 &&
                         !getenv("DC_NO_LAYER16");
    const int upc = aligned ? (narrow ? upc_narrow : upc_wide) : 0;
    const int nwg = aligned ? B * upc : (narrow ? (G + 3) / 4 : (G + 7) / 8);
    // k_layer16: the clip's workgroups share the combine of the previous layer's unit records inside the launch (dc_layer16.hip;
    // DC_L16_OWN_COMBINE=1: every workgroup combines alone, round 4's form).  Tags: captured steps 16 * (graph step + *d_iter) + layer + 1,
    // eager launches from a sequence of their own above them - consecutive launches never share a tag.
    const bool l16_shared = layer16 && !s->l16_own && !getenv("DC_L16_OWN_COMBINE");
    const unsigned l16_tag = folded ? 16u * (unsigned)graph_step : (0x40000000u | (16u * (s->l16_seq++ & 0x3ffffffu)));
    const size_t rec_stride = wgr ? (size_t)nwg * 2 * DC_REC_FLOATS : 0;
    // (the kernels write records at recs + rec_stride + wg * 2 * DC_REC_FLOATS: both alternating buffers must lie inside d_recs)
    if ((wgr ? 2 * rec_stride : (size_t)G * 2 * DC_REC_FLOATS) > s->cap_rec_floats)
        return fail(DC_ERR_INVALID, "unit records of this launch form (%zu floats) exceed the workspace (%zu)",
                    wgr ? 2 * rec_stride : (size_t)G * 2 * DC_REC_FLOATS, s->cap_rec_floats);
    const int Tx = s->Tx;
    // k_embed_front rides in the FiLM GEMM's launch (wide flat units, non-split formats, no test hooks; DC_NO_FUSE_EMBED=1 and the
    // per-kernel profile pass keep the two launches): one kernel boundary less per step, -1.3 % per loop at bs=32
    // (flat units in the non-split formats; the "mixed" mode - f16 GEMM, split-bf16 embedding - on its clip-aligned units)
    const bool mixed_form = ss && !sf && ff == 1 && fs == 0;
    // The last layer of a plain wide step does the NEXT step's front work (embedding of x_{t-1} + layer 0's self-attention front half:
    // k_layer, DC_UPD_EMBED_NEXT) when that step is a plain wide step of the same enqueue sequence; its FiLM launch is then the bare GEMM
    // and it has no front launch.  DC_NO_EMBED_NEXT=1 keeps the front work in every step's own FiLM launch.
    const bool embed_next_on = !getenv("DC_NO_EMBED_NEXT");            // (read per call: a test toggles it)
    const bool wide_plain = wgr && !narrow && !ss && fuse_silu && ff == fs && s->dbg_layers < 0 && s->dbg_stage == 0 && s->dbg_first < 0 &&
                            !want_stamps && !s->cfg.no_eff;
    const bool embed_next = embed_next_on && loop_mode && next_plain && wide_plain;
    if (embedded && !(loop_mode && wide_plain)) return fail(DC_ERR_INVALID, "internal: a step whose front work was done by its predecessor changed its launch form");
    const bool fuse_embed = !embedded && wgr && !narrow && (ss ? (aligned && mixed_form) : ff == fs) && fuse_silu && s->dbg_layers < 0 &&
                            s->dbg_stage == 0 && nwg <= s->num_cu && !want_stamps && !s->prof.on && !getenv("DC_NO_FUSE_EMBED");
    // small batches (narrow clip-aligned units): the embedding's workgroups ride BEHIND the GEMM's in the FiLM launch
    // (film_extra_workgroups, dc_kernels.hip): one launch (15 us at one clip) and one kernel boundary less per step.  DC_NO_FUSE_EMBED=1 keeps the two launches.
    const bool fuse_extra = narrow && aligned && !ss && ff == fs && fuse_silu && s->h_model.film_w16 && s->dbg_first < 0 && !s->prof.on &&
                            !getenv("DC_NO_FUSE_EMBED");
    DcEmbedArgs ea{};
    if (fuse_embed) ea = DcEmbedArgs{dmod, x_src, s->d_h, s->d_recs, s->d_length, M, Tx, nwg, aligned ? upc : 0, ss ? 1 : 0, 0};
    if (fuse_extra) ea = DcEmbedArgs{s->d_model, x_src, s->d_h, s->d_recs, s->d_length, M, Tx, nwg, upc, 0, 1};
    const DcUpdate upd{s->d_zslot, s->d_status, (loop_mode ? s->upd_flags : 0) | (getenv("DC_L16_TEST_DROP_SLICE") ? DC_UPD_TEST_DROP_SLICE : 0) |
                                                (embed_next ? DC_UPD_EMBED_NEXT : 0),
                       folded ? graph_step : -1, nullptr};
    const int film_rounds = s->NT / 16;
    // scale tiles: G' for the plain-operand consumers of this step, G' - 1 for the split-operand ones (dc_dev.h, film_affine)
    // (the production forms of the plain-operand kernels only: test hooks, stamps and the per-group record form keep G' - 1)
#ifdef DC_NO_FILM_G1
    const bool g1_tiles = false;
#else
    const bool g1_tiles = g1_loop && !ss && wgr && !s->cfg.no_eff && s->dbg_stage == 0 && s->dbg_layers < 0 && s->dbg_first < 0 && !want_stamps;
#endif
    const float* film_b = g1_tiles ? s->h_model.film_b_g1 : s->h_model.film_b;
    const float* film_b16 = g1_tiles ? s->h_model.film_b16_g1 : s->h_model.film_b16;
    // DC_DIAG_SKIP_FILM=1 (diagnostic, eager passes only, results invalid): the FiLM GEMM is launched once and never again - the layers then
    // read stale tiles and run without the GEMM's 300 us of power-limited matrix work between them (what the chip's clock management
    // does to the layer launches that follow a GEMM: tools/diag_clock_coupling.py)
    const bool diag_skip_film = getenv("DC_DIAG_SKIP_FILM") && !fuse_embed && !fuse_extra && s->diag_film_done;
    s->diag_film_done = true;
    if (!diag_skip_film)
    LAUNCH(K_FILM, dc_launch_film_gemm(st, ff, sf, s->h_model.film_w, film_b, s->d_s_hi, s->d_s_lo, s->d_E, G, s->NT, 0,
                                       film_rounds, fuse_silu ? s->d_pp : nullptr, s->h_model.temb, t_src, T, B,
                                       want_stamps_film ? s->d_stamps + 252 : nullptr,
                                       adapt ? s->d_film_rate + 1024 * s->film_rate_parity : nullptr,
                                       adapt ? s->d_film_rate + 1024 * (s->film_rate_parity ^ 1) : nullptr, iter_base,
                                       film_tail ? s->h_model.film_w16_tail : s->h_model.film_w16, film_b16,
                                       (fuse_embed || fuse_extra) ? &ea : nullptr, s->d_status));
    s->film_rate_parity ^= 1;
    const int nl_run = (s->dbg_layers >= 0 && s->dbg_layers < L) ? s->dbg_layers : L;
    if (s->cfg.no_eff) {
        LAUNCH(K_EMBED, dc_launch_embed_front_full(st, fs, ss, dmod, x_src, s->d_h, s->d_kv_sa[0], M, T, B, s->KT));
        for (int l = 0; l < nl_run; ++l) {
            DcUpdate u = upd;              // (stage stamps of layer 3, tools/stage_stamps_full.py + a -DDC_FULL_STAMPS build: DcUpdate::stamps carries the buffer)
            u.stamps = (want_stamps && l == 3) ? s->d_stamps : nullptr;
            LAUNCH(K_LAYER, dc_launch_layer_full(st, fs, ss, dmod, l, s->d_h, s->d_E, s->NT, s->d_kv_sa[l & 1], s->d_kv_sa[(l + 1) & 1],
                                                 s->d_kv_ca, s->d_length, x_src, x_dst, loop_mode ? 1 : 0, s->d_coef_cur,
                                                 s->d_snap_cur, s->d_snaps, M, T, B, s->KT,
                                                 (l == nl_run - 1) ? (s->dbg_stage ? s->dbg_stage : (nl_run < L ? 3 : 0)) : 0, u));
        }
        return DC_OK;
    }
    if (fuse_embed || fuse_extra || embedded) {
        // (embedded by the FiLM launch, or by the previous step's last layer)
    } else if (s->dbg_first >= 0)
        LAUNCH(K_EMBED, dc_launch_front_from_h(st, fs, ss, dmod, s->d_h, s->d_recs, s->d_length, M, T, G, B, s->dbg_first));
    else
        LAUNCH(K_EMBED, dc_launch_embed_front(st, fs, ss, wgr, dmod, x_src, s->d_h, s->d_recs, s->d_length, M, T, G, B,
                                              want_stamps_film ? s->d_stamps + 256 : nullptr, narrow, Tx, upc));
    for (int l = s->dbg_first >= 0 ? s->dbg_first : 0; l < nl_run; ++l) {
        const int dbg = (l == nl_run - 1) ? s->dbg_stage : 0;
        if (layer16) {
            DcUpdate u16 = upd;       // (-DDC_L16_STAMPS builds: stage stamps of layer 3, tools/stage_stamps16.py)
            static const bool stamps16 = getenv("DC_L16_STAMPS") != nullptr;
            u16.stamps = (stamps16 && l == 3) ? s->d_stamps : nullptr;
            LAUNCH(K_LAYER, dc_launch_layer16(st, fs, s->d_model, l, s->d_h, s->d_E, s->NT, s->d_a_ca16, s->d_recs, s->d_length, x_src, x_dst,
                                              loop_mode ? 1 : 0, coef_src, snap_src, s->d_snaps, M, T, B, upc16, rec_stride,
                                              l == 0 ? upc_narrow : upc16, l == 0 ? (size_t)2 * DC_REC_FLOATS : (size_t)DC_REC_FLOATS, iter_base, Tx,
                                              u16, l16_shared ? s->d_gran : nullptr, l16_tag, g1_tiles));
            continue;
        }
        if (!wgr) LAUNCH(K_COMBINE, dc_launch_attn_combine(st, fs, s->d_recs, s->d_a_sa, T, (M + s->gran - 1) / s->gran, B, 1, s->gran));
        LAUNCH(K_LAYER, dc_launch_layer(st, fs, ss, wgr, dmod, l, s->d_h, s->d_E, s->NT, s->d_a_sa, s->d_a_ca, s->d_recs,
                                        s->d_length, x_src, x_dst, loop_mode ? 1 : 0, coef_src, snap_src,
                                        s->d_snaps, M, T, G, B, dbg, ((l == 3 || l == 4) && want_stamps) ? s->d_stamps : nullptr, rec_stride,
                                        iter_base, narrow, Tx, upc, upd, g1_tiles));
    }
    s->embedded_by_prev = embed_next;
    return DC_OK;
}

int sync_in(dc_sampler* s, hipStream_t user) {
    HIP_TRY(hipEventRecord(s->ev_in, user));
    HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_in, 0));
    return DC_OK;
}
int sync_out(dc_sampler* s, hipStream_t user) {
    HIP_TRY(hipEventRecord(s->ev_out, s->stream));
    HIP_TRY(hipStreamWaitEvent(user, s->ev_out, 0));
    return DC_OK;
}

int steps_per_graph(int S) {
    if (S <= 64) return S;
    for (int k = 64; k >= 1; --k)
        if (S % k == 0) return k;
    return 1;
}

// h_coef: [S][DC_COEF] per-timestep scalars (dc_common.h); flags: DC_UPD_*; d_step_noise: [S][B][Tx][P] or nullptr
#ifndef DC_BF16_TAIL_DEFAULT
#define DC_BF16_TAIL_DEFAULT 6
#endif
#ifndef DC_BF16_SHORT_CLIP
#define DC_BF16_SHORT_CLIP 100     // bf16 precision: loops over clips of fewer frames run every evaluation split (loop_common)
#endif
}  // namespace
extern "C" DC_EXPORT int32_t dc_precise_tail_default(int32_t precision);
namespace {
// split-operand evaluations (the precise tail, dc_sampler_set_precise_forward) exist for: fp16 / bf16 precision, linear attention, no test hooks
bool can_split_steps(const dc_sampler* s) {
    return (s->cfg.precision == DC_PREC_FP16 || s->cfg.precision == DC_PREC_BF16) && s->dbg_layers < 0 && s->dbg_first < 0 &&
           s->dbg_stage == 0 && s->d_model_split;
}

int loop_common(dc_sampler* s, const float* d_noise, float* d_out, int S, const float* h_coef,
                const int32_t* h_snap_iters, int n_snap, float* d_snaps_user, hipStream_t user, bool profile,
                int flags = 0, const float* d_step_noise = nullptr) {
    if (!s || !s->finalized) return fail(DC_ERR_INVALID, "sampler not finalized");
    if (!s->cond_set) return fail(DC_ERR_INVALID, "dc_sampler_set_conditioning must be called first");
    if (S < 1 || S > s->cfg.max_timesteps) return fail(DC_ERR_INVALID, "num_steps %d outside [1, max_timesteps=%d]", S, s->cfg.max_timesteps);
    if (!d_noise || !d_out || !h_coef) return fail(DC_ERR_INVALID, "null pointer argument");
    // EPSILON model x full attention x eta = 0: outside the 1e-3 bound on some loops even with every GEMM on split operands (scores, weights
    // and values of the attention stay plain fp16, and a deterministic EPSILON chain keeps every evaluation's error in x_t): 2.3e-4 ... 1.26e-3
    // over 14 randomized loops of tools/fuzz_sampler.py (profiles/r06_fuzz_final.txt).  Refused, not returned; with eta > 0 the fresh noise
    // damps it (<= 2e-4 on the same tool), and linear attention reads <= 2.8e-4 at eta = 0.
    if ((flags & DC_UPD_EPS) && s->cfg.no_eff && !(flags & DC_UPD_NOISY) && !getenv("DC_ALLOW_EPSILON_NO_EFF_ETA0"))
        return fail(DC_ERR_UNSUPPORTED, "EPSILON model with full attention (no_eff) at eta = 0 is outside the 1e-3 parity bound (up to 1.3e-3); "
                                        "use linear attention, or eta > 0");
    if (s->smooth_window > 0 && s->Tx < s->smooth_window)       // (before anything is enqueued: a failed call leaves no work and no half-ordered streams)
        return fail(DC_ERR_INVALID, "smoothing window %d exceeds the %d frames of a clip", s->smooth_window, s->Tx);
    int rc;
    if ((rc = ensure_steps(s, S))) return rc;
    const size_t MP = (size_t)s->B * s->Tx * s->cfg.input_feats;          // x, snapshots: the caller's layout
    if ((size_t)n_snap * MP > s->cap_snap) {          // (capacity in elements: the batch may have grown since the last call)
        drop_graph(s);
        s->cap_snap = 0;
        if ((rc = dev_alloc(s, s->d_snaps, (size_t)n_snap * MP * 4))) return rc;
        s->cap_snap = (size_t)n_snap * MP;
    }
    // Per-iteration tables on the device: uploaded only when (S, coefficients, snapshot iterations) differ from the
    // previous call - a sampling service calls the loop with the same schedule every time, and the four small H2D
    // copies + the stream synchronisation they need cost as much as several kernels at bs=1.
    std::vector<int> snap_of_iter(S, -1);
    for (int k = 0; k < n_snap; ++k) {
        if (h_snap_iters[k] < 0 || h_snap_iters[k] >= S) return fail(DC_ERR_INVALID, "snapshot iteration %d outside [0,%d)", h_snap_iters[k], S);
        snap_of_iter[h_snap_iters[k]] = k;
    }
    const bool same_tables = s->tables_S == S && s->tab_snap == snap_of_iter &&
                             memcmp(s->tab_coef.data(), h_coef, (size_t)S * DC_COEF * 4) == 0;
    // eta > 0: the caller's [S][B][Tx][P] draws, or - none given, a seed set - one iteration's draws generated at the head of every step
    const float* zbase = nullptr;
    if (flags & DC_UPD_NOISY) {
        if (d_step_noise) {
            zbase = d_step_noise;
        } else {
            if (MP > s->cap_zstep) {
                drop_graph(s);          // (the buffer's address is an argument of the captured k_step_noise launches)
                s->cap_zstep = 0;
                if ((rc = dev_alloc(s, s->d_zstep, MP * 4))) return rc;
                s->cap_zstep = MP;
            }
            zbase = s->d_zstep;
            flags |= DC_UPD_ZSTEP;
        }
    }
    s->upd_flags = flags;
    if ((rc = sync_in(s, user))) return rc;
    hipStream_t st = s->stream;
    // the status word reports on THIS loop: bits left by earlier work on the sampler (a dc_sampler_denoise, a loop nobody asked
    // about) must not fail it
    HIP_TRY(hipMemsetAsync(s->d_status, 0, 4, st));
    HIP_TRY(dc_launch_set_ptr(st, s->d_zslot, zbase, s->noise_seed, s->noise_first));
    if (flags & DC_UPD_ZSTEP) s->noise_seed_set = false;       // a seed serves ONE loop: a later loop without a new one must not replay its draws
    if (!same_tables) {
        HIP_TRY(hipStreamSynchronize(st));          // an earlier call's copies out of the member vectors are done
        s->tables_S = 0;
        s->tab_t.resize(S);
        for (int i = 0; i < S; ++i) s->tab_t[i] = S - 1 - i;            // indices = range(num_timesteps)[::-1] (gaussian_diffusion.py:943)
        s->tab_snap = snap_of_iter;
        s->tab_coef.assign(h_coef, h_coef + (size_t)S * DC_COEF);
        s->tab_coef_iter.resize((size_t)S * DC_COEF);
        for (int i = 0; i < S; ++i) memcpy(&s->tab_coef_iter[DC_COEF * (size_t)i], h_coef + DC_COEF * (size_t)s->tab_t[i], DC_COEF * 4);
        HIP_TRY(hipMemcpyAsync(s->d_t_of_iter, s->tab_t.data(), S * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->d_snap_of_iter, s->tab_snap.data(), S * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->d_coef_of_t, s->tab_coef.data(), (size_t)S * DC_COEF * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->d_coef_of_iter, s->tab_coef_iter.data(), (size_t)S * DC_COEF * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));          // pageable sources: the copies have left the host vectors
        s->tables_S = S;
    }
    HIP_TRY(hipMemsetAsync(s->d_iter, 0, 16, st));
    HIP_TRY(hipMemcpyAsync(s->d_x, d_noise, MP * 4, hipMemcpyDeviceToDevice, st));
    const bool no_graph = getenv("DC_DISABLE_GRAPH") != nullptr;
    // precise tail: the loop's last `tail` model evaluations on split operands (dc_sampler_set_precise_tail; DC_PRECISE_TAIL=k overrides):
    // fp16 / bf16 precision, linear attention, no test hooks
    int tail = s->tail_split >= 0 ? s->tail_split : dc_precise_tail_default(s->cfg.precision);
    bool tail_asked = s->tail_split >= 0;
    if (const char* e = getenv("DC_PRECISE_TAIL")) tail = atoi(e), tail_asked = true;
    // An EPSILON model's final sample is sqrt(1 / abar) x_t - sqrt(1 / abar - 1) eps, not the last evaluations' prediction: what the plain
    // 16-bit evaluations left in x_t stays (eta = 0: fp16 1.4 - 1.8e-3 whatever the tail, tools/fuzz_sampler.py).  Parity first: unless a
    // tail was asked for, such a loop runs EVERY evaluation on split operands (2.1e-4, at the split precisions' speed).
    if (!tail_asked && (flags & DC_UPD_EPS)) tail = S;
    // Clips of fewer than 100 frames in the bf16 precision: a clip's error is a norm over a few hundred numbers (26 per frame), and the worst of
    // a batch of dozens of such clips reached 1.27e-3 with the default tail (39 clips of 36 frames, lengths down to 1: tools/fuzz_shapes.py,
    // profiles/r06_fuzz_final.txt; 8.6e-4 at 39 frames, <= 7.2e-4 from 100 frames up).  Such loops are bound by launch latency, not by the
    // kernels: they run every evaluation in the split form (the `mixed` precision's evaluations, 9e-5) unless a tail was asked for.
    if (!tail_asked && s->cfg.precision == DC_PREC_BF16 && s->Tx < DC_BF16_SHORT_CLIP) tail = S;
    // (clip strides that are not whole 32-frame groups - T = 900 x 128 unpadded - and short clips run the split evaluations in the
    // per-group record form with its combine launches: no measurable cost at one evaluation per loop, 70.6 vs 70.6 ms at bs = 128 x 900)
    if (!can_split_steps(s)) tail = 0;
    // (a tail of the whole loop splits every replay's graph; any shorter one lives in the last replay and is clipped to its steps)
    const bool tail_all = tail >= S;
    tail = std::max(0, std::min(tail, std::min(S, steps_per_graph(S))));
    s->embedded_by_prev = false;
    if (profile || no_graph) {
        s->prof.on = profile;
        for (int i = 0; i < S; ++i)
            if ((rc = enqueue_step(s, st, true, s->d_x, s->d_x, -1, tail_all || i >= S - tail, tail > 0,
                                   i + 1 < S && !(tail_all || i + 1 >= S - tail)))) {
                s->prof.on = false;
                return rc;
            }
        s->prof.on = false;
    } else {
        const int K = steps_per_graph(S);
        const int replays = S / K;
      // (with a precise tail and several replays per loop - S > 64 - the LAST replay runs a second graph whose final steps are split)
      for (int part = 0; part < ((tail && replays > 1) ? 2 : 1); ++part) {
        const int tail_here = (part == 1 || replays == 1) ? tail : (tail_all ? K : 0);
        const int launches = (tail && replays > 1) ? (part == 0 ? replays - 1 : 1) : replays;
        // (g1 bit: the plain evaluations of a loop WITH a tail read G' scale tiles, those of a loop without one G' - 1 tiles - two
        // different captures of the same part-0 graph when S > 64)
        const unsigned long long fk = form_key(s) | ((unsigned long long)tail_here << 40) | ((tail > 0 ? 1ull : 0ull) << 39);
        auto current = [&]() {
            return s->graph && s->graph_B == s->B && s->graph_T == s->T && s->graph_Tx == s->Tx && s->graph_K == K && s->graph_form == fk;
        };
        if (!current()) {       // park the graph at hand, take this shape's from the park when it has been captured before
            if (s->graph) {
                if (s->graph_park.size() >= 3) {
                    hipGraphExecDestroy(s->graph_park.front().exec);
                    s->graph_park.erase(s->graph_park.begin());
                }
                s->graph_park.push_back({s->graph, s->graph_B, s->graph_T, s->graph_Tx, s->graph_K, s->graph_form, s->graph_folded});
                s->graph = nullptr;
            }
            for (size_t i = 0; i < s->graph_park.size(); ++i) {
                const auto& g = s->graph_park[i];
                if (g.B == s->B && g.T == s->T && g.Tx == s->Tx && g.K == K && g.form == fk) {
                    s->graph = g.exec;
                    s->graph_B = g.B, s->graph_T = g.T, s->graph_Tx = g.Tx, s->graph_K = g.K, s->graph_form = g.form, s->graph_folded = g.folded;
                    s->graph_park.erase(s->graph_park.begin() + i);
                    break;
                }
            }
        }
        if (!current()) {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            s->embedded_by_prev = false;      // (a graph's first step does its own front work, its last step nobody else's)
            for (int i = 0; i < K; ++i)
                if ((rc = enqueue_step(s, st, true, s->d_x, s->d_x, i, i >= K - tail_here, tail > 0, i + 1 < K && !(i + 1 >= K - tail_here)))) {
                    hipStreamEndCapture(st, &g);
                    if (g) hipGraphDestroy(g);
                    return rc;
                }
            // steps that indexed the iteration tables themselves did not advance the counter (see enqueue_step)
            if (s->graph_folded) {
                hipError_t ea = dc_launch_advance_iter(st, s->d_iter, K);
                if (ea != hipSuccess) {
                    hipStreamEndCapture(st, &g);
                    if (g) hipGraphDestroy(g);
                    return fail(DC_ERR_HIP, "k_advance_iter: %s", hipGetErrorString(ea));
                }
            }
            HIP_TRY(hipStreamEndCapture(st, &g));
            hipError_t e = hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
            if (e != hipSuccess) return fail(DC_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
            s->graph_B = s->B;
            s->graph_T = s->T;
            s->graph_Tx = s->Tx;
            s->graph_K = K;
            s->graph_form = fk;
        }
        for (int i = 0; i < launches; ++i) HIP_TRY(hipGraphLaunch(s->graph, st));
      }
    }
    // the final write x0 -> the caller's tensor: a copy, or (dc_sampler_set_smoothing) the Savitzky-Golay filter along time
    // (tools/visualization.py:20-26,126) reading the loop's x0 and writing the caller's tensor directly - no pass of its own
    if (s->smooth_window > 0) {
        HIP_TRY(dc_launch_savgol(st, s->d_x, d_out, s->d_smooth_coef, s->B, s->Tx, s->cfg.input_feats, s->smooth_window));
    } else {
        HIP_TRY(hipMemcpyAsync(d_out, s->d_x, MP * 4, hipMemcpyDeviceToDevice, st));
    }
    if (n_snap > 0 && d_snaps_user)
        HIP_TRY(hipMemcpyAsync(d_snaps_user, s->d_snaps, (size_t)n_snap * MP * 4, hipMemcpyDeviceToDevice, st));
    return sync_out(s, user);
}

// the eta = 0 table [S][4] of dc_ddim_coefficients in the internal [S][DC_COEF] form (sigma = 0)
std::vector<float> widen_coef(const float* h_coef, int S) {
    std::vector<float> c8((size_t)(S > 0 ? S : 0) * DC_COEF, 0.f);
    if (h_coef)
        for (int t = 0; t < S; ++t) memcpy(&c8[(size_t)DC_COEF * t], h_coef + 4 * (size_t)t, 16);
    return c8;
}

}  // namespace

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" {

static_assert(DC_UPDATE_CLIP_DENOISED == DC_UPD_CLIP && DC_UPDATE_EPSILON == DC_UPD_EPS && DC_STATUS_F16_SATURATED == DC_STATUS_F16_SAT &&
                  DC_STATUS_TIMEOUT == DC_STATUS_SYNC_TIMEOUT,
              "include/dc_ddim.h and dc_common.h disagree");

const char* dc_last_error(void) { return g_err.c_str(); }
const char* dc_version(void) { return "dc_ddim 0.1 (gfx950)"; }

int dc_linear_beta_schedule(int32_t n, double* betas, double* ac, double* ac_prev, double* sr, double* srm1) {
    if (n < 1 || !betas) return fail(DC_ERR_INVALID, "bad schedule arguments");
    const double scale = 1000.0 / n, b0 = scale * 0.0001, b1 = scale * 0.02;
    double cp = 1.0;
    for (int i = 0; i < n; ++i) {
        // np.linspace(start, stop, n): start + i*step with step=(stop-start)/(n-1); last point is `stop` exactly
        const double step = n > 1 ? (b1 - b0) / (n - 1) : 0.0;
        betas[i] = (i == n - 1 && n > 1) ? b1 : b0 + i * step;
        const double prev = cp;
        cp *= (1.0 - betas[i]);
        if (ac) ac[i] = cp;
        if (ac_prev) ac_prev[i] = prev;
        if (sr) sr[i] = std::sqrt(1.0 / cp);
        if (srm1) srm1[i] = std::sqrt(1.0 / cp - 1.0);
    }
    return DC_OK;
}

int dc_ddim_coefficients(int32_t n, const double* ac, float* coef) {
    if (n < 1 || !ac || !coef) return fail(DC_ERR_INVALID, "bad coefficient arguments");
    for (int t = 0; t < n; ++t) {
        const float a_prev = t == 0 ? 1.0f : (float)ac[t - 1];
        coef[4 * t + 0] = (float)std::sqrt(1.0 / ac[t]);
        coef[4 * t + 1] = (float)std::sqrt(1.0 / ac[t] - 1.0);
        coef[4 * t + 2] = sqrtf(a_prev);
        coef[4 * t + 3] = sqrtf(1.0f - a_prev);
    }
    return DC_OK;
}

int dc_ddim_coefficients_ex(int32_t n, const double* ac, float eta, float* coef8) {
    if (n < 1 || !ac || !coef8 || !(eta >= 0.f)) return fail(DC_ERR_INVALID, "bad coefficient arguments");
    // fp32 arithmetic on the fp32-rounded table entries, in the order ddim_sample evaluates them (gaussian_diffusion.py:812-826)
    for (int t = 0; t < n; ++t) {
        const float a = (float)ac[t], a_prev = t == 0 ? 1.0f : (float)ac[t - 1];
        const float sigma = eta * sqrtf((1.0f - a_prev) / (1.0f - a)) * sqrtf(1.0f - a / a_prev);
        float* c = coef8 + (size_t)DC_COEF * t;
        c[0] = (float)std::sqrt(1.0 / ac[t]);
        c[1] = (float)std::sqrt(1.0 / ac[t] - 1.0);
        c[2] = sqrtf(a_prev);
        c[3] = sqrtf(1.0f - a_prev - sigma * sigma);
        c[4] = sigma;
        c[5] = c[6] = c[7] = 0.f;
    }
    return DC_OK;
}

int dc_pack_weight(const float* w, int32_t n_out, int32_t k_in, int32_t chained, uint16_t* hi, uint16_t* lo) {
    if (!w || !hi || !lo || n_out < 1 || k_in < 1) return fail(DC_ERR_INVALID, "bad pack arguments");
    pack_weight(w, n_out, k_in, chained != 0, hi, lo);
    return DC_OK;
}

int dc_sampler_create(const dc_config* cfg, dc_sampler** out) {
    if (!cfg || !out) return fail(DC_ERR_INVALID, "null argument");
    if (cfg->latent_dim != DC_D || cfg->num_heads != DC_H || cfg->ff_size != DC_F)
        return fail(DC_ERR_UNSUPPORTED, "built for latent_dim=128, num_heads=8, ff_size=64 (got %d, %d, %d); latent_dim*4 must equal 512 "
                    "(transformer.py:385,404,482)", cfg->latent_dim, cfg->num_heads, cfg->ff_size);
    if (cfg->input_feats < 1 || cfg->input_feats > DC_PMAX) return fail(DC_ERR_UNSUPPORTED, "input_feats must be in [1,32]");
    if (cfg->num_layers < 1 || cfg->num_layers > DC_MAX_LAYERS) return fail(DC_ERR_UNSUPPORTED, "num_layers must be in [1,%d]", DC_MAX_LAYERS);
    // (bf16 attention operands - scores, weights and values on 8 mantissa bits - leave 1.0 - 1.8e-3 on x0 whatever the precise tail
    // (tools/fuzz_shapes.py, profiles/r06_fuzz_bf16_tail.txt): outside the parity bound, so the combination is not offered)
    if (cfg->no_eff && cfg->precision != DC_PREC_FP16)
        return fail(DC_ERR_UNSUPPORTED, "no_eff (full T x T attention) is built for the fp16 precision only (bf16 attention operands leave 1 - 2e-3 on x0: outside the 1e-3 parity bound)");
    if (cfg->precision < DC_PREC_BF16 || cfg->precision > DC_PREC_FP16) return fail(DC_ERR_INVALID, "unknown precision %d", cfg->precision);
    if (cfg->max_timesteps < 1) return fail(DC_ERR_INVALID, "max_timesteps must be >= 1");
    int ndev = 0;
#ifdef DC_HOST_SANITIZE
    // Sanitizer build of the HOST half (address + undefined-behaviour sanitizers on the CPU; GPU sanitizers are not available on this
    // pool): without a device the sampler is created "host only" - parameter store, validation, weight folding and packing into the
    // arena run as in production, nothing is uploaded or launched, and every entry point that needs the device fails with NO_DEVICE.
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        dc_sampler* h = new dc_sampler();
        h->cfg = *cfg;
        h->host_only = true;
        h->num_cu = 256;
        h->split_small = cfg->precision == DC_PREC_MIXED || cfg->precision == DC_PREC_BF16X3;
        h->split_film = cfg->precision == DC_PREC_BF16X3;
        h->small_fmt = cfg->precision == DC_PREC_FP16 ? 1 : 0;
        h->film_fmt = (cfg->precision == DC_PREC_FP16 || cfg->precision == DC_PREC_MIXED) ? 1 : 0;
        *out = h;
        return DC_OK;
    }
#endif
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(DC_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(DC_ERR_INVALID, "device %d out of range (0..%d)", cfg->device, ndev - 1);
    HIP_TRY(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DC_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);
    dc_sampler* s = new dc_sampler();
    s->cfg = *cfg;
    s->num_cu = prop.multiProcessorCount;
    s->split_small = cfg->precision == DC_PREC_MIXED || cfg->precision == DC_PREC_BF16X3;
    s->split_film = cfg->precision == DC_PREC_BF16X3;
    s->small_fmt = cfg->precision == DC_PREC_FP16 ? 1 : 0;
    s->film_fmt = (cfg->precision == DC_PREC_FP16 || cfg->precision == DC_PREC_MIXED) ? 1 : 0;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming) != hipSuccess) {
        delete s;
        return fail(DC_ERR_HIP, "stream/event creation failed");
    }
    *out = s;
    return DC_OK;
}

void dc_sampler_destroy(dc_sampler* s) {
    if (!s) return;
    if (s->host_only) {          // (sanitizer build without a device: the arena is host memory)
        free(s->d_arena);
        delete s;
        return;
    }
    hipSetDevice(s->cfg.device);
    hipDeviceSynchronize();
    drop_graph(s);
    void* ptrs[] = {s->d_arena, s->d_model, s->d_model_split, s->d_length, s->d_pp, s->d_s_hi, s->d_s_lo, s->d_E, s->d_h, s->d_recs, s->d_a_sa,
                    s->d_a_ca, s->d_x, s->d_snaps, s->d_recs_ca, s->d_nh_hi, s->d_nh_lo, s->d_iter,
                    s->d_t_clip, s->d_snap_cur, s->d_t_of_iter, s->d_snap_of_iter, s->d_coef_cur, s->d_coef_of_t, s->d_coef_of_iter,
                    s->d_kv_sa[0], s->d_kv_sa[1], s->d_kv_ca, s->d_stamps, s->d_film_rate, s->d_status, s->d_smooth_coef, s->d_zslot, s->d_zstep, s->d_a_ca16, s->d_gran};
    for (void* p : ptrs)
        if (p) hipFree(p);
    dc_music_destroy(s->music);
    if (s->ev_in) hipEventDestroy(s->ev_in);
    if (s->ev_out) hipEventDestroy(s->ev_out);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

int dc_sampler_set_param(dc_sampler* s, const char* name, const float* data, int64_t numel) {
    if (!s || !name || !data || numel < 0) return fail(DC_ERR_INVALID, "null argument");
    const std::string n(name);
    if (music_param(n)) {
        for (const auto& r : dc_music_required(DC_C))
            if (r.first == n) {
                if ((size_t)numel != r.second) return fail(DC_ERR_PARAM, "parameter %s has %lld elements, expected %zu", name, (long long)numel, r.second);
                s->params[n].assign(data, data + numel);
                s->finalized = false;
                return DC_OK;
            }
        return DC_OK;
    }
    for (const auto& r : required_params(s->cfg))
        if (r.name == n) {
            if ((size_t)numel != r.numel) return fail(DC_ERR_PARAM, "parameter %s has %lld elements, expected %zu", name, (long long)numel, r.numel);
            s->params[n].assign(data, data + numel);
            s->finalized = false;
            return DC_OK;
        }
    return fail(DC_ERR_PARAM, "unknown parameter key '%s'", name);
}

int dc_sampler_finalize_params(dc_sampler* s) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    for (const auto& r : required_params(s->cfg))
        if (!find(s, r.name)) return fail(DC_ERR_PARAM, "missing parameter '%s'", r.name.c_str());
    if (!s->host_only) HIP_TRY(hipSetDevice(s->cfg.device));
    drop_graph(s);
    s->cap_G = 0;   // NT may have changed: force workspace rebuild
    int rc = build_model(s);
    if (rc) return rc;
    if (s->host_only) {          // (the music encoder's weights are folded straight into device memory: not built without one)
        s->finalized = true;
        s->cond_set = false;
        return DC_OK;
    }
    if (s->music) {
        dc_music_destroy(s->music);
        s->music = nullptr;
    }
    bool any_music = false;
    for (const auto& kv : s->params) any_music = any_music || music_param(kv.first);
    if (any_music) {
        std::string err;
        s->music = dc_music_build(s->params, DC_C, &err);
        if (!s->music) return fail(DC_ERR_PARAM, "music encoder: %s", err.c_str());
        // one fp16 plane where the denoiser rounds the features to 16-bit operands anyway, split planes for the split-operand precisions
        dc_music_set_format(s->music, s->me_format >= 0 ? s->me_format : (s->split_small ? DC_ME_SPLIT : DC_ME_FP16));
    }
    s->finalized = true;
    s->cond_set = false;
    return DC_OK;
}

int dc_sampler_set_conditioning(dc_sampler* s, const float* d_xf_proj, const float* d_xf_out, const int32_t* h_length,
                                int32_t B, int32_t T, void* stream) {
    if (!s || !s->finalized) return fail(DC_ERR_INVALID, "sampler not finalized");
    if (!d_xf_proj || !d_xf_out || B < 1 || T < 1) return fail(DC_ERR_INVALID, "bad conditioning arguments (need B >= 1, T >= 1)");
    if (T < 32 && clip_stride(s, B, T) < 32)
        return fail(DC_ERR_UNSUPPORTED, "T=%d: clips shorter than 32 frames need the padded clip stride (linear attention, DC_NO_PAD unset)", T);
    if (clip_stride(s, B, T) / 32 + 2 > 128) return fail(DC_ERR_UNSUPPORTED, "T=%d: the attention combine holds at most 128 token groups per clip (T <= 4032)", T);
    if (T > s->cfg.num_frames) return fail(DC_ERR_INVALID, "T=%d exceeds num_frames=%d rows of sequence_embedding", T, s->cfg.num_frames);
    if (s->host_only) return fail(DC_ERR_NO_DEVICE, "host-only sampler (sanitizer build without a device)");
    HIP_TRY(hipSetDevice(s->cfg.device));
    int rc;
    if ((rc = ensure_workspace(s, B, T))) return rc;
    std::vector<int> len(B, T);
    if (h_length)
        for (int b = 0; b < B; ++b) {
            if (h_length[b] < 1 || h_length[b] > T) return fail(DC_ERR_INVALID, "length[%d]=%d outside [1,%d]", b, h_length[b], T);
            len[b] = h_length[b];
        }
    hipStream_t user = (hipStream_t)stream, st = s->stream;
    if ((rc = sync_in(s, user))) return rc;
    const int M = s->M, G = s->G, L = s->cfg.num_layers;
    const int Tx = T;
    T = s->T;                         // clip stride of the token space from here on (>= Tx)
    // The clip lengths are uploaded only when they differ from what the device holds (a service or an evaluation run passes the same
    // ones batch after batch): the upload reads host memory and needs the stream synchronisation below, which would otherwise make
    // every call wait for the previous batch's sampling loop (evaluate.py keeps the GPU's queue full).
    const bool same_len = s->len_dev == len && s->len_dev_ptr == s->d_length;
    if (!same_len) {
        HIP_TRY(hipStreamSynchronize(st));                 // an earlier upload out of the member vector is done
        s->len_dev = len;
        s->len_dev_ptr = nullptr;
        HIP_TRY(hipMemcpyAsync(s->d_length, s->len_dev.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    }
    // emb's step-invariant term: linear(xf_proj) as fp32 operand image
    // (split-bf16 MFMAs, k_cond_pp64; DC_COND_512=1: the fp32 FMA form)
    if (!getenv("DC_COND_512"))
        HIP_TRY(dc_launch_cond_pp64(st, d_xf_proj, s->h_model.lin_pack, s->h_model.lin_b, s->d_pp, M, G, T, Tx));
    else
        HIP_TRY(dc_launch_cond_embed(st, 0, d_xf_proj, s->h_model.lin_wt, s->h_model.lin_b, s->d_pp, nullptr, nullptr, M, G, T, Tx));
    // cross-attention: linear(xf_out) -> text_norm (affine folded into K/V) -> per-layer K,V -> A_ca; one-time cost: always split
    // precision (plain bf16 here alone costs ~2e-3 on A_cross).  The linear-attention records come straight from the 64 music
    // features (`linear` composed into the projections on the host, k_cond_ca_partials64: an eighth of the products, no [tokens][512]
    // image in between); DC_COND_512=1 and the full-attention keys / values take the image (k_cond_embed<1>).
    const bool cond64 = !s->cfg.no_eff && !getenv("DC_COND_512");
    if (!cond64)
        HIP_TRY(dc_launch_cond_embed(st, 1, d_xf_out, s->h_model.lin_wt, s->h_model.lin_b, nullptr, s->d_nh_hi, s->d_nh_lo, M, G, T, Tx));
    if (s->cfg.no_eff) {
        HIP_TRY(dc_launch_ca_kv(st, s->small_fmt, s->d_model, s->d_nh_hi, s->d_nh_lo, s->d_kv_ca, M, T, G, B, s->KT, L));
    } else {
        if (cond64)      // (1 / std per token goes through the unused image buffer)
            HIP_TRY(dc_launch_ca_partials64(st, s->d_model, d_xf_out, s->h_model.lin_gram, reinterpret_cast<float*>(s->d_nh_hi), s->d_recs_ca, M, T, G, L, Tx));
        else
            HIP_TRY(dc_launch_ca_partials(st, s->d_model, s->d_nh_hi, s->d_nh_lo, s->d_recs_ca, M, T, G, L, Tx));
        HIP_TRY(dc_launch_attn_combine(st, s->small_fmt, s->d_recs_ca, s->d_a_ca, T, G, B, L, 32));
        if (!s->split_small) HIP_TRY(dc_launch_cond_af16(st, s->small_fmt, s->d_a_ca, s->d_a_ca16, L * B));      // (small batches: dc_layer16.hip)
    }
    if (!same_len) {
        HIP_TRY(hipStreamSynchronize(st));   // the lengths came out of host memory
        s->len_dev_ptr = s->d_length;
    }
    s->cond_set = true;
    return sync_out(s, user);
}

int dc_sampler_set_precise_tail(dc_sampler* s, int32_t steps) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    if (steps < -1) return fail(DC_ERR_INVALID, "precise tail: steps >= 0, or -1 for the precision's default");
    s->tail_split = steps;
    return DC_OK;
}

int32_t dc_precise_tail_default(int32_t precision) {
    return precision == DC_PREC_FP16 ? 1 : precision == DC_PREC_BF16 ? DC_BF16_TAIL_DEFAULT : 0;
}

int dc_sampler_set_clip_aligned(dc_sampler* s, int32_t mode) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    if (mode < -1 || mode > 1) return fail(DC_ERR_INVALID, "clip-aligned units: 1 (always), 0 (flat units), -1 (the library's rule)");
    s->clip_aligned = mode;
    return DC_OK;
}

int dc_sampler_set_precise_forward(dc_sampler* s, int32_t on) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    s->precise_forward = on != 0;
    return DC_OK;
}

int dc_sampler_set_combine_exchange(dc_sampler* s, int32_t on) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    s->l16_own = on == 0;
    return DC_OK;
}

int dc_sampler_set_encoder_format(dc_sampler* s, int32_t format) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    if (format != DC_ME_SPLIT && format != DC_ME_FP16) return fail(DC_ERR_INVALID, "encoder format %d (DC_ME_SPLIT = 0, DC_ME_FP16 = 1)", format);
    s->me_format = format;                                  // (kept across dc_sampler_finalize)
    if (s->music) dc_music_set_format(s->music, format);
    return DC_OK;
}

int dc_sampler_encode_music(dc_sampler* s, const float* d_mel, int32_t B, int32_t Tm, int32_t n_mels, float* d_xf_proj,
                            float* d_xf_out, void* stream) {
    if (!s || !s->finalized) return fail(DC_ERR_INVALID, "sampler not finalized");
    if (!s->music) return fail(DC_ERR_PARAM, "music encoder parameters (music_encoder.*, proj.*) were not supplied");
    if (!d_mel || !d_xf_proj || !d_xf_out || B < 1) return fail(DC_ERR_INVALID, "bad encode_music arguments");
    if (n_mels != 128) return fail(DC_ERR_UNSUPPORTED, "mel spectrograms must have 128 bins (conv4 takes 32 channels x 16 bins), got %d", n_mels);
    // (conv3 / conv4 run on (Tm - 1) / 3 + 1 rows, and reflection padding needs two: the reference's ReflectionPad2d raises below 4 frames)
    if (Tm < 4) return fail(DC_ERR_INVALID, "need at least 4 mel frames (reflection padding of the (Tm - 1) / 3 + 1 rows behind the stride-3 pool), got %d", Tm);
    HIP_TRY(hipSetDevice(s->cfg.device));
    hipStream_t user = (hipStream_t)stream, st = s->stream;
    int rc;
    if ((rc = sync_in(s, user))) return rc;
    std::string err;
    const hipError_t e = dc_music_encode(s->music, d_mel, B, Tm, d_xf_proj, d_xf_out, st, &err);
    if (e != hipSuccess) return fail(DC_ERR_HIP, "encode_music: %s %s", hipGetErrorString(e), err.c_str());
    return sync_out(s, user);
}

// Hat matrix H = A (A^T A)^-1 A^T of the degree-`order` polynomial fit over `window` equally spaced samples, fp64 normal
// equations with positions centred and scaled to [-1, 1] (well conditioned for the window sizes in use).
int dc_savgol_coefficients(int32_t window, int32_t order, float* h_coef) {
    if (window < 3 || !(window & 1) || window > 99 || order < 0 || order >= window || order > 10 || !h_coef)
        return fail(DC_ERR_INVALID, "savgol: need an odd window in [3,99] and 0 <= order < window (order <= 10)");
    const int w = window, n = order + 1, hw = w / 2;
    std::vector<double> A((size_t)w * n), G((size_t)n * n, 0.0), Ginv((size_t)n * n, 0.0);
    for (int i = 0; i < w; ++i) {
        const double u = (double)(i - hw) / hw;
        double pw = 1.0;
        for (int j = 0; j < n; ++j) {
            A[(size_t)i * n + j] = pw;
            pw *= u;
        }
    }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double acc = 0.0;
            for (int i = 0; i < w; ++i) acc += A[(size_t)i * n + a] * A[(size_t)i * n + b];
            G[(size_t)a * n + b] = acc;
        }
    // Gauss-Jordan inverse with partial pivoting (n <= 11)
    std::vector<double> M((size_t)n * 2 * n, 0.0);
    for (int a = 0; a < n; ++a) {
        for (int b = 0; b < n; ++b) M[(size_t)a * 2 * n + b] = G[(size_t)a * n + b];
        M[(size_t)a * 2 * n + n + a] = 1.0;
    }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(M[(size_t)r * 2 * n + c]) > std::fabs(M[(size_t)piv * 2 * n + c])) piv = r;
        if (std::fabs(M[(size_t)piv * 2 * n + c]) < 1e-300) return fail(DC_ERR_INVALID, "savgol: singular normal equations");
        if (piv != c)
            for (int k = 0; k < 2 * n; ++k) std::swap(M[(size_t)piv * 2 * n + k], M[(size_t)c * 2 * n + k]);
        const double d = M[(size_t)c * 2 * n + c];
        for (int k = 0; k < 2 * n; ++k) M[(size_t)c * 2 * n + k] /= d;
        for (int r = 0; r < n; ++r)
            if (r != c) {
                const double f = M[(size_t)r * 2 * n + c];
                if (f != 0.0)
                    for (int k = 0; k < 2 * n; ++k) M[(size_t)r * 2 * n + k] -= f * M[(size_t)c * 2 * n + k];
            }
    }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) Ginv[(size_t)a * n + b] = M[(size_t)a * 2 * n + n + b];
    for (int i = 0; i < w; ++i)
        for (int k = 0; k < w; ++k) {
            double acc = 0.0;
            for (int a = 0; a < n; ++a) {
                double t = 0.0;
                for (int b = 0; b < n; ++b) t += Ginv[(size_t)a * n + b] * A[(size_t)k * n + b];
                acc += A[(size_t)i * n + a] * t;
            }
            h_coef[(size_t)i * w + k] = (float)acc;
        }
    return DC_OK;
}

int dc_sampler_set_smoothing(dc_sampler* s, int32_t window, int32_t order) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (window == 0) {
        s->smooth_window = 0;
        return DC_OK;
    }
    if (window == s->smooth_table_window && order == s->smooth_order && s->d_smooth_coef) {      // the table on the device is this one
        s->smooth_window = window;
        return DC_OK;
    }
    std::vector<float> coef((size_t)(window > 0 ? window : 0) * (window > 0 ? window : 0));
    int rc = dc_savgol_coefficients(window, order, coef.data());
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));          // an earlier loop may still read the old table
    if ((rc = dev_alloc(s, s->d_smooth_coef, coef.size() * 4))) return rc;
    HIP_TRY(hipMemcpy(s->d_smooth_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    s->smooth_window = s->smooth_table_window = window;
    s->smooth_order = order;
    return DC_OK;
}

int dc_savgol_filter(const float* d_in, float* d_out, int32_t B, int32_t T, int32_t P, int32_t window, int32_t order, void* stream) {
    if (!d_in || !d_out || d_in == d_out || B < 1 || P < 1) return fail(DC_ERR_INVALID, "savgol: bad arguments (in-place is not supported)");
    if (T < window) return fail(DC_ERR_INVALID, "savgol: T=%d shorter than the window %d", T, window);
    std::vector<float> coef((size_t)window * window);
    int rc = dc_savgol_coefficients(window, order, coef.data());
    if (rc) return rc;
    float* d_coef = nullptr;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMallocAsync((void**)&d_coef, coef.size() * 4, st));
    HIP_TRY(hipMemcpyAsync(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));          // `coef` is pageable host memory
    HIP_TRY(dc_launch_savgol(st, d_in, d_out, d_coef, B, T, P, window));
    HIP_TRY(hipFreeAsync(d_coef, st));
    return DC_OK;
}

int dc_sampler_denoise(dc_sampler* s, const float* d_x, const int32_t* h_timesteps, float* d_out, void* stream) {
    if (!s || !s->finalized) return fail(DC_ERR_INVALID, "sampler not finalized");
    if (!s->cond_set) return fail(DC_ERR_INVALID, "dc_sampler_set_conditioning must be called first");
    if (!d_x || !h_timesteps || !d_out) return fail(DC_ERR_INVALID, "null pointer argument");
    for (int b = 0; b < s->B; ++b)
        if (h_timesteps[b] < 0 || h_timesteps[b] >= s->cfg.max_timesteps)
            return fail(DC_ERR_INVALID, "timestep %d outside [0,%d)", h_timesteps[b], s->cfg.max_timesteps);
    HIP_TRY(hipSetDevice(s->cfg.device));
    hipStream_t user = (hipStream_t)stream, st = s->stream;
    int rc;
    if ((rc = sync_in(s, user))) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_t_clip, h_timesteps, (size_t)s->B * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = enqueue_step(s, st, false, d_x, d_out, -1, s->precise_forward && can_split_steps(s)))) return rc;
    return sync_out(s, user);
}

int dc_sampler_ddim_loop(dc_sampler* s, const float* d_noise, float* d_out, int32_t num_steps, const float* h_coef,
                         const int32_t* h_snap_iters, int32_t n_snap, float* d_snaps, void* stream) {
    if (s) HIP_TRY(hipSetDevice(s->cfg.device));
    if (n_snap < 0 || (n_snap > 0 && (!h_snap_iters || !d_snaps))) return fail(DC_ERR_INVALID, "bad snapshot arguments");
    const std::vector<float> c8 = widen_coef(h_coef, num_steps);
    return loop_common(s, d_noise, d_out, num_steps, h_coef ? c8.data() : nullptr, h_snap_iters, n_snap, d_snaps, (hipStream_t)stream, false);
}

int dc_sampler_ddim_loop_ex(dc_sampler* s, const float* d_noise, float* d_out, int32_t num_steps, const float* h_coef8,
                            int32_t flags, const float* d_step_noise, const int32_t* h_snap_iters, int32_t n_snap, float* d_snaps,
                            void* stream) {
    if (s) HIP_TRY(hipSetDevice(s->cfg.device));
    if (n_snap < 0 || (n_snap > 0 && (!h_snap_iters || !d_snaps))) return fail(DC_ERR_INVALID, "bad snapshot arguments");
    if (flags & ~(DC_UPD_CLIP | DC_UPD_EPS)) return fail(DC_ERR_INVALID, "unknown update flags 0x%x", flags);
    bool noisy = false;
    if (h_coef8 && num_steps > 0)
        for (int t = 0; t < num_steps; ++t) noisy = noisy || h_coef8[(size_t)DC_COEF * t + 4] != 0.f;
    if (noisy && !d_step_noise && !(s && s->noise_seed_set))
        return fail(DC_ERR_INVALID, "sigma != 0 (eta > 0) needs the per-iteration noise: the tensor d_step_noise [S][B][T][P], or a seed "
                    "(dc_sampler_set_step_noise_seed) for draws generated step by step");
    return loop_common(s, d_noise, d_out, num_steps, h_coef8, h_snap_iters, n_snap, d_snaps, (hipStream_t)stream, false,
                       flags | (noisy ? DC_UPD_NOISY : 0), noisy ? d_step_noise : nullptr);
}

int dc_sampler_set_step_noise_seed(dc_sampler* s, uint64_t seed) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    s->noise_seed = seed;
    s->noise_first = 0;
    s->noise_seed_set = true;
    return DC_OK;
}

int dc_sampler_set_step_noise_seed_at(dc_sampler* s, uint64_t seed, uint64_t first_element) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    s->noise_seed = seed;
    s->noise_first = first_element;
    s->noise_seed_set = true;
    return DC_OK;
}

int dc_step_noise_fill(float* d_out, int64_t n, uint64_t seed, int32_t iteration, void* stream) {
    if (!d_out || n < 0 || iteration < 0) return fail(DC_ERR_INVALID, "bad step-noise arguments");
    if (n == 0) return DC_OK;
    HIP_TRY(dc_launch_step_noise((hipStream_t)stream, d_out, (size_t)n, seed, nullptr, nullptr, iteration, nullptr, 0));
    return DC_OK;
}

int dc_sampler_status(dc_sampler* s, int32_t* h_status, int32_t clear) {
    if (!s || !h_status) return fail(DC_ERR_INVALID, "null argument");
    *h_status = 0;
    if (!s->d_status) return DC_OK;                      // nothing has run yet
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(h_status, s->d_status, 4, hipMemcpyDeviceToHost));
    // a timed-out combine exchange means the launch's workgroups were not co-resident (a shared GPU, or a second sampler's launches
    // on another stream): this sampler's later loops run the form without the exchange (a re-run of the failed loop is then valid)
    if (*h_status & DC_STATUS_SYNC_TIMEOUT) s->l16_own = true;
    if ((*h_status & DC_STATUS_NONFINITE) && !(*h_status & DC_STATUS_SYNC_TIMEOUT) && s->d_E && s->G > 0) {
        // diagnosis (failure path only): was it the fp16 storage of the FiLM tiles?  The range check is not in the production GEMM's
        // epilogue (it measured at 4 % of that kernel); the tiles are scanned here instead - the last step's as they stand, then,
        // while nothing was found, the tiles of every other timestep of the last loop (the GEMM re-run per timestep: the modulation
        // depends on t, and a value that saturates only early in the loop would otherwise read as an operand overflow)
        const size_t ebytes = (size_t)s->G * s->NT * 64 * 32;
        HIP_TRY(dc_launch_scan_f16(s->stream, s->d_E, ebytes, s->d_status));
        HIP_TRY(hipStreamSynchronize(s->stream));
        HIP_TRY(hipMemcpy(h_status, s->d_status, 4, hipMemcpyDeviceToHost));
        if (!(*h_status & DC_STATUS_F16_SAT) && s->cond_set && s->tables_S > 1 && s->d_t_clip) {
            std::vector<int> tc((size_t)s->B);
            for (int i = 0; i + 1 < s->tables_S && !(*h_status & DC_STATUS_F16_SAT); ++i) {
                std::fill(tc.begin(), tc.end(), s->tab_t[i]);
                HIP_TRY(hipMemcpy(s->d_t_clip, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
                if (s->split_film)
                    HIP_TRY(dc_launch_silu_emb(s->stream, s->film_fmt, true, s->d_pp, s->h_model.temb, s->d_t_clip, s->d_s_hi, s->d_s_lo, s->G, s->T, s->B));
                HIP_TRY(dc_launch_film_gemm(s->stream, s->film_fmt, s->split_film, s->h_model.film_w, s->h_model.film_b, s->d_s_hi, s->d_s_lo, s->d_E,
                                            s->G, s->NT, 0, s->NT / 16, s->split_film ? nullptr : s->d_pp, s->h_model.temb, s->d_t_clip, s->T, s->B,
                                            nullptr, nullptr, nullptr, nullptr, s->h_model.film_w16, s->h_model.film_b16, nullptr, s->d_status));
                HIP_TRY(dc_launch_scan_f16(s->stream, s->d_E, ebytes, s->d_status));
                HIP_TRY(hipStreamSynchronize(s->stream));
                HIP_TRY(hipMemcpy(h_status, s->d_status, 4, hipMemcpyDeviceToHost));
            }
        }
    }
    if (clear) HIP_TRY(hipMemset(s->d_status, 0, 4));
    return DC_OK;
}

int dc_sampler_profile_loop(dc_sampler* s, const float* d_noise, float* d_out, int32_t num_steps, const float* h_coef,
                            float* h_ms, int32_t* h_count, int32_t n, void* stream) {
    if (!s || !h_ms || !h_count) return fail(DC_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    s->prof.ev.clear();
    s->prof.ids.clear();
    const std::vector<float> c8 = widen_coef(h_coef, num_steps);
    int rc = loop_common(s, d_noise, d_out, num_steps, h_coef ? c8.data() : nullptr, nullptr, 0, nullptr, (hipStream_t)stream, true);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int i = 0; i < n; ++i) {
        h_ms[i] = 0.f;
        h_count[i] = 0;
    }
    for (size_t i = 0; i < s->prof.ids.size(); ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->prof.ev[2 * i], s->prof.ev[2 * i + 1]));
        const int id = s->prof.ids[i];
        if (id < n) {
            h_ms[id] += ms;
            h_count[id] += 1;
        }
    }
    for (hipEvent_t e : s->prof.ev) hipEventDestroy(e);
    s->prof.ev.clear();
    s->prof.ids.clear();
    return DC_OK;
}

int dc_sampler_debug_denoise(dc_sampler* s, const float* d_x, const int32_t* h_timesteps, float* d_out,
                             int32_t n_layers, int32_t stage, void* stream) {
    if (!s) return fail(DC_ERR_INVALID, "null sampler");
    s->dbg_layers = n_layers;
    s->dbg_stage = stage;
    const int rc = dc_sampler_denoise(s, d_x, h_timesteps, d_out, stream);
    s->dbg_layers = -1;
    s->dbg_stage = 0;
    return rc;
}

int dc_sampler_debug_layer(dc_sampler* s, const float* h_h, const int32_t* h_timesteps, int32_t layer, int32_t first_stage,
                           int32_t stage, void* stream) {
    if (!s || !s->finalized || !s->cond_set) return fail(DC_ERR_INVALID, "sampler not ready (finalize + set_conditioning first)");
    if (!h_h || !h_timesteps) return fail(DC_ERR_INVALID, "null pointer argument");
    if (layer < 0 || layer >= s->cfg.num_layers || first_stage < 1 || stage > 3 || first_stage > stage)
        return fail(DC_ERR_INVALID, "layer / stage out of range (need 1 <= first_stage <= last_stage <= 3)");
    if (s->cfg.no_eff) return fail(DC_ERR_UNSUPPORTED, "dc_sampler_debug_layer covers the linear-attention layers");
    HIP_TRY(hipSetDevice(s->cfg.device));
    // row-major [M][128] -> residual-stream image [G][tile][quarter][64 lanes][4] (dc_kernels.hip load_h)
    const size_t G = (size_t)s->G;
    std::vector<float> img(G * 4 * 4 * 64 * 4, 0.f);
    for (size_t g = 0; g < G; ++g)
        for (int t = 0; t < 4; ++t)
            for (int q = 0; q < 4; ++q)
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 4; ++i) {
                        const size_t tok = g * 32 + (l & 31);             // token space: clip stride s->T; h_h rows: [B][Tx]
                        const size_t bb = tok / s->T, nn = tok % s->T;
                        const int f = 32 * t + tile_row(4 * q + i, l >> 5);
                        if (tok < (size_t)s->M && nn < (size_t)s->Tx)
                            img[(((g * 4 + t) * 4 + q) * 64 + l) * 4 + i] = h_h[(bb * s->Tx + nn) * DC_D + f];
                    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(s->d_h, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    s->dbg_first = layer;
    s->dbg_layers = layer + 1;
    s->dbg_stage = stage | ((first_stage - 1) << 16);
    const int rc = dc_sampler_denoise(s, s->d_x, h_timesteps, s->d_x, stream);     // x is not read on this path; out_mode is never reached
    s->dbg_first = -1;
    s->dbg_layers = -1;
    s->dbg_stage = 0;
    return rc;
}

int dc_sampler_debug_read(dc_sampler* s, const char* what, void* h_out, int64_t nbytes) {
    if (!s || !what || !h_out) return fail(DC_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    const std::string w(what);
    const size_t g = (size_t)s->G;
    const void* src = nullptr;
    size_t have = 0;
    if (w == "h") { src = s->d_h; have = g * 4 * 64 * 64; }
    else if (w == "pp") { src = s->d_pp; have = g * 32 * 64 * 32; }
    else if (w == "s_hi") { src = s->d_s_hi; have = g * 32 * 64 * 16; }
    else if (w == "s_lo") { src = s->d_s_lo; have = g * 32 * 64 * 16; }
    else if (w == "E") { src = s->d_E; have = g * s->NT * 64 * 32; }
    else if (w == "recs") { src = s->d_recs; have = g * 2 * DC_REC_FLOATS * 4; }
    else if (w == "a_sa") { src = s->d_a_sa; have = (size_t)s->B * 16 * 1024; }
    else if (w == "a_ca") { src = s->d_a_ca; have = (size_t)s->cfg.num_layers * s->B * 16 * 1024; }
    else if (w == "stamps") { src = s->d_stamps; have = (8 * 32 + 8 + 1024 + 1024 + 256 + 8 + 512) * 8; }
    else if (w == "temb") { src = s->h_model.temb; have = (size_t)s->cfg.max_timesteps * 512 * 4; }
    else if (w == "full_moves") {      // diagnostic builds only: {visits, moves} of the no_eff key loop's reference point, reset by the read
        if (nbytes != 16) return fail(DC_ERR_INVALID, "full_moves is 16 bytes");
        if (dc_full_moves_read((unsigned long long*)h_out, true) != hipSuccess) return fail(DC_ERR_UNSUPPORTED, "not a -DDC_DIAG_FULL_MOVES build");
        return DC_OK;
    }
    else return fail(DC_ERR_INVALID, "unknown debug buffer '%s'", what);
    if (!src) return fail(DC_ERR_INVALID, "buffer '%s' not allocated yet", what);
    if ((size_t)nbytes > have) return fail(DC_ERR_INVALID, "buffer '%s' holds %zu bytes, asked for %lld", what, have, (long long)nbytes);
    HIP_TRY(hipMemcpy(h_out, src, (size_t)nbytes, hipMemcpyDeviceToHost));
    return DC_OK;
}

int32_t dc_sampler_clip_stride(const dc_sampler* s) { return s ? s->T : 0; }

const char* dc_kernel_name(int32_t id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int32_t dc_kernel_count(void) { return K_COUNT; }
int64_t dc_sampler_workspace_bytes(const dc_sampler* s) { return s ? s->ws_bytes + (int64_t)s->arena_bytes : 0; }

}  // extern "C"
