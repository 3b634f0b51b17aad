"""ctypes binding of libdc_ddim.so (include/dc_ddim.h).

There is deliberately no fallback: if the shared library is missing or no gfx950 device
is visible, construction raises.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdc_ddim.so")
CSRC = os.path.join(_HERE, "csrc")

DC_PREC = {"bf16": 0, "mixed": 1, "bf16x3": 2, "fp16": 3}

EXPORTS = [
    "dc_last_error", "dc_version", "dc_linear_beta_schedule", "dc_ddim_coefficients", "dc_pack_weight",
    "dc_sampler_create", "dc_sampler_destroy", "dc_sampler_set_param", "dc_sampler_finalize_params",
    "dc_sampler_set_conditioning", "dc_sampler_encode_music", "dc_sampler_set_encoder_format", "dc_sampler_set_precise_tail", "dc_precise_tail_default", "dc_sampler_set_precise_forward", "dc_sampler_set_combine_exchange", "dc_sampler_set_clip_aligned", "dc_sampler_denoise", "dc_sampler_ddim_loop", "dc_sampler_profile_loop",
    "dc_kernel_name", "dc_kernel_count", "dc_sampler_workspace_bytes", "dc_sampler_clip_stride", "dc_sampler_debug_denoise",
    "dc_sampler_debug_read", "dc_sampler_debug_layer", "dc_savgol_coefficients", "dc_savgol_filter",
    "dc_ddim_coefficients_ex", "dc_sampler_ddim_loop_ex", "dc_sampler_status", "dc_sampler_set_smoothing",
    "dc_sampler_set_step_noise_seed", "dc_sampler_set_step_noise_seed_at", "dc_step_noise_fill",
]

UPDATE_CLIP_DENOISED, UPDATE_EPSILON = 1, 2          # flags of dc_sampler_ddim_loop_ex
STATUS_NONFINITE, STATUS_F16_SATURATED, STATUS_TIMEOUT = 1, 2, 4        # bits of dc_sampler_status


class DcConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_feats", "num_frames", "latent_dim", "ff_size", "num_layers",
                                         "num_heads", "no_eff", "precision", "max_timesteps", "device")]


class DcError(RuntimeError):
    pass


SOURCES = ("dc_kernels.hip", "dc_api.hip", "dc_music.hip", "dc_layer16.hip")
HEADERS = ("dc_common.h", "dc_dev.h", "dc_launch.h", "dc_music.h")
EXTRA_FLAGS = {}      # per-source compiler flags


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libdc_ddim.so (hipcc cross-compiles without a GPU).  Each source becomes an
    object under build/ (compiled in parallel, rebuilt only when it or a header changed), then one link."""
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    hdrs = [os.path.join(CSRC, f) for f in HEADERS if os.path.exists(os.path.join(CSRC, f))] + \
        [os.path.join(os.path.dirname(_HERE), "include", "dc_ddim.h")]
    hmax = max(os.path.getmtime(h) for h in hdrs)
    bdir = os.path.join(_HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value"]
    objs, jobs = [], []
    for src in srcs:
        obj = os.path.join(bdir, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hmax):
            cmd = [hipcc, *flags, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if jobs or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library (never builds implicitly on a GPU box: the prebuilt .so travels)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("DC_DDIM_LIB", LIB_PATH)      # A/B builds of the same ABI side by side (tools/ab.sh); default in-tree
    if not os.path.exists(path):
        raise DcError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback for the sampler)")
    # PyTorch-ROCm ships its own libamdhip64: it must be the copy this process binds (the tensors, streams and the library's
    # kernels have to live in ONE HIP runtime).  Loaded after ours, it becomes a second runtime and the library's own one
    # (/opt/rocm's) sees no device - so torch goes first, also when a caller only wants build() + version().
    import torch  # noqa: F401
    L = C.CDLL(path)
    L.dc_last_error.restype = C.c_char_p
    L.dc_version.restype = C.c_char_p
    L.dc_kernel_name.restype = C.c_char_p
    L.dc_kernel_name.argtypes = [C.c_int32]
    L.dc_kernel_count.restype = C.c_int32
    L.dc_sampler_workspace_bytes.restype = C.c_int64
    L.dc_sampler_workspace_bytes.argtypes = [C.c_void_p]
    L.dc_sampler_clip_stride.restype = C.c_int32
    L.dc_sampler_clip_stride.argtypes = [C.c_void_p]
    dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.dc_linear_beta_schedule.argtypes = [C.c_int32, dp, dp, dp, dp, dp]
    L.dc_ddim_coefficients.argtypes = [C.c_int32, dp, fp]
    L.dc_pack_weight.argtypes = [fp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
    L.dc_sampler_create.argtypes = [C.POINTER(DcConfig), C.POINTER(C.c_void_p)]
    L.dc_sampler_destroy.argtypes = [C.c_void_p]
    L.dc_sampler_destroy.restype = None
    L.dc_sampler_set_param.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int64]
    L.dc_sampler_finalize_params.argtypes = [C.c_void_p]
    L.dc_sampler_set_conditioning.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, ip, C.c_int32, C.c_int32, C.c_void_p]
    L.dc_sampler_encode_music.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
    L.dc_sampler_set_encoder_format.argtypes = [C.c_void_p, C.c_int32]
    L.dc_sampler_set_precise_tail.argtypes = [C.c_void_p, C.c_int32]
    L.dc_sampler_set_combine_exchange.argtypes = [C.c_void_p, C.c_int32]
    L.dc_sampler_set_clip_aligned.argtypes = [C.c_void_p, C.c_int32]
    L.dc_precise_tail_default.argtypes = [C.c_int32]
    L.dc_precise_tail_default.restype = C.c_int32
    L.dc_sampler_set_precise_forward.argtypes = [C.c_void_p, C.c_int32]
    L.dc_sampler_denoise.argtypes = [C.c_void_p, C.c_void_p, ip, C.c_void_p, C.c_void_p]
    L.dc_savgol_coefficients.argtypes = [C.c_int32, C.c_int32, fp]
    L.dc_savgol_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.dc_sampler_ddim_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, fp, ip, C.c_int32, C.c_void_p, C.c_void_p]
    L.dc_ddim_coefficients_ex.argtypes = [C.c_int32, dp, C.c_float, fp]
    L.dc_sampler_ddim_loop_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, fp, C.c_int32, C.c_void_p, ip, C.c_int32,
                                          C.c_void_p, C.c_void_p]
    L.dc_sampler_set_step_noise_seed.argtypes = [C.c_void_p, C.c_uint64]
    L.dc_sampler_set_step_noise_seed_at.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.dc_step_noise_fill.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_int32, C.c_void_p]
    L.dc_sampler_status.argtypes = [C.c_void_p, ip, C.c_int32]
    L.dc_sampler_set_smoothing.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.dc_sampler_debug_denoise.argtypes = [C.c_void_p, C.c_void_p, ip, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.dc_sampler_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    L.dc_sampler_debug_layer.argtypes = [C.c_void_p, fp, ip, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.dc_sampler_profile_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, fp, fp, ip, C.c_int32, C.c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise DcError(f"libdc_ddim error {rc}: {lib().dc_last_error().decode()}")


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


# ---------------------------------------------------------------------------------------
# host-only helpers (work without a GPU)
# ---------------------------------------------------------------------------------------
def linear_beta_schedule(num_steps: int) -> dict:
    """dc_linear_beta_schedule: the fp64 tables of GaussianDiffusion.__init__."""
    n = int(num_steps)
    arrs = [np.empty(n, np.float64) for _ in range(5)]
    dp = C.POINTER(C.c_double)
    _check(lib().dc_linear_beta_schedule(n, *[a.ctypes.data_as(dp) for a in arrs]))
    keys = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")
    return dict(zip(keys, arrs))


def ddim_coefficients(alphas_cumprod: np.ndarray, eta=None) -> np.ndarray:
    """eta None: dc_ddim_coefficients, [S, 4] (eta = 0).  eta a float: dc_ddim_coefficients_ex, [S, 8] with sigma folded in."""
    ac = np.ascontiguousarray(alphas_cumprod, np.float64)
    dp = ac.ctypes.data_as(C.POINTER(C.c_double))
    if eta is None:
        out = np.empty((ac.shape[0], 4), np.float32)
        _check(lib().dc_ddim_coefficients(ac.shape[0], dp, _fptr(out)))
    else:
        out = np.empty((ac.shape[0], 8), np.float32)
        _check(lib().dc_ddim_coefficients_ex(ac.shape[0], dp, C.c_float(float(eta)), _fptr(out)))
    return out


def savgol_coefficients(window: int, order: int) -> np.ndarray:
    """dc_savgol_coefficients: the [window, window] hat matrix (host only)."""
    out = np.empty((window, window), np.float32)
    _check(lib().dc_savgol_coefficients(int(window), int(order), _fptr(out)))
    return out


def savgol_filter(poses, window: int = 19, order: int = 5):
    """dc_savgol_filter on a CUDA(ROCm) fp32 tensor [B, T, P] (or [B, T, J, 2]); returns a new tensor."""
    import torch
    assert poses.is_cuda and poses.dtype == torch.float32
    x = poses.contiguous()
    B, T = x.shape[0], x.shape[1]
    P = x.numel() // (B * T)
    out = torch.empty_like(x)
    _check(lib().dc_savgol_filter(x.data_ptr(), out.data_ptr(), B, T, P, int(window), int(order),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def step_noise(shape, seed: int, iteration: int, device):
    """dc_step_noise_fill: the library's N(0, 1) draws of one DDIM iteration for `seed` as a new fp32 tensor of `shape` on `device`
    (what a seeded eta > 0 loop adds at that iteration)."""
    import torch
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _check(lib().dc_step_noise_fill(out.data_ptr(), out.numel(), C.c_uint64(int(seed) & (2 ** 64 - 1)), int(iteration),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def describe_status(st: int, precision: str) -> str:
    msg = []
    if st & STATUS_NONFINITE:
        msg.append(f"the denoiser produced a non-finite x0 in precision '{precision}' (fp16 operands overflow beyond +-65504): "
                   "use precision='mixed' (bf16-range operands) or precision='auto'")
    if st & STATUS_F16_SATURATED:
        msg.append("a FiLM modulation value left the fp16 range in which every precision mode stores it: this checkpoint is "
                   "outside what the library supports")
    if st & STATUS_TIMEOUT:
        msg.append("a workgroup of the small-batch layer kernel gave up waiting for its clip's combine slices (is the GPU shared with other "
                   "work?): the loop's results are invalid; DC_L16_OWN_COMBINE=1 selects the form without the in-launch exchange")
    return "; ".join(msg) or "ok"


def precise_tail_default(precision: str) -> int:
    """The number of split-operand evaluations at the end of a sampling loop a precision runs unless told otherwise (dc_ddim.h)."""
    return int(lib().dc_precise_tail_default(DC_PREC[precision])) if precision in DC_PREC else 0


def algorithmic_work(n_tokens: int) -> dict:
    """Algorithmic work per launch of the loop's two big kernels at `n_tokens` = B*T, the numerators of
    bench.py's roofline objects.  Both are priced against the MFMA roof, the path's primary bound:
    k_film_gemm is the [6144 x 512] x [512 x tokens] FiLM GEMM; k_layer (one decoder layer for all tokens) carries an eighth of
    the step's remaining algorithmic FLOPs.  `design_bytes`: what THIS decomposition moves through HBM per k_layer launch - per token
    the fp32 residual stream 512 B in + 512 B out, its 24 FiLM tiles x 64 B (written by the GEMM one launch earlier) and the
    workgroup records 72 B out + 36 B in - a property of the design, not SURVEY section 8d's compulsory bytes (2.3 MB per clip-step)."""
    # k_layer's FLOPs: the step's algorithmic 2 * 4 250 112 per token minus the FiLM GEMM's, over 8 layers
    return {"k_layer": {"bound": "mfma", "flops": (2 * 4250112 - 2 * 512 * 6144) // 8 * n_tokens,
                        "design_bytes": (512 + 512 + 24 * 64 + 72 + 36) * n_tokens},
            "k_film_gemm": {"bound": "mfma", "flops": 2 * 512 * 6144 * n_tokens}}


def pack_weight(w: np.ndarray, chained: bool):
    w = np.ascontiguousarray(w, np.float32)
    n_out, k_in = w.shape
    ne = ((n_out + 31) // 32) * ((k_in + 31) // 32) * 2 * 64 * 8
    hi, lo = np.empty(ne, np.uint16), np.empty(ne, np.uint16)
    u16 = C.POINTER(C.c_uint16)
    _check(lib().dc_pack_weight(_fptr(w), n_out, k_in, int(chained), hi.ctypes.data_as(u16), lo.ctypes.data_as(u16)))
    return hi, lo


def tile_row(r, hh):
    return (r & 3) + 8 * (r >> 2) + 4 * hh


def unpack_ft(raw):
    """[G][NT][64 lanes][16 regs] FT tiles -> row-major [32*G tokens][32*NT features]."""
    G, NT = raw.shape[0], raw.shape[1]
    out = np.empty((G, 32, NT, 32), raw.dtype)
    lane = np.arange(64)
    for r in range(16):
        feat = tile_row(r, lane >> 5)
        out[:, lane & 31, :, feat] = raw[:, :, lane, r].transpose(2, 0, 1)
    return out.reshape(G * 32, NT * 32)


# ---------------------------------------------------------------------------------------
# the sampler object
# ---------------------------------------------------------------------------------------
class NativeSampler:
    """Owns one dc_sampler.  Tensors are torch CUDA(ROCm) tensors; only their data_ptr()
    crosses the ABI."""

    def __init__(self, cfg, precision="fp16", max_timesteps=1000, device=0):
        self._h = C.c_void_p()
        c = DcConfig(cfg.input_feats, cfg.num_frames, cfg.latent_dim, cfg.ff_size, cfg.num_layers, cfg.num_heads,
                     int(bool(cfg.no_eff)), DC_PREC[precision], int(max_timesteps), int(device))
        _check(lib().dc_sampler_create(C.byref(c), C.byref(self._h)))
        self.cfg, self.precision, self.device, self.max_timesteps = cfg, precision, int(device), int(max_timesteps)
        self.B = self.T = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().dc_sampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, state_dict):
        """state_dict: name -> torch tensor / ndarray, with the reference's keys."""
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            if a.dtype.kind != "f":
                continue   # BatchNorm.num_batches_tracked
            a = np.ascontiguousarray(a, np.float32)
            _check(lib().dc_sampler_set_param(self._h, k.encode(), _fptr(a), a.size))
        _check(lib().dc_sampler_finalize_params(self._h))

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def set_conditioning(self, xf_proj, xf_out, length=None):
        import torch
        assert xf_proj.is_cuda and xf_proj.dtype == torch.float32 and xf_proj.is_contiguous()
        assert xf_out.is_cuda and xf_out.dtype == torch.float32 and xf_out.is_contiguous()
        B, T, Cc = xf_proj.shape
        assert Cc == 64 and tuple(xf_out.shape) == (B, T, 64)
        lp = None
        if length is not None:
            la = np.ascontiguousarray(np.asarray(length.cpu() if hasattr(length, "cpu") else length), np.int32)
            assert la.shape == (B,)
            lp = _iptr(la)
        _check(lib().dc_sampler_set_conditioning(self._h, xf_proj.data_ptr(), xf_out.data_ptr(), lp, B, T, self._stream()))
        self.B, self.T = B, T

    def set_precise_tail(self, steps):
        """The loop's last `steps` model evaluations on split operands (fp16: golden DDIM-50 5.0e-4 -> 2.3e-4 / 1.6e-4 with 1 / 2 steps at
        +0.45 % of the loop each, default 1; bf16: see DESIGN.md section 5; -1: the precision's default).  DC_PRECISE_TAIL=k in the environment overrides it."""
        _check(lib().dc_sampler_set_precise_tail(self._h, int(steps)))

    def set_clip_aligned(self, mode):
        """Units of the wide launch form: True clip-aligned (batch-invariant results), False flat (throughput), None the library's rule -
        aligned whenever that costs no extra round of workgroups (dc_ddim.h, dc_sampler_set_clip_aligned)."""
        _check(lib().dc_sampler_set_clip_aligned(self._h, -1 if mode is None else (1 if mode else 0)))

    def set_precise_forward(self, on):
        """denoise() on split operands (the precise tail's evaluation form; dc_ddim.h, dc_sampler_set_precise_forward)."""
        _check(lib().dc_sampler_set_precise_forward(self._h, 1 if on else 0))

    def set_combine_exchange(self, on):
        """Small batches: whether a clip's workgroups exchange their combine slices inside a layer launch (default) or every workgroup
        combines alone (no in-launch wait; dc_ddim.h, dc_sampler_set_combine_exchange).  dc_sampler_status latches `off` after a timeout."""
        _check(lib().dc_sampler_set_combine_exchange(self._h, 1 if on else 0))

    def set_encoder_format(self, fmt):
        """MusicEncoder activation format: "split" (two bf16 planes, 6e-6 at the encoder's output) or "f16" (one fp16 plane, 3.7e-4, half
        the time; the default beside an fp16 / bf16 denoiser).  DC_ME_PREC=f16|split in the environment overrides it per call."""
        _check(lib().dc_sampler_set_encoder_format(self._h, {"split": 0, "f16": 1, "fp16": 1}[fmt]))

    def encode_music(self, mel, out=None):
        """mel fp32 [B, Tm, 128] on the device -> (xf_proj, xf_out), each [B, (Tm-1)//3+1, 64] (`out`: the pair to fill, e.g.
        batch slices of larger tensors)."""
        import torch
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.dim() == 3
        B, Tm, nm = mel.shape
        T = (Tm - 1) // 3 + 1
        if out is None:
            xf_proj = torch.empty((B, T, 64), dtype=torch.float32, device=mel.device)
            xf_out = torch.empty_like(xf_proj)
        else:
            xf_proj, xf_out = out
            for t in out:
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (B, T, 64)
        _check(lib().dc_sampler_encode_music(self._h, mel.data_ptr(), B, Tm, nm, xf_proj.data_ptr(), xf_out.data_ptr(),
                                             self._stream()))
        return xf_proj, xf_out

    def denoise(self, x, timesteps):
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert tuple(x.shape) == (self.B, self.T, self.cfg.input_feats)
        ta = np.ascontiguousarray(np.asarray(timesteps.cpu() if hasattr(timesteps, "cpu") else timesteps), np.int32)
        assert ta.shape == (self.B,)
        out = torch.empty_like(x)
        _check(lib().dc_sampler_denoise(self._h, x.data_ptr(), _iptr(ta), out.data_ptr(), self._stream()))
        return out

    def ddim_loop(self, noise, coef, snap_iters=(), flags=0, step_noise=None, noise_seed=None):
        """coef [S, 4] (dc_sampler_ddim_loop: START_X, no clipping, eta = 0) or [S, 8] (dc_sampler_ddim_loop_ex with `flags` =
        UPDATE_* and, when any sigma != 0, either step_noise [S, B, T, P] on the device or noise_seed: the library then generates
        each iteration's draws at the head of its step, dc_sampler_set_step_noise_seed; a pair (seed, first_element) places this
        sampler's clips inside a larger batch's draw, dc_sampler_set_step_noise_seed_at).  A seed serves one loop."""
        import torch
        assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
        assert tuple(noise.shape) == (self.B, self.T, self.cfg.input_feats)
        coef = np.ascontiguousarray(coef, np.float32)
        S = coef.shape[0]
        out = torch.empty_like(noise)
        si = np.ascontiguousarray(np.asarray(list(snap_iters), np.int32))
        snaps = torch.empty((len(si),) + tuple(noise.shape), dtype=torch.float32, device=noise.device) if len(si) else None
        sip, snp = (_iptr(si) if len(si) else None), (snaps.data_ptr() if snaps is not None else None)
        if coef.shape[1] == 4:
            assert flags == 0 and step_noise is None, "clip / epsilon / eta > 0 need the [S, 8] table of ddim_coefficients(.., eta)"
            _check(lib().dc_sampler_ddim_loop(self._h, noise.data_ptr(), out.data_ptr(), S, _fptr(coef), sip, len(si), snp, self._stream()))
        else:
            assert coef.shape[1] == 8
            zp = None
            if step_noise is not None:
                assert step_noise.is_cuda and step_noise.dtype == torch.float32 and step_noise.is_contiguous()
                assert tuple(step_noise.shape) == (S,) + tuple(noise.shape)
                zp = step_noise.data_ptr()
            elif noise_seed is not None:
                seed, first = noise_seed if isinstance(noise_seed, (tuple, list)) else (noise_seed, 0)
                _check(lib().dc_sampler_set_step_noise_seed_at(self._h, C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint64(int(first))))
            _check(lib().dc_sampler_ddim_loop_ex(self._h, noise.data_ptr(), out.data_ptr(), S, _fptr(coef), int(flags), zp, sip,
                                                 len(si), snp, self._stream()))
        return out, snaps

    def set_smoothing(self, window=19, order=5):
        """dc_sampler_set_smoothing: the loops write Savitzky-Golay-smoothed poses from now on (window 0: off)."""
        _check(lib().dc_sampler_set_smoothing(self._h, int(window), int(order)))

    def status(self, clear=True):
        """dc_sampler_status: waits for the sampler's work; OR of STATUS_NONFINITE / STATUS_F16_SATURATED."""
        v = C.c_int32(0)
        _check(lib().dc_sampler_status(self._h, C.byref(v), int(bool(clear))))
        return int(v.value)

    def profile_loop(self, noise, coef):
        import torch
        coef = np.ascontiguousarray(coef, np.float32)
        n = lib().dc_kernel_count()
        ms, cnt = np.zeros(n, np.float32), np.zeros(n, np.int32)
        out = torch.empty_like(noise)
        _check(lib().dc_sampler_profile_loop(self._h, noise.data_ptr(), out.data_ptr(), coef.shape[0], _fptr(coef),
                                             _fptr(ms), _iptr(cnt), n, self._stream()))
        names = [lib().dc_kernel_name(i).decode() for i in range(n)]
        return {names[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}, out

    # ---- test hooks ---------------------------------------------------------------------
    def debug_denoise(self, x, timesteps, n_layers, stage):
        import torch
        ta = np.ascontiguousarray(np.asarray(timesteps), np.int32)
        out = torch.zeros_like(x)
        _check(lib().dc_sampler_debug_denoise(self._h, x.data_ptr(), _iptr(ta), out.data_ptr(), n_layers, stage,
                                              self._stream()))
        return out

    def debug_layer(self, h, timesteps, layer, first_stage, last_stage):
        """Blocks first_stage..last_stage (1 = SA, 2 = CA, 3 = FFN) of decoder layer `layer` alone on the residual stream
        h [B,T,128] (host array); returns h' [B,T,128]."""
        ha = np.ascontiguousarray(np.asarray(h, np.float32).reshape(self.B * self.T, 128))
        ta = np.ascontiguousarray(np.asarray(timesteps), np.int32)
        _check(lib().dc_sampler_debug_layer(self._h, _fptr(ha), _iptr(ta), int(layer), int(first_stage), int(last_stage),
                                            self._stream()))
        return self.read_h()[:self.B * self.T].reshape(self.B, self.T, 128)

    def debug_read(self, what, dtype, count):
        a = np.empty(count, dtype)
        _check(lib().dc_sampler_debug_read(self._h, what.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def clip_stride(self):
        """Frames per clip of the internal token space (T, padded to whole 32-frame groups where the clip-aligned kernels run)."""
        return int(lib().dc_sampler_clip_stride(self._h))

    def read_h(self):
        """Residual stream [>= B*T, 128] (rows b*T + n) unpacked from the FT-tile image [G][4][64][16]."""
        Tp = self.clip_stride()
        G = (self.B * Tp + 31) // 32
        # device order [G][tile][quarter][lane][4] -> [G][tile][lane][16 regs]
        raw = self.debug_read("h", np.float32, G * 4 * 64 * 16).reshape(G, 4, 4, 64, 4)
        rows = unpack_ft(raw.transpose(0, 1, 3, 2, 4).reshape(G, 4, 64, 16))
        if Tp != self.T:       # drop the padding frames of every clip
            rows = rows[:self.B * Tp].reshape(self.B, Tp, 128)[:, :self.T].reshape(self.B * self.T, 128)
        return rows

    def workspace_bytes(self):
        return int(lib().dc_sampler_workspace_bytes(self._h))
