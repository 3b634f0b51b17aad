"""Child of tests/test_host_sanitize.py: drives the HOST half of libdc_ddim (built with -fsanitize=address,undefined -DDC_HOST_SANITIZE,
loaded from DC_DDIM_LIB with the ASan runtime preloaded) through raw ctypes - no torch, no device.  Any sanitizer report aborts
the process (halt_on_error / -fno-sanitize-recover), so exit code 0 means a clean pass."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion_conductor_amd.synthetic import synthetic_state_dict  # noqa: E402  (numpy only)

L = C.CDLL(os.environ["DC_DDIM_LIB"])
L.dc_last_error.restype = C.c_char_p
L.dc_version.restype = C.c_char_p
L.dc_kernel_name.restype = C.c_char_p
dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_feats", "num_frames", "latent_dim", "ff_size", "num_layers", "num_heads", "no_eff",
                                         "precision", "max_timesteps", "device")]


def err():
    return L.dc_last_error().decode()


assert b"gfx950" in L.dc_version()
# ---- schedule / coefficient tables, Savitzky-Golay hat matrix, weight packing (pure host functions)
for n in (1, 21, 50, 1000):
    a = [np.empty(n) for _ in range(5)]
    assert L.dc_linear_beta_schedule(n, *[x.ctypes.data_as(dp) for x in a]) == 0
    c4, c8 = np.empty((n, 4), np.float32), np.empty((n, 8), np.float32)
    assert L.dc_ddim_coefficients(n, a[1].ctypes.data_as(dp), c4.ctypes.data_as(fp)) == 0
    L.dc_ddim_coefficients_ex.argtypes = [C.c_int32, dp, C.c_float, fp]
    assert L.dc_ddim_coefficients_ex(n, a[1].ctypes.data_as(dp), 0.5, c8.ctypes.data_as(fp)) == 0
assert L.dc_linear_beta_schedule(0, None, None, None, None, None) != 0
for w, o in ((19, 5), (5, 2), (51, 5)):
    h = np.empty((w, w), np.float32)
    assert L.dc_savgol_coefficients(w, o, h.ctypes.data_as(fp)) == 0
h = np.empty((18, 18), np.float32)
assert L.dc_savgol_coefficients(18, 5, h.ctypes.data_as(fp)) != 0 and L.dc_savgol_coefficients(5, 7, h.ctypes.data_as(fp)) != 0
rng = np.random.default_rng(0)
for (no, ki, ch) in ((128, 128, 1), (26, 128, 1), (128, 26, 1), (6144, 512, 0), (64, 128, 1), (1, 1, 0), (33, 47, 1)):
    wm = rng.standard_normal((no, ki)).astype(np.float32)
    ne = ((no + 31) // 32) * ((ki + 31) // 32) * 2 * 64 * 8
    hi, lo = np.empty(ne, np.uint16), np.empty(ne, np.uint16)
    u16 = C.POINTER(C.c_uint16)
    assert L.dc_pack_weight(wm.ctypes.data_as(fp), no, ki, ch, hi.ctypes.data_as(u16), lo.ctypes.data_as(u16)) == 0
for i in range(-1, L.dc_kernel_count() + 1):
    L.dc_kernel_name(i)

# ---- sampler host state: create (validation), parameters (sizes, unknown keys, missing ones), finalize (folding + packing of
# every image into the arena, all four precision modes and no_eff), destroy - twice over, and re-finalisation after a parameter change
bad = Cfg(26, 1800, 64, 64, 8, 8, 0, 3, 1000, 0)
hnd = C.c_void_p()
assert L.dc_sampler_create(C.byref(bad), C.byref(hnd)) != 0 and "latent_dim" in err()
for kw in (dict(input_feats=40), dict(num_layers=17), dict(precision=9), dict(max_timesteps=0), dict(no_eff=1, precision=1)):
    c = Cfg(26, 1800, 128, 64, 8, 8, 0, 3, 1000, 0)
    for k, v in kw.items():
        setattr(c, k, v)
    assert L.dc_sampler_create(C.byref(c), C.byref(hnd)) != 0, kw
assert L.dc_sampler_create(None, C.byref(hnd)) != 0
sd = {k: np.ascontiguousarray(v, np.float32) for k, v in synthetic_state_dict().items()}
L.dc_sampler_set_param.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int64]
L.dc_sampler_destroy.argtypes = [C.c_void_p]
L.dc_sampler_destroy.restype = None
L.dc_sampler_finalize_params.argtypes = [C.c_void_p]
for prec, no_eff, layers in ((3, 0, 8), (1, 0, 8), (2, 0, 8), (0, 0, 8), (3, 1, 8), (3, 0, 1)):
    c = Cfg(26, 1800, 128, 64, layers, 8, no_eff, prec, 1000, 0)
    hnd = C.c_void_p()
    assert L.dc_sampler_create(C.byref(c), C.byref(hnd)) == 0, err()      # host-only: no device in this process
    assert L.dc_sampler_finalize_params(hnd) != 0 and "missing parameter" in err()
    assert L.dc_sampler_set_param(hnd, b"no.such.key", sd["out.bias"].ctypes.data_as(fp), 26) != 0 and "unknown parameter" in err()
    assert L.dc_sampler_set_param(hnd, b"out.bias", sd["out.bias"].ctypes.data_as(fp), 25) != 0 and "elements" in err()
    mk = next(k for k in sd if k.startswith("music_encoder.") and k.endswith("conv2d_layer.0.weight"))
    assert L.dc_sampler_set_param(hnd, mk.encode(), sd["out.bias"].ctypes.data_as(fp), 3) != 0 and "elements" in err()
    assert L.dc_sampler_set_param(hnd, b"music_encoder.not.consumed", sd["out.bias"].ctypes.data_as(fp), 3) == 0     # accepted and dropped
    assert L.dc_sampler_set_param(hnd, None, None, 0) != 0
    for k, v in sd.items():
        if k.startswith("temporal_decoder_blocks."):
            if int(k.split(".")[1]) >= layers:
                continue
        rc = L.dc_sampler_set_param(hnd, k.encode(), v.ctypes.data_as(fp), v.size)
        assert rc == 0, (k, err())
    assert L.dc_sampler_finalize_params(hnd) == 0, err()
    # a changed parameter un-finalises; finalising again rebuilds the arena (the old one is freed)
    v = sd["out.weight"] * 2
    assert L.dc_sampler_set_param(hnd, b"out.weight", v.ctypes.data_as(fp), v.size) == 0
    assert L.dc_sampler_finalize_params(hnd) == 0, err()
    # argument validation of the entry points that need the device afterwards: they must fail, never touch memory
    L.dc_sampler_set_conditioning.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p]
    dummy = C.c_void_p(64)
    assert L.dc_sampler_set_conditioning(hnd, None, None, None, 1, 64, None) != 0
    assert L.dc_sampler_set_conditioning(hnd, dummy, dummy, None, 0, 64, None) != 0
    assert L.dc_sampler_set_conditioning(hnd, dummy, dummy, None, 1, 16, None) != 0
    assert L.dc_sampler_set_conditioning(hnd, dummy, dummy, None, 1, 5000, None) != 0
    assert L.dc_sampler_set_conditioning(hnd, dummy, dummy, None, 1, 64, None) != 0 and "host-only" in err()
    L.dc_sampler_ddim_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, fp, C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p]
    c4 = np.zeros((50, 4), np.float32)
    assert L.dc_sampler_ddim_loop(hnd, dummy, dummy, 50, c4.ctypes.data_as(fp), None, 0, None, None) != 0
    assert L.dc_sampler_ddim_loop(hnd, dummy, dummy, 50, c4.ctypes.data_as(fp), None, -1, None, None) != 0
    L.dc_sampler_set_step_noise_seed_at.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    assert L.dc_sampler_set_step_noise_seed_at(hnd, 7, 12345) == 0 and L.dc_sampler_set_step_noise_seed_at(None, 7, 0) != 0
    L.dc_sampler_set_smoothing.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    assert L.dc_sampler_set_smoothing(hnd, 18, 5) != 0
    st = C.c_int32(5)
    L.dc_sampler_status.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
    assert L.dc_sampler_status(hnd, C.byref(st), 1) == 0 and st.value == 0          # nothing has run
    L.dc_sampler_workspace_bytes.restype = C.c_int64
    L.dc_sampler_workspace_bytes.argtypes = [C.c_void_p]
    assert L.dc_sampler_workspace_bytes(hnd) > 1 << 20
    L.dc_sampler_destroy(hnd)
L.dc_sampler_destroy(None)
print("host sanitize pass: ok")
