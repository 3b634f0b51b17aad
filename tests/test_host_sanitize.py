"""SURVEY section 5 (race / memory checking): GPU AddressSanitizer is not available on this pool, so the HOST half of the C-ABI
library - parameter store, validation, LayerNorm / FiLM folding, weight packing into the arena, schedule and filter tables - is
built with -fsanitize=address,undefined (host code only, -DDC_HOST_SANITIZE: a sampler can be created without a device) and
driven through its entry points in a child process with the ASan runtime preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "diffusion-conductor_amd")


def test_host_half_under_address_and_ub_sanitizers():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no ASan runtime in this toolchain")
    sys.path.insert(0, ROOT)
    from diffusion_conductor_amd import native
    native.build_library()                         # the other translation units' objects (device code: not instrumented)
    bdir = os.path.join(PKG, "build", "san")
    os.makedirs(bdir, exist_ok=True)
    obj, so = os.path.join(bdir, "dc_api_san.o"), os.path.join(bdir, "libdc_ddim_san.alt")
    src = os.path.join(PKG, "csrc", "dc_api.hip")
    deps = [src] + [os.path.join(PKG, "csrc", h) for h in native.HEADERS] + [os.path.join(ROOT, "include", "dc_ddim.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value",
                        "-DDC_HOST_SANITIZE", "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize-recover=undefined",
                        "-Xarch_host", "-fno-omit-frame-pointer", "-c", src, "-o", obj], check=True)
        others = [os.path.join(PKG, "build", f.replace(".hip", ".o")) for f in native.SOURCES if f != "dc_api.hip"]
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", obj, *others,
                        "-o", so], check=True)
    env = dict(os.environ, DC_DDIM_LIB=so, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ROCR_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    env["LD_LIBRARY_PATH"] = os.path.dirname(rt) + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san_child.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "host sanitize pass: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
