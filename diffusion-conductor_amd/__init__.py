"""MI355X-native DDIM sampler for Diffusion-Conductor's Diffusion_Stage.

Host side (Python, mirrors the reference's call surface) over a C-ABI HIP library
(`csrc/` -> `libdc_ddim.so`, declared in `include/dc_ddim.h`).
"""
from .param_spec import DenoiserConfig, param_shapes  # noqa: F401

__version__ = "0.1.0"
