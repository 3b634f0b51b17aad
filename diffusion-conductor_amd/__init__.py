"""MI355X-native DDIM sampler for Diffusion-Conductor's Diffusion_Stage.

Host side (Python, mirrors the reference's call surface) over a C-ABI HIP library
(`csrc/` -> `libdc_ddim.so`, declared in `include/dc_ddim.h`).

    from diffusion_conductor_amd import MotionTransformer, GaussianDiffusion, DDPMTrainer

Importing the package does not touch the GPU or load the shared library.
"""
from .param_spec import DenoiserConfig, param_shapes  # noqa: F401

__version__ = "0.1.0"
__all__ = ["MotionTransformer", "GaussianDiffusion", "DDPMTrainer", "DenoiserConfig", "param_shapes", "evaluate_dataset",
           "smooth_motion"]


def __getattr__(name):   # lazy: torch is only imported when the classes are used
    if name == "MotionTransformer":
        from .denoiser import MotionTransformer
        return MotionTransformer
    if name in ("GaussianDiffusion", "ModelMeanType", "ModelVarType", "LossType", "get_named_beta_schedule"):
        from . import sampler
        return getattr(sampler, name)
    if name == "DDPMTrainer":
        from .harness import DDPMTrainer
        return DDPMTrainer
    if name in ("evaluate_dataset", "smooth_motion", "mse_loss", "list_clips"):
        from . import evaluate
        return getattr(evaluate, name)
    raise AttributeError(name)
