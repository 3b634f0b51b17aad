"""Drop-in for the sampling half of the reference's ``DDPMTrainer``.

Mirrors Diffusion_Stage/trainers/ddpm_trainer.py: constructor (:82-108, minus the training-only
MotionPretrain/ST-GCN and mmcv imports), ``load`` (:303-319), ``eval_mode``, ``to`` and
``generate_music_motion`` (:183-201).  Extensions over the reference: ``music_mel`` may be a
batch ``[B,5400,128]``; ``noise=`` / ``seed=`` make sampling reproducible; under an initialised
``torch.distributed`` group the clips are sharded over ranks (sharding.py).
"""
from __future__ import annotations

import numpy as np
import torch

from .sampler import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule
from .sharding import dist_info, sharded_sample


class DDPMTrainer(object):
    def __init__(self, args, encoder):
        self.opt = args
        self.device = args.device
        self.encoder = encoder
        self.diffusion_steps = args.diffusion_steps
        betas = get_named_beta_schedule("linear", self.diffusion_steps)
        self.diffusion = GaussianDiffusion(betas=betas, model_mean_type=ModelMeanType.START_X,
                                           model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
        if getattr(args, "is_train", False):
            raise NotImplementedError("this package covers the sampling path only")
        self.to(self.device)

    def to(self, device):
        self.encoder.to(device)

    def eval_mode(self):
        self.encoder.eval()

    def load(self, model_dir):
        """ddpm_trainer.py:303-319 (inference branch): checkpoint['encoder'] with strict=False."""
        checkpoint = torch.load(model_dir, map_location="cpu")
        self.encoder.load_state_dict(checkpoint["encoder"], strict=False)
        return checkpoint.get("ep", 0), checkpoint.get("total_it", 0)

    def _sample_local(self, mel, noise, dim_pose, idxs, smooth=None):
        xf_proj, xf_out = self.encoder.encode_music(mel, self.device)
        B, T = mel.shape[0], xf_proj.shape[1]
        try:
            return self.diffusion.ddim_sample_loop(
                self.encoder, (B, T, dim_pose), noise=noise, clip_denoised=False, progress=False,
                model_kwargs={"xf_proj": xf_proj, "xf_out": xf_out,
                              "length": torch.LongTensor([T] * B)},
                idxs=idxs, smooth=smooth)
        except FloatingPointError:
            # The loop's numeric check failed.  If the music features themselves are not finite, the fp16-plane MusicEncoder
            # overflowed (an activation beyond 65504; the reference's mel is normalised to [0, 1], so this takes unusual input):
            # encode once more on the split bf16 planes (fp32 range) and sample again.  Anything else is the caller's to see.
            if getattr(self.encoder, "encoder_format", "split") == "split" or bool(torch.isfinite(xf_out).all()):
                raise
            self.encoder.encoder_format = "split"
            nat = getattr(self.encoder, "_native", None)
            if nat is not None:
                nat.set_encoder_format("split")
            return self._sample_local(mel, noise, dim_pose, idxs, smooth)

    def generate_music_motion(self, music_mel, dim_pose, batch_size=1024, idxs=[], noise=None, seed=None, smooth=None):
        """music_mel: np.ndarray/tensor [5400,128] (reference) or [B,5400,128] -> tensor [B,1800,dim_pose].
        smooth: None, or the Savitzky-Golay kernel size (order 5) tools/visualization.py:126 smooths the keypoints with - applied
        by the sampling loop's final write."""
        mel = torch.as_tensor(np.asarray(music_mel) if not torch.is_tensor(music_mel) else music_mel)
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        # A pinned fp32 host batch stays on the host: encode_music copies it in chunks beside the encoder (denoiser.py,
        # _encode_music_pipelined), and a rank of a sharded run copies only its own clips.  Everything else (pageable host
        # memory - the driver stages it through its own pinned buffers anyway -, other dtypes, device tensors) goes to the
        # device here, as ddpm_trainer.py:185 does.
        if not (not mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous() and mel.is_pinned()):
            mel = mel.to(self.device, dtype=torch.float32)
        B, T = mel.shape[0], (mel.shape[1] - 1) // 3 + 1       # frames encode_music produces (MusicEncoder pools time by 3)
        _, world = dist_info()
        grouped = torch.distributed.is_available() and torch.distributed.is_initialized()     # a world-size-1 group still gathers
        if noise is None and (seed is not None or world > 1):
            # x_T for the WHOLE batch from one generator, sliced per shard: with every rank drawing from its own default
            # generator, identically seeded ranks would hand all shards the same noise rows.  Without a seed, rank 0's draw
            # is broadcast.
            g = torch.Generator()
            if seed is not None:
                g.manual_seed(int(seed))
            else:
                s0 = torch.tensor([g.seed() & 0x7fffffffffffffff], dtype=torch.int64, device=self.device if world > 1 else "cpu")
                if world > 1:
                    import torch.distributed as dist
                    dist.broadcast(s0, src=0)
                g.manual_seed(int(s0.item()))
            noise = torch.randn(B, T, dim_pose, generator=g)
        if noise is not None:
            noise = torch.as_tensor(noise).to(self.device, dtype=torch.float32)
        sm = (int(smooth), 5) if smooth else None
        with torch.no_grad():
            if not grouped or len(idxs):
                return self._sample_local(mel, noise, dim_pose, idxs, sm)
            return sharded_sample(lambda m, n: self._sample_local(m, n, dim_pose, [], sm), mel, noise, out_shape=(T, dim_pose),
                                  device=self.device)
