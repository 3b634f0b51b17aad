"""Parameter layout of the Diffusion_Stage denoiser checkpoint.

The reference saves ``MotionTransformer.state_dict()`` under ``state['encoder']``
(reference: Diffusion_Stage/trainers/ddpm_trainer.py:290-319).  This module lists
every entry (name, shape) so the host side can validate a checkpoint and hand the
tensors to the native library by name.  Layout facts follow
Diffusion_Stage/models/transformer.py:360-445 (MotionTransformer.__init__).

Nothing here touches a GPU.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass


@dataclass(frozen=True)
class DenoiserConfig:
    """Hyper-parameters the sampler path consumes (transformer.py:360-374)."""
    input_feats: int = 26
    num_frames: int = 1800
    latent_dim: int = 128
    ff_size: int = 64
    num_layers: int = 8
    num_heads: int = 8
    music_dim: int = 64          # MusicEncoder output channels (transformer.py:328)
    music_latent_dim: int = 512  # self.linear = nn.Linear(64, 512) (transformer.py:404-405)
    mel_bins: int = 128
    no_eff: bool = False

    @property
    def time_embed_dim(self) -> int:
        return self.latent_dim * 4   # transformer.py:385

    @property
    def head_dim(self) -> int:
        return self.latent_dim // self.num_heads


def _conv_res_layer(prefix, cin, cout, residual_conv):
    """Conv2dResLayer entries (transformer.py:289-311)."""
    e = OrderedDict()
    e[f"{prefix}.conv2d_layer.0.weight"] = (cout, cin, 3, 3)
    e[f"{prefix}.conv2d_layer.0.bias"] = (cout,)
    for s in ("weight", "bias", "running_mean", "running_var"):
        e[f"{prefix}.conv2d_layer.1.{s}"] = (cout,)
    e[f"{prefix}.conv2d_layer.1.num_batches_tracked"] = ()
    if residual_conv:
        e[f"{prefix}.residual.0.weight"] = (cout, cin, 1, 1)
        e[f"{prefix}.residual.0.bias"] = (cout,)
        for s in ("weight", "bias", "running_mean", "running_var"):
            e[f"{prefix}.residual.1.{s}"] = (cout,)
        e[f"{prefix}.residual.1.num_batches_tracked"] = ()
    return e


def _stylization(prefix, D, E):
    """StylizationBlock entries (transformer.py:53-66)."""
    e = OrderedDict()
    e[f"{prefix}.emb_layers.1.weight"] = (2 * D, E)
    e[f"{prefix}.emb_layers.1.bias"] = (2 * D,)
    e[f"{prefix}.norm.weight"] = (D,)
    e[f"{prefix}.norm.bias"] = (D,)
    e[f"{prefix}.out_layers.2.weight"] = (D, D)
    e[f"{prefix}.out_layers.2.bias"] = (D,)
    return e


def param_shapes(cfg: DenoiserConfig = DenoiserConfig()) -> "OrderedDict[str, tuple]":
    """All state_dict entries in the reference's registration order."""
    D, E, L, F = cfg.latent_dim, cfg.time_embed_dim, cfg.music_latent_dim, cfg.ff_size
    e = OrderedDict()
    e["sequence_embedding"] = (cfg.num_frames, D)
    me = "music_encoder"
    e.update(_conv_res_layer(f"{me}.conv1.0", 1, 16, False))
    e.update(_conv_res_layer(f"{me}.conv1.1", 16, 16, False))
    e.update(_conv_res_layer(f"{me}.conv1.2", 16, 16, False))
    e.update(_conv_res_layer(f"{me}.conv2.0", 16, 32, True))
    e.update(_conv_res_layer(f"{me}.conv2.1", 32, 32, False))
    e.update(_conv_res_layer(f"{me}.conv3.0", 32, 32, False))
    e.update(_conv_res_layer(f"{me}.conv3.1", 32, 32, False))
    e[f"{me}.conv4.0.weight"] = (cfg.music_dim, 32 * 16, 1)
    e[f"{me}.conv4.0.bias"] = (cfg.music_dim,)
    for s in ("weight", "bias", "running_mean", "running_var"):
        e[f"{me}.conv4.1.{s}"] = (cfg.music_dim,)
    e[f"{me}.conv4.1.num_batches_tracked"] = ()
    e["linear.weight"] = (L, cfg.music_dim)
    e["linear.bias"] = (L,)
    e["joint_embed.weight"] = (D, cfg.input_feats)
    e["joint_embed.bias"] = (D,)
    e["time_embed.0.weight"] = (E, D)
    e["time_embed.0.bias"] = (E,)
    e["time_embed.2.weight"] = (E, E)
    e["time_embed.2.bias"] = (E,)
    for i in range(cfg.num_layers):
        p = f"temporal_decoder_blocks.{i}"
        e[f"{p}.sa_block.norm.weight"] = (D,)
        e[f"{p}.sa_block.norm.bias"] = (D,)
        for n in ("query", "key", "value"):
            e[f"{p}.sa_block.{n}.weight"] = (D, D)
            e[f"{p}.sa_block.{n}.bias"] = (D,)
        e.update(_stylization(f"{p}.sa_block.proj_out", D, E))
        e[f"{p}.ca_block.norm.weight"] = (D,)
        e[f"{p}.ca_block.norm.bias"] = (D,)
        e[f"{p}.ca_block.text_norm.weight"] = (L,)
        e[f"{p}.ca_block.text_norm.bias"] = (L,)
        e[f"{p}.ca_block.query.weight"] = (D, D)
        e[f"{p}.ca_block.query.bias"] = (D,)
        for n in ("key", "value"):
            e[f"{p}.ca_block.{n}.weight"] = (D, L)
            e[f"{p}.ca_block.{n}.bias"] = (D,)
        e.update(_stylization(f"{p}.ca_block.proj_out", D, E))
        e[f"{p}.ffn.linear1.weight"] = (F, D)
        e[f"{p}.ffn.linear1.bias"] = (F,)
        e[f"{p}.ffn.linear2.weight"] = (D, F)
        e[f"{p}.ffn.linear2.bias"] = (D,)
        e.update(_stylization(f"{p}.ffn.proj_out", D, E))
    e["out.weight"] = (cfg.input_feats, D)
    e["out.bias"] = (cfg.input_feats,)
    e["proj.weight"] = (cfg.music_dim, cfg.music_dim)
    e["proj.bias"] = (cfg.music_dim,)
    return e


def is_float_param(name: str) -> bool:
    return not name.endswith("num_batches_tracked")
