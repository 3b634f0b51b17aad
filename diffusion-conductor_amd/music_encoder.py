"""MusicEncoder forward on PyTorch-ROCm ops (MIOpen convolutions), eval mode.

Follows Diffusion_Stage/models/transformer.py:289-340 (Conv2dResLayer, MusicEncoder.forward).
This is the one piece of the sampling path still on library ops in this round; it runs once
per clip before the DDIM loop (SURVEY.md section 8f item 1 schedules its HIP version next).
"""
import torch
import torch.nn.functional as F


def _sub(node, path):
    for p in path.split("."):
        node = node._modules[p]
    return node


def _bn(x, bn, eps=1e-5):
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, eps)


def _res_layer(layer, x, residual):
    conv, bn = _sub(layer, "conv2d_layer.0"), _sub(layer, "conv2d_layer.1")
    y = F.relu(_bn(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), conv.weight, conv.bias), bn))
    if residual == "none":
        return y
    if residual == "identity":
        return y + x
    rc, rb = _sub(layer, "residual.0"), _sub(layer, "residual.1")
    return y + _bn(F.conv2d(x, rc.weight, rc.bias), rb)


def music_encoder_forward(model, mel):
    """mel [B,Tm,128] -> [B,Tm/3,64]."""
    me = model._modules["music_encoder"]
    x = mel.unsqueeze(1)
    x = _res_layer(_sub(me, "conv1.0"), x, "none")
    x = _res_layer(_sub(me, "conv1.1"), x, "identity")
    x = _res_layer(_sub(me, "conv1.2"), x, "identity")
    x = F.max_pool2d(x, (5, 5), (1, 2), (2, 2))
    x = _res_layer(_sub(me, "conv2.0"), x, "conv")
    x = _res_layer(_sub(me, "conv2.1"), x, "identity")
    x = F.max_pool2d(x, (5, 5), (3, 2), (2, 2))
    x = _res_layer(_sub(me, "conv3.0"), x, "identity")
    x = _res_layer(_sub(me, "conv3.1"), x, "identity")
    x = F.max_pool2d(x, (3, 3), (1, 2), (1, 1))
    x = x.transpose(1, 2).flatten(start_dim=2).transpose(1, 2)
    c4, b4 = _sub(me, "conv4.0"), _sub(me, "conv4.1")
    x = _bn(F.conv1d(x, c4.weight, c4.bias), b4)
    return x.transpose(1, 2).contiguous()
