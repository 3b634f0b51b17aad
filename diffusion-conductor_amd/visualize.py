"""Drop-in for the sampling half of the reference's ``tools/visualization.py`` command line.

    python -m diffusion_conductor_amd.visualize --opt_path <.../opt.txt> --music_path <mel.npy | dir of *.npy> \
        --npy_path out.npy --gpu_id 0 [--smooth] [--model latest.tar] [--seed 0]

Same flags and call sequence as Diffusion_Stage/tools/visualization.py:180-223: ``get_opt`` (this package's reader of the
training run's ``opt.txt``; the reference's is utils/get_opt.py:29-105) -> ``build_models`` (:169-178) -> ``DDPMTrainer`` ->
``trainer.load(<model_dir>/latest.tar)`` -> ``eval_mode`` -> ``generate_music_motion(mel, opt.dim_pose)`` -> reshape
``[T,13,2]`` (:217-218) -> (``smooth_motion(kernel=19)``, :126, with ``--smooth``) -> ``np.save(npy_path)``.

What stays out (rendering is out of scope; the image has no librosa / cv2 / moviepy): the mp3 -> mel
extraction (``extract_mel_feature`` :152-167) and the video rendering.  ``--music_path`` therefore takes the mel spectrogram
itself: a ``.npy`` of shape ``[5400,128]`` (what ``extract_mel_feature`` returns and the dataset stores as ``mel.npy``), or
a directory of such files, which are sampled as ONE batch.  The reference parses ``--npy_path`` but never writes it
(``plot_music2motion`` :143 ignores the argument); here it receives the keypoints: ``[T,13,2]`` for one clip,
``[B,T,13,2]`` for a directory.
"""
from __future__ import annotations

import argparse
import os
import re
from argparse import Namespace
from os.path import join as pjoin

import numpy as np

# opt.txt is what train.py writes (options/base_options.py:79-89): one "key: value" line per option between two banner lines.
# A value is a bool ("True" / "False"), a signed integer, a signed decimal with digits on both sides of the point, or text -
# anything else (a list, scientific notation) stays text, which is how the reference's reader (utils/get_opt.py:8-26, 37-48)
# types them too.
_LINE = re.compile(r"^\s*([^:\s][^:]*?)\s*:\s(.*?)\s*$")
_INT = re.compile(r"^[+-]?\d+$")
_DEC = re.compile(r"^[+-]?\d+\.\d+$")

# per dataset: (joints, frames per clip).  Only the conducting-motion dataset has a sampler here.
_DATASETS = {"ConductorMotion100": (13, 1800)}

# options the sampler reads, with the value an opt.txt that lacks them implies (utils/get_opt.py:50-59)
_SAMPLER_DEFAULTS = {"num_layers": 8, "latent_dim": 512, "diffusion_steps": 1000, "no_clip": False, "no_eff": False}


def _literal(text):
    if text in ("True", "False"):
        return text == "True"
    if _DEC.match(text):
        return float(text)
    if _INT.match(text):
        return int(text)
    return text


def get_opt(opt_path, device):
    """Reads a training run's ``opt.txt`` into the options object the sampling entry point needs (the role of
    utils/get_opt.py:29-105).  Every ``key: value`` line becomes an attribute (typed by `_literal`); on top of that the fields
    the sampler consumes are filled in: the denoiser's size (``num_layers``, ``latent_dim``, ``no_eff``, ``no_clip``),
    ``diffusion_steps``, the pose geometry of the dataset (``joints_num``, ``dim_pose``, ``max_motion_length``) and where the
    checkpoint lives (``model_dir`` = <checkpoints_dir>/<dataset_name>/<name>/model, trainers load ``latest.tar`` from it);
    inference is forced (``is_train`` / ``is_continue`` False, ``which_epoch`` "latest")."""
    print("Reading", opt_path)
    opt = Namespace(**_SAMPLER_DEFAULTS)
    with open(opt_path) as f:
        for raw in f:
            m = _LINE.match(raw)
            if m is None or raw.lstrip().startswith("-"):          # banner / blank lines
                continue
            setattr(opt, m.group(1), _literal(m.group(2)))
    for need in ("checkpoints_dir", "dataset_name", "name"):
        if not hasattr(opt, need):
            raise KeyError(f"{opt_path}: no '{need}' line")
    if opt.dataset_name not in _DATASETS:
        raise KeyError(f"Dataset not recognized: {opt.dataset_name!r} (this package samples {sorted(_DATASETS)})")
    opt.joints_num, opt.max_motion_length = _DATASETS[opt.dataset_name]
    opt.dim_pose = 2 * opt.joints_num
    opt.save_root = pjoin(opt.checkpoints_dir, opt.dataset_name, opt.name)
    opt.model_dir = pjoin(opt.save_root, "model")
    opt.which_epoch = "latest"
    opt.is_train = False
    opt.is_continue = False
    opt.device = device
    return opt


def build_models(opt, precision="fp16"):
    """tools/visualization.py:169-178."""
    from . import MotionTransformer
    return MotionTransformer(input_feats=opt.dim_pose, num_frames=opt.max_motion_length, num_layers=opt.num_layers,
                             latent_dim=opt.latent_dim, device=opt.device, no_clip=opt.no_clip, no_eff=opt.no_eff,
                             music_model_path=None, precision=precision)


def load_mels(music_path):
    """`--music_path`: one mel .npy [Tm,128], or a directory whose *.npy (sorted) form one batch [B,Tm,128]."""
    if os.path.isdir(music_path):
        files = sorted(f for f in os.listdir(music_path) if f.endswith('.npy'))
        if not files:
            raise FileNotFoundError(f"no .npy mel spectrograms under {music_path}")
        mels = [np.load(pjoin(music_path, f)) for f in files]
        if len({m.shape for m in mels}) != 1:
            raise ValueError(f"mel spectrograms under {music_path} differ in shape: {sorted({m.shape for m in mels})}")
        return np.stack(mels).astype(np.float32), files
    if not music_path.endswith('.npy'):
        raise ValueError("--music_path takes the mel spectrogram (.npy [5400,128]) or a directory of them: audio decoding "
                         "(librosa) is outside this package - run the reference's extract_mel_feature "
                         "(tools/visualization.py:152-167) and np.save its result")
    mel = np.load(music_path)
    if mel.ndim != 2 or mel.shape[1] != 128:
        raise ValueError(f"{music_path}: expected a [Tm,128] mel spectrogram, got {mel.shape}")
    return mel.astype(np.float32), [os.path.basename(music_path)]


def make_parser():
    parser = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    parser.add_argument('--opt_path', type=str, required=True, help='Opt path')
    parser.add_argument('--music_path', type=str, required=True, help='mel spectrogram .npy [5400,128] (or a directory of them)')
    parser.add_argument('--motion_length', type=int, default=60, help='kept for compatibility (reference asserts <= 196)')
    parser.add_argument('--result_path', type=str, default="test_sample.gif", help='ignored: rendering is not part of this package')
    parser.add_argument('--npy_path', type=str, default="", help='Path to save the keypoints sequence')
    parser.add_argument('--gpu_id', type=int, default=0, help="which gpu to use")
    parser.add_argument('--model', type=str, default=None, help="checkpoint; default <model_dir>/latest.tar as in the reference")
    parser.add_argument('--smooth', action='store_true', help="Savitzky-Golay smoothing (kernel 19, order 5) as vis_motion applies it")
    parser.add_argument('--seed', type=int, default=None, help="seed of x_T (reproducible sampling)")
    parser.add_argument('--precision', type=str, default="fp16", choices=["fp16", "mixed", "bf16x3", "bf16", "auto"])
    return parser


def main(argv=None):
    import torch
    from . import DDPMTrainer
    args = make_parser().parse_args(argv)
    if args.gpu_id == -1:
        raise SystemExit("this package has no CPU path: --gpu_id must name an MI355X")
    device = torch.device('cuda:%d' % args.gpu_id)
    opt = get_opt(args.opt_path, device)
    assert args.motion_length <= 196
    torch.cuda.set_device(device)
    encoder = build_models(opt, precision=args.precision).to(device)
    trainer = DDPMTrainer(opt, encoder)
    trainer.load(args.model if args.model else pjoin(opt.model_dir, 'latest.tar'))
    trainer.eval_mode()
    trainer.to(opt.device)
    with torch.no_grad():
        mel, names = load_mels(args.music_path)
        # [B, T, 26] on the device; --smooth: smooth_motion(kernel=19) (:126) happens in the loop's final write
        pred_motions = trainer.generate_music_motion(mel, opt.dim_pose, seed=args.seed, smooth=19 if args.smooth else None)
        B, T = pred_motions.shape[0], pred_motions.shape[1]
        motion = pred_motions.view(B, T, 13, 2).cpu().numpy()
    out = motion[0] if mel.ndim == 2 else motion
    print(" #%d frames x %d clip(s): %s" % (T, B, ", ".join(names)))
    if args.npy_path:
        np.save(args.npy_path, out)
        print("saved", args.npy_path, out.shape)
    return out


if __name__ == '__main__':
    main()
