"""Drop-in for the sampling half of the reference's ``tools/visualization.py`` command line.

    python -m diffusion_conductor_amd.visualize --opt_path <.../opt.txt> --music_path <mel.npy | dir of *.npy> \
        --npy_path out.npy --gpu_id 0 [--smooth] [--model latest.tar] [--seed 0]

Same flags and call sequence as Diffusion_Stage/tools/visualization.py:180-223: ``get_opt`` (utils/get_opt.py:29-105:
the ``opt.txt`` round trip with its type sniffing and forced fields) -> ``build_models`` (:169-178) -> ``DDPMTrainer`` ->
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

DIM_POS_OHOT = 15      # len(POS_enumerator), utils/word_vectorizer.py:5-21 (a field of opt the sampling path never reads)


def is_float(numStr):
    """utils/get_opt.py:8-18: digits '.' digits, optional sign."""
    numStr = str(numStr).strip().lstrip('-').lstrip('+')
    return re.match(r'^[-+]?[0-9]+\.[0-9]+$', numStr) is not None


def is_number(numStr):
    """utils/get_opt.py:21-26."""
    return str(numStr).strip().lstrip('-').lstrip('+').isdigit()


def get_opt(opt_path, device):
    """utils/get_opt.py:29-105: reads the ``key: value`` lines train.py wrote (options/base_options.py:79-89), sniffs
    bool / float / int / str exactly as the reference does, then forces the inference fields."""
    opt = Namespace()
    opt_dict = vars(opt)
    skip = ('-------------- End ----------------', '------------ Options -------------', '\n')
    print('Reading', opt_path)
    with open(opt_path) as f:
        for line in f:
            if line.strip() not in skip:
                key, value = line.strip().split(': ')
                if value in ('True', 'False'):
                    opt_dict[key] = value == 'True'
                elif is_float(value):
                    opt_dict[key] = float(value)
                elif is_number(value):
                    opt_dict[key] = int(value)
                else:
                    opt_dict[key] = str(value)
    opt_dict['which_epoch'] = 'latest'
    opt_dict.setdefault('num_layers', 8)
    opt_dict.setdefault('latent_dim', 512)
    opt_dict.setdefault('diffusion_steps', 1000)
    opt_dict.setdefault('no_clip', False)
    opt_dict.setdefault('no_eff', False)
    opt.save_root = pjoin(opt.checkpoints_dir, opt.dataset_name, opt.name)
    opt.model_dir = pjoin(opt.save_root, 'model')
    opt.meta_dir = pjoin(opt.save_root, 'meta')
    if opt.dataset_name == 'ConductorMotion100':
        opt.data_root = '/mnt/data/zhuoran/'
        opt.joints_num = 13
        opt.max_motion_length = 1800
    elif opt.dataset_name == 't2m':
        opt.data_root = './data/HumanML3D'
        opt.motion_dir = pjoin(opt.data_root, 'new_joint_vecs')
        opt.text_dir = pjoin(opt.data_root, 'texts')
        opt.joints_num = 22
        opt.dim_pose = 263
        opt.max_motion_length = 196
    elif opt.dataset_name == 'kit':
        opt.data_root = './data/KIT-ML'
        opt.motion_dir = pjoin(opt.data_root, 'new_joint_vecs')
        opt.text_dir = pjoin(opt.data_root, 'texts')
        opt.joints_num = 21
        opt.dim_pose = 251
        opt.max_motion_length = 196
    else:
        raise KeyError('Dataset not recognized')
    opt.dim_word = 300
    opt.num_classes = 200 // opt.unit_length
    opt.dim_pos_ohot = DIM_POS_OHOT
    opt.is_train = False
    opt.is_continue = False
    opt.device = device
    return opt


def build_models(opt):
    """tools/visualization.py:169-178."""
    from . import MotionTransformer
    return MotionTransformer(input_feats=opt.dim_pose, num_frames=opt.max_motion_length, num_layers=opt.num_layers,
                             latent_dim=opt.latent_dim, device=opt.device, no_clip=opt.no_clip, no_eff=opt.no_eff,
                             music_model_path=None)


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
    parser.add_argument('--precision', type=str, default="fp16", choices=["fp16", "mixed", "bf16x3", "bf16"])
    return parser


def main(argv=None):
    import torch
    from . import DDPMTrainer
    from .evaluate import smooth_motion
    args = make_parser().parse_args(argv)
    if args.gpu_id == -1:
        raise SystemExit("this package has no CPU path: --gpu_id must name an MI355X")
    device = torch.device('cuda:%d' % args.gpu_id)
    opt = get_opt(args.opt_path, device)
    opt.do_denoise = True
    assert args.motion_length <= 196
    opt.joints_num = 13
    opt.dim_pose = 26
    torch.cuda.set_device(device)
    encoder = build_models(opt)
    encoder.precision = args.precision
    encoder = encoder.to(device)
    trainer = DDPMTrainer(opt, encoder)
    trainer.load(args.model if args.model else pjoin(opt.model_dir, 'latest.tar'))
    trainer.eval_mode()
    trainer.to(opt.device)
    with torch.no_grad():
        mel, names = load_mels(args.music_path)
        pred_motions = trainer.generate_music_motion(mel, opt.dim_pose, seed=args.seed)      # [B, T, 26] on the device
        B, T = pred_motions.shape[0], pred_motions.shape[1]
        motion = pred_motions.view(B, T, 13, 2)
        if args.smooth:
            motion = smooth_motion(motion, kernel=19)
        motion = motion.cpu().numpy()
    out = motion[0] if mel.ndim == 2 else motion
    print(" #%d frames x %d clip(s): %s" % (T, B, ", ".join(names)))
    if args.npy_path:
        np.save(args.npy_path, out)
        print("saved", args.npy_path, out.shape)
    return out


if __name__ == '__main__':
    main()
