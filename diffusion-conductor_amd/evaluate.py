"""Batched evaluation driver and pose post-processing.

Replaces the serial B=1 loops of the reference's evaluation scripts
(Diffusion_Stage/tools/eval_new.py:104-134, eval_old_metrics.py:175-191, eval_new_metrics.py:129-156):
walk a dataset directory of ``<clip id>/mel.npy`` (``[5400,128]``) + ``<clip id>/motion.npy`` (``[1800,13,2]``;
on-disk format of the reference's README.md:53-84), sample every clip and report the per-clip MSE the way
eval_new.py does (``np.mean((pred - gt) ** 2)``, summed in directory order, then divided by the clip count).

Here the clips go through the sampler ``batch_size`` at a time: mel files are read by a background thread
into pinned host buffers while the GPU samples the previous batch, the batch is sharded over the ranks of an
initialised ``torch.distributed`` group by ``DDPMTrainer.generate_music_motion``, and the poses come back
with one copy.  ``smooth_motion`` is tools/visualization.py:20-26 (Savitzky-Golay, kernel 19, order 5 at
:126) on the GPU.
"""
from __future__ import annotations

import os
import threading
import time
from os.path import join as pjoin

import numpy as np
import torch


def mse_loss(gt_motion, pred_motion):
    """eval_new.py:37-44."""
    return np.mean((pred_motion - gt_motion) ** 2)


def list_clips(root):
    """Clip ids (sub-directories holding mel.npy and motion.npy), sorted for a reproducible order
    (the reference iterates os.listdir order, eval_new.py:106-113)."""
    return sorted(d for d in os.listdir(root)
                  if os.path.isfile(pjoin(root, d, "mel.npy")) and os.path.isfile(pjoin(root, d, "motion.npy")))


def smooth_motion(kp_pred, kernel=11, order=5):
    """tools/visualization.py:20-26 for a pose tensor on the GPU: [T,13,2], [B,T,13,2] or [B,T,26]."""
    from .native import savgol_filter
    x = kp_pred if torch.is_tensor(kp_pred) else torch.as_tensor(np.asarray(kp_pred))
    single = x.dim() == 3 and x.shape[-1] == 2            # [T, J, 2]
    y = x.unsqueeze(0) if single else x
    if not y.is_cuda:
        raise RuntimeError("smooth_motion runs on the MI355X only (no CPU path); pass a device tensor")
    out = savgol_filter(y.float(), kernel, order).view(y.shape)
    return out[0] if single else out


def clip_noise(seed, index, T, dim_pose):
    """x_T of clip `index` (position in sorted order): its own generator, independent of batching."""
    g = torch.Generator().manual_seed((int(seed) * 1000003 + int(index)) & 0x7fffffffffff)
    return torch.randn(T, dim_pose, generator=g)


class _Prefetcher:
    """Reads the next batch's .npy files into a pinned buffer on a background thread."""

    def __init__(self, root, ids, batch_size, mel_shape):
        self.root, self.ids, self.bs = root, ids, batch_size
        pin = torch.cuda.is_available()
        self.buf = [torch.empty((batch_size,) + mel_shape, dtype=torch.float32, pin_memory=pin) for _ in range(2)]
        self.result = None
        self.error = None
        self.thread = None

    def _load(self, k, slot):
        try:
            self._load_batch(k, slot)
        except BaseException as e:      # handed to the consumer: a dead loader thread must not leave the previous batch behind
            self.error = e

    def _load_batch(self, k, slot):
        ids = self.ids[k * self.bs:(k + 1) * self.bs]
        mel = self.buf[slot][:len(ids)]
        gts = []
        for i, cid in enumerate(ids):
            m = np.load(pjoin(self.root, cid, "mel.npy"))
            if m.shape != tuple(mel.shape[1:]):
                raise ValueError(f"{cid}/mel.npy has shape {m.shape}, expected {tuple(mel.shape[1:])}")
            mel[i].copy_(torch.from_numpy(np.ascontiguousarray(m, np.float32)))
            gts.append(np.load(pjoin(self.root, cid, "motion.npy")))
        self.result = (ids, mel, gts)

    def start(self, k):
        self.result = None
        self.error = None
        self.thread = threading.Thread(target=self._load, args=(k, k & 1), daemon=True)
        self.thread.start()

    def take(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.result


def evaluate_dataset(trainer, root, dim_pose=26, batch_size=32, limit=None, seed=0, smooth=False, verbose=True):
    """Samples every clip under `root` and returns {"per_clip": {id: mse}, "total_loss", "final_mse", "clips",
    "seconds", "frames_per_s"}.  Clip i (in sorted order) starts from noise seeded with (seed, i), so the result
    does not depend on batch_size or on the number of ranks."""
    ids = list_clips(root)
    if limit is not None:
        ids = ids[:int(limit)]
    if not ids:
        raise FileNotFoundError(f"no <id>/mel.npy + <id>/motion.npy pairs under {root}")
    mel_shape = tuple(np.load(pjoin(root, ids[0], "mel.npy"), mmap_mode="r").shape)
    T = (mel_shape[0] - 1) // 3 + 1
    nb = (len(ids) + batch_size - 1) // batch_size
    pf = _Prefetcher(root, ids, batch_size, mel_shape)
    pf.start(0)
    per_clip, total_loss = {}, 0.0
    out_h = None
    t0 = time.perf_counter()
    for k in range(nb):
        bid, mel, gts = pf.take()
        if k + 1 < nb:
            pf.start(k + 1)
        noise = torch.stack([clip_noise(seed, k * batch_size + i, T, dim_pose) for i in range(len(bid))])
        # [B, T, dim_pose] on the device; smoothing (tools/visualization.py:126) happens in the sampling loop's final write
        pred = trainer.generate_music_motion(mel, dim_pose, noise=noise, smooth=19 if smooth else None)
        if out_h is None or out_h.shape[0] < pred.shape[0] or out_h.shape[1:] != pred.shape[1:]:
            out_h = torch.empty(tuple(pred.shape), dtype=pred.dtype, pin_memory=pred.is_cuda)     # pageable D2H costs ~3x the copy
        out_h[:pred.shape[0]].copy_(pred, non_blocking=True)
        if pred.is_cuda:
            torch.cuda.current_stream().synchronize()
        pred = out_h[:pred.shape[0]].numpy()
        for i, cid in enumerate(bid):
            pm = pred[i].reshape([pred[i].shape[0], dim_pose // 2, 2])          # eval_new.py:124-125
            cur = mse_loss(gts[i], pm)
            per_clip[cid] = float(cur)
            total_loss += cur
            if verbose:
                print("cur_loss: ", cur)
                print("total_loss: ", total_loss)
    dt = time.perf_counter() - t0
    final_mse = total_loss / len(ids)
    if verbose:
        print("final total loss: ", total_loss)
        print("final_mse: ", final_mse)
    return {"per_clip": per_clip, "total_loss": float(total_loss), "final_mse": float(final_mse), "clips": len(ids),
            "seconds": dt, "frames_per_s": len(ids) * T / dt}


def main(argv=None):
    """python -m diffusion_conductor_amd.evaluate --data_root <dir> [--model latest.tar] [--batch_size 32]"""
    import argparse
    import types
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--data_root", required=True, help="directory of <id>/mel.npy + <id>/motion.npy (the reference's split dirs)")
    ap.add_argument("--model", default=None, help="checkpoint (.tar with an 'encoder' state_dict, ddpm_trainer.py:303-319); "
                                                  "omitted = seeded synthetic weights")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--diffusion_steps", type=int, default=1000, help="reference default train_options.py:10")
    ap.add_argument("--gpu_id", type=int, default=0)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--smooth", action="store_true", help="Savitzky-Golay smoothing as tools/visualization.py:126")
    ap.add_argument("--no_eff", action="store_true")
    args = ap.parse_args(argv)
    from . import DDPMTrainer, MotionTransformer
    dev = torch.device("cuda", args.gpu_id)
    torch.cuda.set_device(dev)
    enc = MotionTransformer(input_feats=26, num_frames=1800, num_layers=8, latent_dim=128, no_clip=True, no_eff=args.no_eff,
                            device=dev, music_model_path=None)
    opt = types.SimpleNamespace(device=dev, diffusion_steps=args.diffusion_steps, is_train=False)
    tr = DDPMTrainer(opt, enc)
    if args.model:
        tr.load(args.model)
    else:
        from .synthetic import synthetic_state_dict
        enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_state_dict().items()}, strict=True)
    tr.eval_mode()
    r = evaluate_dataset(tr, args.data_root, 26, args.batch_size, args.limit, args.seed, args.smooth)
    print(f"{r['clips']} clips in {r['seconds']:.2f} s = {r['frames_per_s']:.0f} frames/s")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
