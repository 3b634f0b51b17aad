"""Batched evaluation driver and pose post-processing.

Replaces the serial B=1 loops of the reference's evaluation scripts
(Diffusion_Stage/tools/eval_new.py:104-134, eval_old_metrics.py:175-191, eval_new_metrics.py:129-156):
walk a dataset directory of ``<clip id>/mel.npy`` (``[5400,128]``) + ``<clip id>/motion.npy`` (``[1800,13,2]``;
on-disk format of the reference's README.md:53-84), sample every clip and report the per-clip MSE the way
eval_new.py does (``np.mean((pred - gt) ** 2)``, summed in directory order, then divided by the clip count).

Here the clips go through the sampler ``batch_size`` at a time: mel files and noise are prepared by background
threads in pinned host buffers while the GPU samples the previous batch, the batch is sharded over the ranks of an
initialised ``torch.distributed`` group by ``DDPMTrainer.generate_music_motion``, the poses come back with one
copy behind an event and are scored on a third thread - the GPU's queue never runs dry (tools/time_evaluate.py).  ``smooth_motion`` is tools/visualization.py:20-26 (Savitzky-Golay, kernel 19, order 5 at
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


def clip_noise(seed, index, T, dim_pose, out=None):
    """x_T of clip `index` (position in sorted order): its own generator, independent of batching.  `out`: a [T, dim_pose] fp32 CPU
    tensor to draw into (a row of the pinned batch buffer: a `copy_` between two CPU tensors goes through torch's thread pool and
    costs milliseconds per clip - it was what bound the evaluation driver's loader)."""
    g = torch.Generator().manual_seed((int(seed) * 1000003 + int(index)) & 0x7fffffffffff)
    return torch.randn(T, dim_pose, generator=g) if out is None else torch.randn(T, dim_pose, generator=g, out=out)


def _read_npy_into(path, dst):
    """One clip's mel.npy into a row of the pinned batch buffer with a single copy (np.load would allocate an array first and the
    batch of 32 x 2.8 MB would be copied twice per 40 ms of GPU work)."""
    m = np.load(path, mmap_mode="r")
    if m.shape != tuple(dst.shape):
        raise ValueError(f"{os.path.basename(os.path.dirname(path))}/mel.npy has shape {m.shape}, expected {tuple(dst.shape)}")
    np.copyto(dst.numpy(), m, casting="same_kind")


class _Prefetcher:
    """Prepares the next batch on background threads while the GPU samples the current one: the .npy files into a pinned buffer
    (a small pool reads the clips of a batch side by side), the ground-truth motions, and - `noise=(seed, T, dim_pose)` - every
    clip's x_T into a second pinned buffer.  Two slots; `start(k, after=event)` waits for `event` (the GPU is done with the slot's
    previous batch) before it overwrites the slot."""

    def __init__(self, root, ids, batch_size, mel_shape, noise=None, workers=None):
        self.root, self.ids, self.bs = root, ids, batch_size
        pin = torch.cuda.is_available()
        self.buf = [torch.empty((batch_size,) + tuple(mel_shape), dtype=torch.float32, pin_memory=pin) for _ in range(2)]
        self.noise_spec = noise
        self.nbuf = [torch.empty((batch_size, noise[1], noise[2]), dtype=torch.float32, pin_memory=pin) for _ in range(2)] if noise else None
        # (four threads: with 16 the loaders fight the enqueueing thread for the interpreter lock - 288 clips took 1.35 s instead of 0.40,
        # profiles/r05_time_evaluate.txt)
        self.workers = max(1, int(workers)) if workers else 4
        self.result = None
        self.noise = None
        self.error = None
        self.thread = None

    def _load(self, k, slot, after):
        try:
            if after is not None:
                after.synchronize()
            self._load_batch(k, slot)
        except BaseException as e:      # handed to the consumer: a dead loader thread must not leave the previous batch behind
            self.error = e

    def _load_batch(self, k, slot):
        ids = self.ids[k * self.bs:(k + 1) * self.bs]
        mel = self.buf[slot][:len(ids)]
        gts = [None] * len(ids)
        nz = self.nbuf[slot][:len(ids)] if self.nbuf else None

        def one(i):
            _read_npy_into(pjoin(self.root, ids[i], "mel.npy"), mel[i])
            gts[i] = np.load(pjoin(self.root, ids[i], "motion.npy"))
            if nz is not None:
                seed, T, dim_pose = self.noise_spec
                clip_noise(seed, k * self.bs + i, T, dim_pose, out=nz[i])

        if self.workers > 1 and len(ids) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(self.workers) as ex:
                list(ex.map(one, range(len(ids))))      # (re-raises a worker's exception)
        else:
            for i in range(len(ids)):
                one(i)
        self.noise = nz
        self.result = (ids, mel, gts)

    def start(self, k, after=None):
        self.result = None
        self.noise = None
        self.error = None
        self.thread = threading.Thread(target=self._load, args=(k, k & 1, after), daemon=True)
        self.thread.start()

    def take(self):
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.result


class _Scorer:
    """Per-clip MSE of a batch (eval_new.py:124-131) on a background thread, behind the event that says the batch's poses have
    landed in the pinned buffer - the main thread is enqueueing the next batch meanwhile."""

    def __init__(self, dim_pose):
        import queue
        self.dim_pose = dim_pose
        self.q = queue.Queue()
        self.done = {}                  # batch -> [(clip id, mse)] | exception
        self.cv = threading.Condition()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            k, bid, gts, pred_h, event, _keep = job
            try:
                if event is not None:
                    event.synchronize()
                pred = pred_h.numpy()
                if not np.isfinite(pred).all():
                    raise FloatingPointError(f"non-finite poses in batch {k}")
                res = []
                for i, cid in enumerate(bid):
                    pm = pred[i].reshape([pred[i].shape[0], self.dim_pose // 2, 2])          # eval_new.py:124-125
                    res.append((cid, mse_loss(gts[i], pm)))
            except BaseException as e:
                res = e
            with self.cv:
                self.done[k] = res
                self.cv.notify_all()

    def submit(self, k, bid, gts, pred_h, event, keep):
        self.q.put((k, bid, gts, pred_h, event, keep))

    def wait_for(self, k, reraise=True):
        """Blocks until batch k has been scored (k < 0: nothing to wait for); returns its [(clip id, mse)] or raises its error.
        reraise=False only waits (the sampling loop frees a pinned slot with it: a failed batch is dealt with when the results are
        collected, where a non-finite one is sampled again the checked way)."""
        if k < 0:
            return None
        with self.cv:
            self.cv.wait_for(lambda: k in self.done)
            res = self.done[k]
        if isinstance(res, BaseException):
            if reraise:
                raise res
            return None
        return res

    def close(self):
        self.q.put(None)


def evaluate_dataset(trainer, root, dim_pose=26, batch_size=32, limit=None, seed=0, smooth=False, verbose=True):
    """Samples every clip under `root` and returns {"per_clip": {id: mse}, "total_loss", "final_mse", "clips",
    "seconds", "frames_per_s"}.  Clip i (in sorted order) starts from noise seeded with (seed, i), so the result
    does not depend on batch_size or on the number of ranks.

    The host stays out of the GPU's way: batch k + 1's files and noise are prepared, and batch k - 1's MSEs computed, on
    background threads while batch k is sampled; the poses come back through a pinned double buffer behind an event, not a stream
    synchronisation; the sampler's per-loop numeric check (a status read that waits for the GPU) is replaced by a finiteness check
    of the poses on the scoring thread, and a batch that fails it is sampled again the checked way (which is where
    precision="auto" falls back to the bf16-range mode).  At most two batches are in flight."""
    ids = list_clips(root)
    if limit is not None:
        ids = ids[:int(limit)]
    if not ids:
        raise FileNotFoundError(f"no <id>/mel.npy + <id>/motion.npy pairs under {root}")
    mel_shape = tuple(np.load(pjoin(root, ids[0], "mel.npy"), mmap_mode="r").shape)
    T = (mel_shape[0] - 1) // 3 + 1
    nb = (len(ids) + batch_size - 1) // batch_size
    pf = _Prefetcher(root, ids, batch_size, mel_shape, noise=(seed, T, dim_pose))
    pf.start(0)
    scorer = _Scorer(dim_pose)
    enc = getattr(trainer, "encoder", None)
    serial = bool(os.environ.get("DC_EVAL_SERIAL"))      # A/B switch (tools/time_evaluate.py): round 4's behaviour - the status read's stream
    checked = None if serial else getattr(enc, "check_numerics", None)      # synchronisation and the MSEs between two batches
    dev = getattr(trainer, "device", None)
    on_gpu = dev is not None and torch.device(dev).type == "cuda"
    out_h = [None, None]
    slot_free = [None, None]             # event: the GPU has consumed the slot's pinned mel / noise buffers
    results = {}
    exchange_before = getattr(enc, "combine_exchange", None)
    ev_first = ev_last = None            # completion events of the first and the last batch: the GPU's own steady-state period
    n_after_first = 0
    waits = {"loader": 0.0, "enqueue": 0.0, "scorer": 0.0}      # where the main thread spent its time (seconds): waiting for the next
    t0 = time.perf_counter()                                    # batch's files, inside generate_music_motion, waiting for batch k - 2's scores
    try:
        if checked:
            enc.check_numerics = False   # (see above: checked on the scoring thread instead)
            # ... which cannot see the one failure that leaves finite poses: a timed-out combine exchange of the small-batch layer kernel
            # (dc_ddim.h, DC_STATUS_TIMEOUT; possible only on a GPU shared with other work).  Unchecked loops run the form without it.
            enc.combine_exchange = False
        for k in range(nb):
            tw = time.perf_counter()
            bid, mel, gts = pf.take()
            waits["loader"] += time.perf_counter() - tw
            noise = pf.noise
            if k + 1 < nb:
                pf.start(k + 1, after=slot_free[(k + 1) & 1])
            if on_gpu:
                noise = noise.to(dev, non_blocking=True)      # (a blocking copy on the default stream would wait for the previous batch)
            # [B, T, dim_pose] on the device; smoothing (tools/visualization.py:126) happens in the sampling loop's final write
            tw = time.perf_counter()
            pred = trainer.generate_music_motion(mel, dim_pose, noise=noise, smooth=19 if smooth else None)
            waits["enqueue"] += time.perf_counter() - tw
            tw = time.perf_counter()
            scorer.wait_for(k - 2, reraise=False)             # the pinned pose buffer of this slot has been scored (errors: collected below)
            waits["scorer"] += time.perf_counter() - tw
            s = k & 1
            if out_h[s] is None or out_h[s].shape[0] < pred.shape[0] or out_h[s].shape[1:] != pred.shape[1:]:
                out_h[s] = torch.empty((batch_size,) + tuple(pred.shape[1:]), dtype=pred.dtype, pin_memory=pred.is_cuda)     # pageable D2H costs ~3x the copy
            ph = out_h[s][:pred.shape[0]]
            ph.copy_(pred, non_blocking=True)
            ev = None
            if pred.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                slot_free[s] = ev
                ev_first = ev_first or ev
                ev_last, n_after_first = ev, (n_after_first + len(bid) if ev_first is not ev else 0)
            scorer.submit(k, bid, gts, ph, ev, pred)
            if serial:
                scorer.wait_for(k, reraise=False)
        for k in range(nb):
            try:
                results[k] = scorer.wait_for(k)
            except FloatingPointError:
                if not checked:
                    raise
                # the checked path (status read, precision="auto" fallback) for this batch alone
                enc.check_numerics = True
                pf2 = _Prefetcher(root, ids, batch_size, mel_shape, noise=(seed, T, dim_pose))
                pf2._load_batch(k, 0)
                bid, mel, gts = pf2.result
                pred = trainer.generate_music_motion(mel, dim_pose, noise=pf2.noise, smooth=19 if smooth else None).cpu().numpy()
                results[k] = [(cid, mse_loss(gts[i], pred[i].reshape([pred[i].shape[0], dim_pose // 2, 2]))) for i, cid in enumerate(bid)]
                enc.check_numerics = False
    finally:
        scorer.close()
        if checked:
            enc.check_numerics = checked
            enc.combine_exchange = exchange_before
    dt = time.perf_counter() - t0
    per_clip, total_loss = {}, 0.0
    for k in range(nb):
        for cid, cur in results[k]:
            per_clip[cid] = float(cur)
            total_loss += cur
            if verbose:
                print("cur_loss: ", cur)
                print("total_loss: ", total_loss)
    final_mse = total_loss / len(ids)
    if verbose:
        print("final total loss: ", total_loss)
        print("final_mse: ", final_mse)
    out = {"per_clip": per_clip, "total_loss": float(total_loss), "final_mse": float(final_mse), "clips": len(ids),
           "seconds": dt, "frames_per_s": len(ids) * T / dt, "main_thread_s": {k: round(v, 4) for k, v in waits.items()}}
    if ev_first is not None and ev_last is not ev_first:
        # clips of batches 2 .. n over the time between the completion of batch 1 and of batch n ON THE GPU: the pipeline's rate
        # without the two things nothing can hide (the first batch's load before any GPU work, the last batch's scoring after it)
        out["steady_frames_per_s"] = n_after_first * T / (ev_first.elapsed_time(ev_last) * 1e-3)
    return out


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
