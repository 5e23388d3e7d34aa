"""Snippet loader: the mirror of the reference's ``data_loader.DataLoader``.

On-disk format (data/prepare_train_data.py:33-51, 99-113): ``<root>/<folder>/<frame>.jpg`` holds the seq = N+1
frames of a snippet side by side, ``<frame>_cam.txt`` one line ``fx,0.,cx,0.,fy,cy,0.,0.,1.``, and ``train.txt`` /
``val.txt`` one ``folder frame`` pair per line.

Split of the work (MI355X first, not a translation of the TF queue graph):
  * host: file lists, shuffling, JPEG decode (PIL) in a thread pool straight into PINNED uint8 batch buffers, a
    prefetch thread that keeps ``prefetch`` decoded batches ahead, an asynchronous copy to the GPU on a copy stream;
  * device (``csrc/snippet.hip``, ``sfm_snippet_unpack``): slicing into target / source stack, the augmentation,
    both uint8 quantisations, the [-1, 1] mapping of ``preprocess_image`` and the per-scale intrinsics - two launches,
    the fp32 tensors are written once, in the layout ``sfm_amd.multi_scale_loss`` reads.
The random draws are made on the host (``random.Random(seed)``) and passed by value; TF's generator is not
reproducible outside TF, so "same draws" is the contract, not "same stream".

Method names and meaning follow data_loader.py; where the reference returns graph tensors fed by queue runners,
``load_train_batch`` / ``load_val_batch`` return an iterator of ready device batches.  No CPU path.
"""
from __future__ import annotations

import os
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_of

Tensor = torch.Tensor


class AugmentDraws(tuple):
    """(brightness, (col_r, col_g, col_b), swap, contrast): one set of draws per batch (data_loader.py:137, 227-242)."""

    def __new__(cls, brightness: float, colors: Sequence[float], swap: bool, contrast: float):
        return super().__new__(cls, (float(brightness), tuple(float(c) for c in colors), bool(swap), float(contrast)))

    @classmethod
    def draw(cls, rng: random.Random) -> "AugmentDraws":
        # the order of the reference's tf.random_uniform calls: brightness, colours, swap, contrast
        bright = rng.uniform(1 / 1.2, 1 / 0.8)
        colors = [rng.uniform(0.8, 1.2) for _ in range(3)]
        swap = rng.uniform(0, 1) < 0.5
        return cls(bright, colors, swap, rng.uniform(0.5, 1.5))


def snippet_unpack(image_seq: Tensor, intrinsics: Tensor, img_height: int, img_width: int, num_source: int,
                   num_scales: int, draws: Optional[AugmentDraws] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Decoded uint8 snippets ``[B, H, (N+1) W, 3]`` + ``[B, 3, 3]`` intrinsics on the GPU ->
    ``(tgt_image [B,H,W,3], src_image_stack [B,H,W,3N], intrinsics [B,S,3,3])``, images fp32 in [-1, 1]: what
    ``build_train_graph`` holds after its two ``preprocess_image`` calls (SfMLearner_uni.py:31-32).
    ``draws=None`` is ``fetch_val_batch`` (no augmentation), otherwise ``fetch_train_batch``."""
    if not isinstance(image_seq, torch.Tensor) or not image_seq.is_cuda or image_seq.dtype != torch.uint8:
        raise TypeError("image_seq must be a uint8 tensor on a ROCm device (sfm_amd has no CPU path)")
    if not isinstance(intrinsics, torch.Tensor) or intrinsics.device != image_seq.device or intrinsics.dtype != torch.float32:
        raise TypeError("intrinsics must be an fp32 tensor on the device of image_seq")
    n_f = num_source + 1
    if image_seq.dim() != 4 or tuple(image_seq.shape[1:]) != (img_height, n_f * img_width, 3):
        raise ValueError(f"image_seq must be [B, {img_height}, {n_f * img_width}, 3], got {tuple(image_seq.shape)}")
    bsz = image_seq.shape[0]
    if tuple(intrinsics.shape) != (bsz, 3, 3):
        raise ValueError(f"intrinsics must be [{bsz}, 3, 3], got {tuple(intrinsics.shape)}")
    if num_source not in (2, 4):
        raise ValueError("num_source must be 2 or 4: unpack_image_sequence takes N/2 frames either side of the target")
    image_seq, intrinsics = image_seq.contiguous(), intrinsics.contiguous()
    dev = image_seq.device
    tgt = torch.empty(bsz, img_height, img_width, 3, dtype=torch.float32, device=dev)
    src = torch.empty(bsz, img_height, img_width, 3 * num_source, dtype=torch.float32, device=dev)
    k_ms = torch.empty(bsz, num_scales, 3, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws = None
        if draws is not None:
            n_ws = _lib.load().sfm_snippet_workspace_floats(bsz, img_height, img_width, num_source)
            ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        bright, colors, swap, contrast = draws if draws is not None else (1.0, (1.0, 1.0, 1.0), False, 1.0)
        call("sfm_snippet_unpack", ptr(image_seq), ptr(intrinsics), bsz, img_height, img_width, num_source, num_scales,
             int(draws is not None), bright, colors[0], colors[1], colors[2], int(swap), contrast, ptr(tgt), ptr(src),
             ptr(k_ms), ptr(ws), stream_of(image_seq))
    return tgt, src, k_ms


class DataLoader(object):
    """``DataLoader(dataset_dir, batch_size, img_height, img_width, num_source, num_scales)`` as in data_loader.py:7-20.
    Extra keyword arguments configure what TF's queue runners did implicitly."""

    def __init__(self, dataset_dir=None, batch_size=None, img_height=None, img_width=None, num_source=None,
                 num_scales=None, device="cuda", seed: Optional[int] = None, num_workers: int = 8, prefetch: int = 2):
        self.dataset_dir = dataset_dir
        self.batch_size = batch_size
        self.img_height = img_height
        self.img_width = img_width
        self.num_source = num_source
        self.num_scales = num_scales
        self.device = torch.device(device)
        self.seed = seed
        self.num_workers = max(1, int(num_workers))
        self.prefetch = max(1, int(prefetch))
        self.steps_per_epoch = None

    # ------------------------------------------------------------------ host side
    def format_file_list(self, data_root: str, split: str) -> Dict[str, List[str]]:
        """data_loader.py:282-294."""
        with open(data_root + "/%s.txt" % split, "r") as f:
            frames = [ln for ln in f.readlines() if ln.strip()]
        subfolders = [x.split(" ")[0] for x in frames]
        frame_ids = [x.split(" ")[1].rstrip("\n") for x in frames]
        return {
            "image_file_list": [os.path.join(data_root, s, f + ".jpg") for s, f in zip(subfolders, frame_ids)],
            "cam_file_list": [os.path.join(data_root, s, f + "_cam.txt") for s, f in zip(subfolders, frame_ids)],
        }

    @staticmethod
    def read_cam_file(path: str) -> np.ndarray:
        """The ``tf.decode_csv`` + reshape of data_loader.py:114-123: nine floats -> [3, 3]."""
        with open(path, "r") as f:
            vals = f.readline().strip().split(",")
        if len(vals) != 9:
            raise ValueError(f"{path}: expected 9 comma-separated intrinsics, got {len(vals)}")
        return np.array([np.float32(v) for v in vals], dtype=np.float32).reshape(3, 3)

    def _decode_into(self, path: str, out: np.ndarray) -> None:
        from PIL import Image  # host-side JPEG decode (tf.image.decode_jpeg, data_loader.py:109)
        with Image.open(path) as im:
            arr = np.asarray(im.convert("RGB"))
        if arr.shape != out.shape:
            raise ValueError(f"{path}: decoded {arr.shape}, expected {out.shape} = [H, (num_source+1) W, 3]")
        out[...] = arr

    def _host_batches(self, split: str, shuffle: bool, epochs: Optional[int]):
        """Generator of (pinned uint8 [B,H,(N+1)W,3], pinned fp32 [B,3,3]) decoded on ``num_workers`` threads."""
        files = self.format_file_list(self.dataset_dir, split)
        pairs = list(zip(files["image_file_list"], files["cam_file_list"]))
        n_batches = len(pairs) // self.batch_size
        if n_batches == 0:
            raise ValueError(f"{split}.txt lists {len(pairs)} snippets, fewer than one batch of {self.batch_size}")
        rng = random.Random(self.seed)
        pin = self.device.type == "cuda" and torch.cuda.is_available()
        shape = (self.batch_size, self.img_height, (self.num_source + 1) * self.img_width, 3)
        # slot = [pinned images, pinned intrinsics, event of the slot's last host -> device copy (set by the consumer)].
        # In flight: the queue + the one being filled + the one being copied; a slot comes round again only after the
        # consumer has taken the batch behind it, i.e. after that event was recorded.
        ring = [[torch.empty(shape, dtype=torch.uint8, pin_memory=pin),
                 torch.empty(self.batch_size, 3, 3, dtype=torch.float32, pin_memory=pin), None]
                for _ in range(self.prefetch + 2)]
        slot, epoch = 0, 0
        with ThreadPoolExecutor(self.num_workers) as pool:
            while epochs is None or epoch < epochs:
                order = list(pairs)
                if shuffle:  # one permutation for images and intrinsics (string_input_producer with a shared seed)
                    rng.shuffle(order)
                for bi in range(n_batches):
                    entry = ring[slot]
                    img_buf, cam_buf = entry[0], entry[1]
                    if entry[2] is not None:   # the copy that last read this pinned buffer: waited for HERE, on the producer
                        entry[2].synchronize()  # thread (long finished by the time the ring comes round), never by the consumer
                        entry[2] = None
                    slot = (slot + 1) % len(ring)
                    img_np, cam_np = img_buf.numpy(), cam_buf.numpy()
                    chunk = order[bi * self.batch_size:(bi + 1) * self.batch_size]
                    jobs = [pool.submit(self._decode_into, p, img_np[j]) for j, (p, _) in enumerate(chunk)]
                    for j, (_, c) in enumerate(chunk):
                        cam_np[j] = self.read_cam_file(c)
                    for job in jobs:
                        job.result()
                    yield entry
                epoch += 1

    def _device_batches(self, split: str, augment: bool, shuffle: bool, epochs: Optional[int]) -> Iterator:
        if self.device.type != "cuda":
            raise TypeError("DataLoader produces device batches: device must be a ROCm device (no CPU path)")
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def producer():
            try:
                for item in self._host_batches(split, shuffle, epochs):
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as exc:  # surface decode errors in the consumer
                q.put(exc)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        draw_rng = random.Random(None if self.seed is None else self.seed + 1)
        copy_stream = torch.cuda.Stream(self.device)
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                img_buf, cam_buf = item[0], item[1]
                with torch.cuda.stream(copy_stream):
                    seq_d = img_buf.to(self.device, non_blocking=True)
                    cam_d = cam_buf.to(self.device, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(copy_stream)
                torch.cuda.current_stream(self.device).wait_event(done)
                seq_d.record_stream(torch.cuda.current_stream(self.device))
                cam_d.record_stream(torch.cuda.current_stream(self.device))
                draws = AugmentDraws.draw(draw_rng) if augment else None
                out = snippet_unpack(seq_d, cam_d, self.img_height, self.img_width, self.num_source, self.num_scales,
                                     draws)
                item[2] = done  # the ring's producer waits for it before it refills the pinned buffer (no host sync here)
                yield out
        finally:
            stop.set()

    # ------------------------------------------------------------------ the reference's entry points
    def load_train_batch(self, epochs: Optional[int] = None) -> Iterator[Tuple[Tensor, Tensor, Tensor]]:
        """data_loader.py:75-124 + fetch_train_batch :126-144: shuffled, augmented batches, endlessly (or ``epochs``)."""
        files = self.format_file_list(self.dataset_dir, "train")
        self.steps_per_epoch = int(len(files["image_file_list"]) // self.batch_size)
        return self._device_batches("train", augment=True, shuffle=True, epochs=epochs)

    def load_val_batch(self, epochs: Optional[int] = None) -> Iterator[Tuple[Tensor, Tensor, Tensor]]:
        """data_loader.py:22-57 + fetch_val_batch :60-73: shuffled batches without augmentation."""
        return self._device_batches("val", augment=False, shuffle=True, epochs=epochs)

    def fetch_train_batch(self, image_seq: Tensor, intrinsics: Tensor, draws: Optional[AugmentDraws] = None):
        """The device half of fetch_train_batch (:126-144) on an already decoded uint8 batch."""
        if draws is None:
            draws = AugmentDraws.draw(random.Random(self.seed))
        return snippet_unpack(image_seq, intrinsics, self.img_height, self.img_width, self.num_source, self.num_scales,
                              draws)

    def fetch_val_batch(self, image_seq: Tensor, intrinsics: Tensor):
        """The device half of fetch_val_batch (:60-73)."""
        return snippet_unpack(image_seq, intrinsics, self.img_height, self.img_width, self.num_source, self.num_scales)

    def batch_unpack_image_sequence(self, image_seq: Tensor, img_height: int, img_width: int, num_source: int):
        """data_loader.py:326-346 on uint8 snippets: (tgt, src stack) in [-1, 1] (the callers preprocess first, SfMLearner_uni.py:438-442)."""
        eye = torch.eye(3, dtype=torch.float32, device=image_seq.device).expand(image_seq.shape[0], 3, 3).contiguous()
        tgt, src, _ = snippet_unpack(image_seq, eye, img_height, img_width, num_source, 1)
        return tgt, src

    def unpack_image_sequence(self, image_seq: Tensor, img_height: int, img_width: int, num_source: int):
        """data_loader.py:296-324 on one decoded snippet [H, (N+1) W, 3]."""
        tgt, src = self.batch_unpack_image_sequence(image_seq[None], img_height, img_width, num_source)
        return tgt[0], src[0]

    def get_multi_scale_intrinsics(self, intrinsics: Tensor, num_scales: int) -> Tensor:
        """data_loader.py:348-359: [B,3,3] -> [B,S,3,3] (same kernel, on a 1-pixel dummy snippet batch)."""
        if not intrinsics.is_cuda:
            raise TypeError("intrinsics must be on a ROCm device (sfm_amd has no CPU path)")
        bsz = intrinsics.shape[0]
        dummy = torch.zeros(bsz, 1, 3, 3, dtype=torch.uint8, device=intrinsics.device)
        return snippet_unpack(dummy, intrinsics.float(), 1, 1, 2, num_scales)[2]
