"""CPU restatement (NumPy) of the snippet loader: data_loader.py of the reference, the live lines only.
TEST INFRASTRUCTURE ONLY - see the header of oracle/ref_graph.py for who may import it.

Reference lines followed (paths relative to the reference repo):
  format_file_list data_loader.py:282-294; cam line decode :114-123 (written by data/prepare_train_data.py:49-51);
  unpack_image_sequence :296-324, batch_unpack_image_sequence :326-346; fetch_train_batch :126-144;
  data_augmentation :157-280 (live lines 272, 274-276, 279: brightness, colour, uint8 round trip, width mirror;
  rotate / scaling / cropping / gamma / hue are commented out there); tf.image.random_contrast :137;
  get_multi_scale_intrinsics :348-359 + make_intrinsics_matrix :146-155; preprocess_image SfMLearner_uni.py:449-452.

PARITY UNPINNED by the reference itself (TensorFlow is not installed here, the reference has no fixtures).
TF-1 semantics encoded from its documented behaviour - ASSUMPTIONS:
  * tf.image.convert_image_dtype(uint8 -> float32) = cast * fp32(1/255); (float32 -> uint8, saturate=True) =
    x * 255.5, clamped to [0, 255], cast (truncation);
  * tf.image.adjust_contrast on uint8 = convert to float32, (x - mean_HW) * factor + mean_HW per (image, channel)
    (AdjustContrastv2), convert back with saturate=True.  The fp32 summation order of TF's mean is unknown; here the
    mean is the fp64 sum rounded once to fp32.  A different order could move a mean by an ulp and with it single
    uint8 values by one step;
  * the random draws (tf.random_uniform) are inputs here: TF's generator is not reproducible outside TF;
  * tf.image.decode_jpeg's pixels = libjpeg's default decode (PIL gives the same on libjpeg-turbo); not restated.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import numpy as np

f32 = np.float32


def format_file_list(data_root: str, split: str) -> Dict[str, List[str]]:
    """data_loader.py:282-294: lines 'folder frame' -> <root>/<folder>/<frame>.jpg and ..._cam.txt."""
    with open(data_root + "/%s.txt" % split, "r") as f:
        frames = f.readlines()
    subfolders = [x.split(" ")[0] for x in frames]
    frame_ids = [x.split(" ")[1][:-1] for x in frames]
    return {
        "image_file_list": [os.path.join(data_root, subfolders[i], frame_ids[i] + ".jpg") for i in range(len(frames))],
        "cam_file_list": [os.path.join(data_root, subfolders[i], frame_ids[i] + "_cam.txt") for i in range(len(frames))],
    }


def decode_cam_line(line: str) -> np.ndarray:
    """tf.decode_csv with nine float columns, reshaped [3, 3] (data_loader.py:114-123)."""
    vals = [f32(v) for v in line.strip().split(",")]
    if len(vals) != 9:
        raise ValueError("expected 9 comma-separated intrinsics")
    return np.array(vals, dtype=f32).reshape(3, 3)


def unpack_image_sequence(image_seq: np.ndarray, img_height: int, img_width: int, num_source: int):
    """data_loader.py:296-324 on one decoded jpeg [H, (N+1) W, 3]."""
    half = num_source // 2
    tgt_start = int(img_width * half)
    tgt = image_seq[:, tgt_start:tgt_start + img_width, :]
    src_1 = image_seq[:, 0:int(img_width * half), :]
    src_2 = image_seq[:, int(tgt_start + img_width):int(tgt_start + img_width) + int(img_width * half), :]
    src_seq = np.concatenate([src_1, src_2], axis=1)
    src_stack = np.concatenate([src_seq[:, i * img_width:(i + 1) * img_width, :] for i in range(num_source)], axis=2)
    assert tgt.shape == (img_height, img_width, 3) and src_stack.shape == (img_height, img_width, 3 * num_source)
    return tgt, src_stack


def batch_unpack_image_sequence(image_seq: np.ndarray, img_height: int, img_width: int, num_source: int):
    """data_loader.py:326-346 on a batch [B, H, (N+1) W, 3]."""
    outs = [unpack_image_sequence(image_seq[b], img_height, img_width, num_source) for b in range(image_seq.shape[0])]
    return np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])


def to_float(u8: np.ndarray) -> np.ndarray:
    return u8.astype(f32) * f32(1.0 / 255)


def to_uint8_saturate(x: np.ndarray) -> np.ndarray:
    s = x.astype(f32) * f32(255.5)
    return np.clip(s, f32(0), f32(255)).astype(np.uint8)  # astype truncates toward zero


def data_augmentation(im_u8: np.ndarray, brightness: float, colors: Sequence[float], swap: bool) -> np.ndarray:
    """data_loader.py:272-279 on image_all [B, H, W, 3 (N+1)] uint8 (intrinsics pass through unchanged)."""
    im = to_float(im_u8)
    im = im * f32(brightness)                                                  # random_brightness :239-242
    color = np.array([colors[c % 3] for c in range(im.shape[3])], dtype=f32)   # random_saturation :232-238
    im = im * color
    im = to_uint8_saturate(im)                                                 # :276
    if swap:                                                                   # random_swapping :227-231
        im = im[:, :, ::-1, :]
    return np.ascontiguousarray(im)


def adjust_contrast(im_u8: np.ndarray, factor: float) -> np.ndarray:
    """tf.image.random_contrast's deterministic part (data_loader.py:137) on uint8 [B, H, W, C]."""
    x = to_float(im_u8)
    mean = (x.astype(np.float64).sum(axis=(1, 2), keepdims=True) / (x.shape[1] * x.shape[2])).astype(f32)
    return to_uint8_saturate((x - mean) * f32(factor) + mean)


def get_multi_scale_intrinsics(intrinsics: np.ndarray, num_scales: int) -> np.ndarray:
    """data_loader.py:348-359: [B, 3, 3] -> [B, S, 3, 3] with fx, fy, cx, cy divided by 2**s."""
    k = intrinsics.astype(f32)
    out = np.zeros((k.shape[0], num_scales, 3, 3), dtype=f32)
    for s in range(num_scales):
        d = f32(2 ** s)
        out[:, s, 0, 0] = k[:, 0, 0] / d
        out[:, s, 1, 1] = k[:, 1, 1] / d
        out[:, s, 0, 2] = k[:, 0, 2] / d
        out[:, s, 1, 2] = k[:, 1, 2] / d
        out[:, s, 2, 2] = 1.0
    return out


def preprocess_image(u8: np.ndarray) -> np.ndarray:
    """SfMLearner_uni.py:449-452."""
    return to_float(u8) * f32(2.0) - f32(1.0)


def fetch_batch(image_seq_u8: np.ndarray, intrinsics: np.ndarray, img_height: int, img_width: int, num_source: int,
                num_scales: int, augment: bool, brightness: float = 1.0, colors: Sequence[float] = (1.0, 1.0, 1.0),
                swap: bool = False, contrast: float = 1.0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """decoded batch [B, H, (N+1) W, 3] uint8 -> (tgt [-1,1] fp32, src stack, multi-scale intrinsics): what reaches
    build_train_graph's first two lines.  augment=True: fetch_train_batch (:126-144); False: fetch_val_batch (:60-73)."""
    tgt, src = batch_unpack_image_sequence(image_seq_u8, img_height, img_width, num_source)
    if augment:
        image_all = np.concatenate([tgt, src], axis=3)
        image_all = data_augmentation(image_all, brightness, colors, swap)
        image_all = adjust_contrast(image_all, contrast)
        tgt, src = image_all[..., :3], image_all[..., 3:]
    return preprocess_image(tgt), preprocess_image(src), get_multi_scale_intrinsics(intrinsics, num_scales)
