"""CPU restatement of the image half of the reference's input pipeline.  TEST INFRASTRUCTURE ONLY (see oracle/ref_cpu.py).

Downstream/CV/data_utils/dataset.py:77-81,101-112: ``Image.fromarray(rec.get_image()).convert('RGB')`` ->
``transforms.Resize((R, R))`` -> ``ToTensor`` -> ``Normalize(0.5, 0.5)``.  torchvision is absent from the image; its
``Resize`` on a PIL image is ``img.resize((R, R), Image.BILINEAR)``, i.e. Pillow's two-pass fixed-point resampler
(third party: Pillow ``src/libImaging/Resample.c``: ``precompute_coeffs``, ``normalize_coeffs_8bpc``,
``ImagingResampleHorizontal_8bpc`` / ``Vertical_8bpc``), restated here in numpy and PINNED against the installed Pillow
itself by tests/test_image_io.py (bit-exact on random sizes, up- and down-scaling).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """-> (bounds [out, 2] (xmin, count), kk [out, ksize] int32 fixed-point weights) of the bilinear (triangle) filter."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(xmax, np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0
        for x in range(xmax):
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        bounds[xx] = (xmin, xmax)
        for x in range(xmax):
            kk[xx, x] = int(0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] >= 0 else int(-0.5 + w[x] * (1 << PRECISION_BITS))
    return bounds, kk


def _pass(img, out_size, axis):
    bounds, kk = coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.zeros((out_size,) + src.shape[1:], np.int64)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bilinear_u8(img, R):
    """uint8 [H, W, C] -> uint8 [R, R, C]: horizontal pass, rounded to 8 bits, then vertical pass (Pillow's order)."""
    if img.shape[1] != R:
        img = _pass(img, R, 1)
    if img.shape[0] != R:
        img = _pass(img, R, 0)
    return img


def transform(img_u8, R):
    """The reference's full transform: float32 [3, R, R] = Normalize(ToTensor(Resize(img)))."""
    x = resize_bilinear_u8(img_u8, R).astype(np.float32) / np.float32(255.0)
    return ((x - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)
