"""CPU oracle (test infrastructure only) for the decoded-image preparation of the thermal loaders:
`cv2.resize(img, (960, 320), INTER_LINEAR | INTER_NEAREST)` + the column crops `[:, 150:850]` / `[:, 148:852]`
(models/confusion_maximization/thermal_loader.py:621-640, 783-791).

PARITY UNPINNED: the arithmetic lives in OpenCV (`cv2`), which is not installed in the build container, and the reference
holds no fixtures for it.  This file restates OpenCV's published algorithm (modules/imgproc/src/resize.cpp, 4.x):
  * INTER_LINEAR, uint8: fixed point.  Source coordinate fx = float((dx + 0.5) * scale - 0.5), sx = floor(fx), clamped at
    both borders with the fraction forced to 0; coefficients saturate_cast<short>((1 - fx) * 2048) and (fx * 2048)
    (round half to even); horizontal pass in int32, vertical pass
    ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
  * INTER_LINEAR, uint16: the same coordinates with float32 coefficients, float32 horizontal and vertical passes (multiply,
    then add), saturate_cast<ushort> (round half to even).
  * INTER_LINEAR with an exact 2 x 2 shrink is routed to the INTER_AREA fast path: (s00 + s01 + s10 + s11 + 2) >> 2.
  * INTER_NEAREST: sx = min(floor(dx * scale), W - 1).
Only tests/ and bench tooling import this module; the product path is the HIP kernel `hn_thermal_resize`.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _linear_coords(dsize, ssize):
    """(s0, s1, frac) per destination index: the two taps (s1 == s0 where the border clamps) and the float32 fraction."""
    scale = float(ssize) / float(dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    frac = (f - s.astype(np.float32)).astype(np.float32)
    low, high = s < 0, s >= ssize - 1
    frac[low | high] = 0.0
    s = np.where(low, 0, np.where(high, ssize - 1, s))
    s1 = np.minimum(s + 1, ssize - 1)
    return s, s1, frac


def _coef_short(frac):
    a0 = np.rint((np.float32(1.0) - frac).astype(np.float32) * np.float32(COEF_SCALE)).astype(np.int64)  # rint: half to even
    a1 = np.rint(frac * np.float32(COEF_SCALE)).astype(np.int64)
    return a0, a1


def resize_linear(img, dsize_wh):
    """cv2.resize(img, dsize_wh, interpolation=cv2.INTER_LINEAR) for uint8 (H, W[, C]) and uint16 (H, W) arrays."""
    dw, dh = dsize_wh
    h, w = img.shape[:2]
    if w == 2 * dw and h == 2 * dh:  # exact 2 x 2 shrink: the INTER_AREA fast path
        s = img.astype(np.int64)
        out = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
        return out.astype(img.dtype)
    x0, x1, fx = _linear_coords(dw, w)
    y0, y1, fy = _linear_coords(dh, h)
    if img.dtype == np.uint8:
        a0, a1 = _coef_short(fx)
        b0, b1 = _coef_short(fy)
        src = img.astype(np.int64)
        shape = (1, dw) + (1,) * (img.ndim - 2)
        rows = src[:, x0] * a0.reshape(shape) + src[:, x1] * a1.reshape(shape)  # [H, dw(, C)] int
        r0, r1 = rows[y0], rows[y1]
        bshape = (dh, 1) + (1,) * (img.ndim - 2)
        out = (((b0.reshape(bshape) * (r0 >> 4)) >> 16) + ((b1.reshape(bshape) * (r1 >> 4)) >> 16) + 2) >> 2
        return np.clip(out, 0, 255).astype(np.uint8)
    if img.dtype == np.uint16:
        src = img.astype(np.float32)
        a0, a1 = (np.float32(1.0) - fx).astype(np.float32), fx
        rows = (src[:, x0] * a0[None, :]).astype(np.float32) + (src[:, x1] * a1[None, :]).astype(np.float32)
        rows = rows.astype(np.float32)
        b0, b1 = (np.float32(1.0) - fy).astype(np.float32), fy
        out = (rows[y0] * b0[:, None]).astype(np.float32) + (rows[y1] * b1[:, None]).astype(np.float32)
        return np.clip(np.rint(out.astype(np.float32)), 0, 65535).astype(np.uint16)
    raise TypeError("resize_linear: uint8 or uint16 images")


def resize_nearest(img, dsize_wh):
    """cv2.resize(img, dsize_wh, interpolation=cv2.INTER_NEAREST)."""
    dw, dh = dsize_wh
    h, w = img.shape[:2]
    xs = np.minimum(np.floor(np.arange(dw, dtype=np.float64) * (float(w) / dw)).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(dh, dtype=np.float64) * (float(h) / dh)).astype(np.int64), h - 1)
    return img[ys][:, xs]


def prepare_sample(rgb, ir, label, cols=(150, 850), res=(960, 320)):
    """thermal_loader.py:621-640 (train: cols 150:850) / :783-791 (validation: cols 148:852) for one decoded sample:
    rgb (H, W, 3) uint8 already in RGB order, ir (H, W) uint16, label (H, W) uint8 or None."""
    c0, c1 = cols
    rgb_r = resize_linear(rgb, res)[:, c0:c1]
    ir_r = resize_linear(ir, res)[:, c0:c1]
    lab_r = resize_nearest(label, res)[:, c0:c1] if label is not None else None
    return rgb_r, ir_r, lab_r


def val_tensors(rgb_c, ir_c, normalize=True):
    """ThermalTestDataLoader.__getitem__ after the crop (:793-814): thermal clip to [21800, 25000] and scale to [0, 1],
    rgb / 255, then (x - 0.5) / 0.5 for both when `normalize`.  Returns float32 (rgb [3,H,W], rgb_org, ir [1,H,W], ir_org)."""
    ir = np.clip(ir_c.astype(np.float64), 21800, 25000)
    ir = ((ir - 21800) / (25000 - 21800)).astype(np.float32)[None]
    rgb = (rgb_c.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).astype(np.float32)
    if not normalize:
        return rgb, rgb.copy(), ir, ir.copy()
    return ((rgb - np.float32(0.5)) / np.float32(0.5)).astype(np.float32), rgb, \
        ((ir - np.float32(0.5)) / np.float32(0.5)).astype(np.float32), ir
