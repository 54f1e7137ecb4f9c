"""Batch augmentation of the thermal data loader on the GPU.

Replaces, for a batch of already resized samples held as uint8 / uint16 device tensors, the per-sample CPU work of
`ThermalDataLoader.__getitem__` (models/confusion_maximization/thermal_loader.py:642-735) and the trainer's
`rectDropTensor` (train_trgb_segnet_conf.py:82-86) by ONE kernel launch (`hn_thermal_augment`): random crop, thermal
clip / scale, horizontal flip, rotation (Pillow's bilinear / nearest affine transforms, bit-exact), to_tensor,
normalisation to [-1, 1], modality block drop.  `draw_params` draws the random decisions with the loader's own
distributions; `augment_batch` applies given decisions.  `resize_crop` is the step in front of it -- cv2.resize to 960x320
and the column crop (:621-640) -- and `val_batch` the whole of ThermalTestDataLoader.__getitem__ after decoding (:777-814).
Image decoding (cv2.imread) stays on the host.
"""
import ctypes
import math
import random

import torch

from . import _lib
from ._lib import check


class HnAugSample(ctypes.Structure):
    _fields_ = [("crop_i", ctypes.c_int32), ("crop_j", ctypes.c_int32), ("flip", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("m", ctypes.c_double * 6), ("fx", ctypes.c_longlong * 6),
                ("drop_i", ctypes.c_int32), ("drop_j", ctypes.c_int32), ("drop_h", ctypes.c_int32),
                ("drop_w", ctypes.c_int32), ("drop_target", ctypes.c_int32), ("ir_scale", ctypes.c_float)]

AUG_CLASSES = 32  # HN_AUG_CLASSES


def rotate_matrix(w, h, angle_deg):
    """PIL.Image.rotate's inverse affine matrix for expand=False, center=None (Pillow Image.py)."""
    angle = -math.radians(angle_deg % 360.0)
    a, b = round(math.cos(angle), 15), round(math.sin(angle), 15)
    d, e = round(-math.sin(angle), 15), round(math.cos(angle), 15)
    cx, cy = w / 2, h / 2
    return a, b, a * -cx + b * -cy + 0.0 + cx, d, e, d * -cx + e * -cy + 0.0 + cy


def _fixed(matrix):
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    a, b, c, d, e, f = matrix
    return fix(a), fix(b), fix(c + a * 0.5 + b * 0.5), fix(d), fix(e), fix(f + d * 0.5 + e * 0.5)


def draw_params(n, src_hw, out_hw, rng=random):
    """The loader's random decisions for n samples (thermal_loader.py:642, 663-667, 690-712): crop corner uniform over
    the valid range, flip with probability 1/2, rotation with probability 1/2 by (u - 0.5) * 40 degrees, drop rectangle of
    int(U(100, 300)) x int(U(100, 500)) pixels placed like a random crop."""
    (h0, w0), (h, w) = src_hw, out_hw
    out = []
    for _ in range(n):
        i = rng.randint(0, h0 - h) if h0 > h else 0
        j = rng.randint(0, w0 - w) if w0 > w else 0
        dh, dw = min(int(rng.uniform(100, 300)), h), min(int(rng.uniform(100, 500)), w)
        di, dj = rng.randint(0, h - dh), rng.randint(0, w - dw)
        flip = rng.random() > 0.5
        angle = (rng.random() - 0.5) * 40 if rng.random() > 0.5 else None
        out.append(dict(crop=(i, j), flip=flip, angle=angle, drop=(di, dj, dh, dw)))
    return out


def augment_batch(rgb_u8, ir_u16, label_u8, out_hw, params, drop_target=None, class_scale=None):
    """rgb_u8 [N,H0,W0,3] uint8, ir_u16 [N,H0,W0] uint16 (or int32 holding 16-bit counts), label_u8 [N,H0,W0] uint8 or
    None, all on the GPU; `params`: list of dicts(crop=(i,j), flip=bool, angle=deg|None, drop=(i,j,h,w)|None);
    `drop_target`: None, "rgb" or "ir" (which tensor the block drop zeroes); a dict may carry `ir_scale` (the trainer's
    `ir_day = scale * ir_day`, train_trgb_segnet_conf.py:402-405); `class_scale`: per-class factors on the thermal tensor
    (smartAugment, :101-110; needs labels).  Returns (rgb [N,3,h,w], ir [N,1,h,w],
    label [N,h,w] int64 or None) -- logical NCHW over NHWC memory, what the network's first convolutions consume."""
    if not rgb_u8.is_cuda:
        raise RuntimeError("augment_batch runs on the MI355X only; there is no CPU fallback")
    n, h0, w0, _ = rgb_u8.shape
    h, w = out_hw
    if rgb_u8.dtype != torch.uint8 or ir_u16.dtype not in (torch.uint16, torch.int16):
        raise RuntimeError("augment_batch: rgb must be uint8 and ir uint16 (raw thermal counts)")
    if len(params) != n:
        raise ValueError("augment_batch: one parameter set per sample")
    arr = (HnAugSample * n)()
    for k, p in enumerate(params):
        e = arr[k]
        e.crop_i, e.crop_j = p["crop"]
        e.flip = 1 if p.get("flip") else 0
        angle = p.get("angle")
        e.mode = 0
        if angle is not None:
            a360 = angle % 360.0
            if a360 == 0:
                e.mode = 0
            elif a360 == 180:
                e.mode = 2
            elif a360 in (90, 270) and h == w:
                raise ValueError("augment_batch: Pillow's 90/270-degree transpose path on square crops is not implemented")
            else:
                e.mode = 1
                m = rotate_matrix(w, h, angle)
                e.m[:] = m
                e.fx[:] = _fixed(m)
        e.ir_scale = float(p.get("ir_scale", 1.0))
        drop = p.get("drop") if drop_target else None
        if drop is not None:
            e.drop_i, e.drop_j, e.drop_h, e.drop_w = drop
            e.drop_target = 1 if drop_target == "rgb" else 2
    dev = rgb_u8.device
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    pdev = host.to(dev)
    rgb = torch.empty((n, h, w, 3), device=dev, dtype=torch.float32)
    ir = torch.empty((n, h, w, 1), device=dev, dtype=torch.float32)
    lab = torch.empty((n, h, w), device=dev, dtype=torch.int64) if label_u8 is not None else None
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    cs = None
    if class_scale is not None:
        vals = list(class_scale) + [1.0] * (AUG_CLASSES - len(class_scale))
        cs = torch.tensor(vals[:AUG_CLASSES], dtype=torch.float32).to(dev)
    check(_lib.load().hn_thermal_augment(ptr(rgb_u8.contiguous()), ptr(ir_u16.contiguous()),
                                         ptr(label_u8.contiguous()) if label_u8 is not None else None, n, h0, w0, ptr(pdev),
                                         ptr(cs), h, w, ptr(rgb), ptr(ir), ptr(lab),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "hn_thermal_augment")
    return rgb.permute(0, 3, 1, 2), ir.permute(0, 3, 1, 2), lab


TRAIN_COLS, VAL_COLS, NET_RES = (150, 850), (148, 852), (960, 320)  # thermal_loader.py:635-640, 789-791, 622 / 784


def resize_crop(rgb_u8, ir_u16, label_u8, cols=TRAIN_COLS, res=NET_RES):
    """Decoded samples -> loader resolution: rgb_u8 [N,H0,W0,3] uint8 (RGB order), ir_u16 [N,H0,W0] uint16, label_u8
    [N,H0,W0] uint8 or None (GPU tensors) -> the same three at [N, res[1], cols[1]-cols[0]]: cv2.resize(INTER_LINEAR /
    INTER_NEAREST for the labels) to `res` = (width, height) and the column crop (cm/thermal_loader.py:621-640)."""
    if not rgb_u8.is_cuda:
        raise RuntimeError("resize_crop runs on the MI355X only; there is no CPU fallback")
    if rgb_u8.dtype != torch.uint8 or ir_u16.dtype not in (torch.uint16, torch.int16):
        raise RuntimeError("resize_crop: rgb must be uint8 and ir uint16 (raw thermal counts)")
    n, h0, w0, _ = rgb_u8.shape
    dw, dh = res
    c0, c1 = cols
    dev = rgb_u8.device
    rgb = torch.empty((n, dh, c1 - c0, 3), device=dev, dtype=torch.uint8)
    ir = torch.empty((n, dh, c1 - c0), device=dev, dtype=ir_u16.dtype)
    lab = torch.empty((n, dh, c1 - c0), device=dev, dtype=torch.uint8) if label_u8 is not None else None
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    check(_lib.load().hn_thermal_resize(ptr(rgb_u8.contiguous()), ptr(ir_u16.contiguous()),
                                        ptr(label_u8.contiguous()) if label_u8 is not None else None, n, h0, w0, dh, dw, c0, c1,
                                        ptr(rgb), ptr(ir), ptr(lab),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "hn_thermal_resize")
    return rgb, ir, lab


def val_batch(rgb_u8, ir_u16, label_u8, normalize=True):
    """ThermalTestDataLoader.__getitem__ (cm/thermal_loader.py:777-814) for a batch of decoded samples on the GPU: resize to
    960x320, columns 148:852, thermal clip / scale, to_tensor, normalisation.  Returns the loader's dict: `rgb` [N,3,320,704],
    `ir` [N,1,320,704] (normalised to [-1, 1] when `normalize`), `rgb_org`, `ir_org` (the [0, 1] tensors the loader keeps for
    visualisation) and `label` [N,320,704] int64."""
    rgb_c, ir_c, lab_c = resize_crop(rgb_u8, ir_u16, label_u8, cols=VAL_COLS)
    n, h, w, _ = rgb_c.shape
    identity = [dict(crop=(0, 0), flip=False, angle=None)] * n
    rgb, ir, lab = augment_batch(rgb_c, ir_c, lab_c, (h, w), identity)  # clip / scale / to_tensor / (x - 0.5) / 0.5
    # the un-normalised copies are display data: x_org = the same [0, 1] values before (x - 0.5) / 0.5
    rgb_org = rgb_c.permute(0, 3, 1, 2).float() / 255.0
    ir_org = ((ir_c.to(torch.int32) & 0xFFFF).clamp(21800, 25000).double() - 21800.0).div(3200.0).float().unsqueeze(1)
    if not normalize:
        rgb, ir = rgb_org.clone(), ir_org.clone()
    return {"rgb": rgb, "rgb_org": rgb_org, "ir": ir, "ir_org": ir_org, "label": lab}
