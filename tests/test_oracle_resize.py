"""CPU: the OpenCV-resize restatement (oracle/thermal_resize_ref.py; parity against cv2 unpinned -- cv2 is not installed)
against properties any correct INTER_LINEAR / INTER_NEAREST has, and against an independent float64 bilinear."""
import numpy as np

from oracle import thermal_resize_ref as T


def _float_bilinear(img, dw, dh):
    h, w = img.shape[:2]
    xs = np.clip((np.arange(dw) + 0.5) * (w / dw) - 0.5, 0, w - 1)
    ys = np.clip((np.arange(dh) + 0.5) * (h / dh) - 0.5, 0, h - 1)
    x0, y0 = np.floor(xs).astype(int), np.floor(ys).astype(int)
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    fx, fy = (xs - x0)[None, :], (ys - y0)[:, None]
    s = img.astype(np.float64)
    top = s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def test_identity_and_constant_images():
    r = np.random.RandomState(0)
    a = r.randint(0, 256, (320, 960, 3)).astype(np.uint8)
    assert np.array_equal(T.resize_linear(a, (960, 320)), a)
    b = r.randint(0, 65536, (320, 960)).astype(np.uint16)
    assert np.array_equal(T.resize_linear(b, (960, 320)), b)
    assert np.array_equal(T.resize_nearest(a, (960, 320)), a)
    for v in (0, 7, 255):
        assert (T.resize_linear(np.full((511, 1283), v, np.uint8), (960, 320)) == v).all()
    assert (T.resize_linear(np.full((650, 1920), 23456, np.uint16), (960, 320)) == 23456).all()


def test_close_to_float_bilinear():
    r = np.random.RandomState(1)
    a = r.randint(0, 256, (650, 1777)).astype(np.uint8)
    d = T.resize_linear(a, (960, 320)).astype(np.float64) - _float_bilinear(a, 960, 320)
    assert np.abs(d).max() <= 1.0  # 11-bit coefficients + two truncations: within one grey level
    b = r.randint(20000, 27000, (650, 1777)).astype(np.uint16)
    d = T.resize_linear(b, (960, 320)).astype(np.float64) - _float_bilinear(b, 960, 320)
    assert np.abs(d).max() <= 1.0  # float32 source coordinates: ~6e-5 of the neighbour difference (<= 7000 counts) + rounding


def test_area_fast_path_and_nearest():
    r = np.random.RandomState(2)
    a = r.randint(0, 256, (640, 1920)).astype(np.int64)
    want = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(T.resize_linear(a.astype(np.uint8), (960, 320)), want.astype(np.uint8))
    lab = r.randint(0, 13, (650, 1920)).astype(np.uint8)
    out = T.resize_nearest(lab, (960, 320))
    assert np.array_equal(out, lab[np.floor(np.arange(320) * 650 / 320).astype(int)][:, 2 * np.arange(960)])
    rgb_c, ir_c, lab_c = T.prepare_sample(np.zeros((650, 1920, 3), np.uint8), np.zeros((650, 1920), np.uint16), lab)
    assert rgb_c.shape == (320, 700, 3) and ir_c.shape == (320, 700) and np.array_equal(lab_c, out[:, 150:850])
