"""CPU: the augmentation oracle (oracle/thermal_aug_ref.py) against outputs of Pillow itself (tests/golden/aug_pillow.npz,
made by tests/golden/make_golden_aug.py) -- uint8 images and labels bit for bit, float32 images to one ulp -- and the
host-side parameter code of heatnet_amd.augment against the oracle's."""
import os

import numpy as np

from oracle import thermal_aug_ref as A


def test_oracle_matches_pillow(golden_dir):
    g = np.load(os.path.join(golden_dir, "aug_pillow.npz"))
    angles = g["angles"]
    for tag in ("a", "sq"):
        rgb, ir, lab = g["rgb_" + tag], g["ir_" + tag], g["lab_" + tag]
        assert np.array_equal(A.hflip(rgb), g["rgb_%s_flip" % tag]) and np.array_equal(A.hflip(ir), g["ir_%s_flip" % tag])
        for k, ang in enumerate(angles):
            assert np.array_equal(A.rotate(rgb, float(ang)), g["rgb_%s_rot%d" % (tag, k)]), (tag, ang)
            assert np.array_equal(A.rotate(lab, float(ang), nearest=True), g["lab_%s_rot%d" % (tag, k)]), (tag, ang)
            assert np.abs(A.rotate(ir, float(ang)) - g["ir_%s_rot%d" % (tag, k)]).max() <= 6e-8, (tag, ang)


def test_ir_normalisation_and_tensor_conventions():
    ir = np.array([[0, 21800, 21801, 23400], [25000, 25001, 65535, 24999]], dtype=np.uint16)
    out = A.ir_clip_normalize(ir)
    assert out.dtype == np.float32 and out[0, 0] == 0.0 and out[0, 1] == 0.0 and out[1, 0] == 1.0 and out[1, 2] == 1.0
    assert out[0, 3] == np.float32(0.5) and out[0, 2] == np.float32(1.0 / 3200.0)
    rgb = np.zeros((2, 2, 3), np.uint8)
    rgb[0, 0] = (255, 0, 128)
    t = A.to_tensor_normalize(rgb)
    assert t.shape == (3, 2, 2) and t[0, 0, 0] == 1.0 and t[1, 0, 0] == -1.0
    assert t[2, 0, 0] == (np.float32(128) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    dropped = A.rect_drop(np.ones((3, 4, 5), np.float32), (1, 2, 2, 2))
    assert dropped[:, 1:3, 2:4].sum() == 0 and dropped.sum() == 3 * (20 - 4)


def test_host_parameters_match_oracle():
    from heatnet_amd import augment
    for (w, h, ang) in ((64, 48, -13.37), (704, 320, 19.999), (53, 37, 0.75)):
        assert augment.rotate_matrix(w, h, ang) == A.rotate_matrix(w, h, ang)
    import random
    ps = augment.draw_params(16, (320, 700), (256, 512), random.Random(3))
    assert all(0 <= p["crop"][0] <= 64 and 0 <= p["crop"][1] <= 188 for p in ps)
    assert all(p["angle"] is None or -20.0 <= p["angle"] <= 20.0 for p in ps)
    assert all(p["drop"][0] + p["drop"][2] <= 256 and p["drop"][1] + p["drop"][3] <= 512 for p in ps)
