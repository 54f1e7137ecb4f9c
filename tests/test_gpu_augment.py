"""GPU: hn_thermal_augment (heatnet_amd.augment.augment_batch) against the CPU oracle of the loader's per-sample path
(oracle/thermal_aug_ref.py, pinned against Pillow).  Labels and everything derived from uint8 pixels must agree exactly;
the float32 thermal channel to one ulp of its [0, 1] value (the oracle blends in float64 and rounds once, like Pillow)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def _sample_batch(n, h0, w0, seed):
    r = np.random.RandomState(seed)
    rgb = r.randint(0, 256, (n, h0, w0, 3)).astype(np.uint8)
    ir = r.randint(20000, 27000, (n, h0, w0)).astype(np.uint16)   # below, inside and above the [21800, 25000] window
    lab = r.randint(0, 13, (n, h0, w0)).astype(np.uint8)
    return rgb, ir, lab


@pytest.mark.parametrize("geom", [((40, 60), (32, 48)), ((37, 53), (37, 53)), ((64, 64), (33, 47))])
def test_augment_matches_oracle(dev, geom):
    from heatnet_amd import augment
    from oracle import thermal_aug_ref as A
    (h0, w0), (h, w) = geom
    n = 10
    rgb, ir, lab = _sample_batch(n, h0, w0, seed=h0 * 100 + w)
    rng = random.Random(5)
    params = augment.draw_params(n, (h0, w0), (h, w), rng)
    for p in params:  # small images: keep the drop rectangle inside, as the loader's RandomCrop.get_params would
        p["drop"] = (rng.randint(0, h // 2), rng.randint(0, w // 2), h // 3, w // 3)
    params[0].update(flip=False, angle=None)
    params[1].update(flip=True, angle=None)
    params[2].update(flip=True, angle=-20.0)
    params[3].update(flip=False, angle=180.0)
    params[4].update(flip=False, angle=0.0)
    for target in (None, "rgb", "ir"):
        got_rgb, got_ir, got_lab = augment.augment_batch(torch.from_numpy(rgb).to(dev), torch.from_numpy(ir).to(dev),
                                                        torch.from_numpy(lab).to(dev), (h, w), params, drop_target=target)
        assert got_rgb.shape == (n, 3, h, w) and got_ir.shape == (n, 1, h, w) and got_lab.dtype == torch.int64
        for k, p in enumerate(params):
            er, ei, el = A.augment_sample(rgb[k], ir[k], lab[k], p["crop"] + (h, w), p["flip"], p["angle"],
                                          drop=p["drop"] if target else None, drop_rgb=target == "rgb")
            assert np.array_equal(got_lab[k].cpu().numpy(), el.astype(np.int64)), (k, p)
            assert np.array_equal(got_rgb[k].cpu().numpy(), er), (k, p)
            assert np.abs(got_ir[k].cpu().numpy() - ei).max() <= 2.5e-7, (k, p)


def test_thermal_intensity_augmentations(dev):
    """`ir_day = scale * ir_day` and smartAugment (train_trgb_segnet_conf.py:101-110, 402-410) folded into the same pass."""
    from heatnet_amd import augment
    from oracle import thermal_aug_ref as A
    rgb, ir, lab = _sample_batch(3, 24, 32, seed=9)
    params = [dict(crop=(0, 0), flip=False, angle=None, ir_scale=0.37), dict(crop=(1, 2), flip=True, angle=5.0),
              dict(crop=(0, 0), flip=False, angle=-11.0, ir_scale=0.9)]
    cls = [0.1 + 0.07 * k for k in range(13)]
    _, got_ir, got_lab = augment.augment_batch(torch.from_numpy(rgb).to(dev), torch.from_numpy(ir).to(dev),
                                               torch.from_numpy(lab).to(dev), (20, 28), params, class_scale=cls)
    for k, p in enumerate(params):
        _, ei, el = A.augment_sample(rgb[k], ir[k], lab[k], p["crop"] + (20, 28), p["flip"], p["angle"])
        want = A.smart_augment(ei * np.float32(p.get("ir_scale", 1.0)), el, cls)
        assert np.array_equal(got_lab[k].cpu().numpy(), el.astype(np.int64))
        assert np.abs(got_ir[k].cpu().numpy() - want).max() <= 3e-7, k


def test_augment_without_labels_and_argument_checks(dev):
    from heatnet_amd import augment
    rgb, ir, _ = _sample_batch(2, 20, 24, seed=1)
    params = [dict(crop=(0, 0), flip=False, angle=3.0, drop=None), dict(crop=(2, 4), flip=True, angle=None, drop=None)]
    r, i, l = augment.augment_batch(torch.from_numpy(rgb).to(dev), torch.from_numpy(ir).to(dev), None, (16, 20), params)
    assert l is None and r.shape == (2, 3, 16, 20) and float(r.abs().max()) <= 1.0 and float(i.abs().max()) <= 1.0
    with pytest.raises(RuntimeError):
        augment.augment_batch(torch.from_numpy(rgb), torch.from_numpy(ir), None, (16, 20), params)       # CPU tensors
    with pytest.raises(RuntimeError, match="invalid argument"):
        augment.augment_batch(torch.from_numpy(rgb).to(dev), torch.from_numpy(ir).to(dev), None, (32, 20), params)  # h > H0


@pytest.mark.parametrize("src", [(650, 1920), (640, 1920), (512, 1280), (320, 960), (301, 977)], ids=str)
@pytest.mark.parametrize("cols", [(150, 850), (148, 852)], ids=str)
def test_resize_crop_matches_oracle_bit_exact(dev, src, cols):
    """hn_thermal_resize == oracle/thermal_resize_ref.py (OpenCV's INTER_LINEAR / INTER_NEAREST restated; parity against
    cv2 itself is unpinned): integer results, compared exactly -- uint8 fixed-point path, uint16 float path, the 2x2 area
    fast path (640x1920), identity (320x960) and ragged source sizes."""
    from heatnet_amd import augment
    from oracle import thermal_resize_ref as T
    h0, w0 = src
    rgb, ir, lab = _sample_batch(2, h0, w0, seed=h0 + w0)
    got_rgb, got_ir, got_lab = augment.resize_crop(torch.from_numpy(rgb).to(dev), torch.from_numpy(ir).to(dev),
                                                   torch.from_numpy(lab).to(dev), cols=cols)
    assert got_rgb.shape == (2, 320, cols[1] - cols[0], 3) and got_ir.shape == (2, 320, cols[1] - cols[0])
    for k in range(2):
        er, ei, el = T.prepare_sample(rgb[k], ir[k], lab[k], cols=cols)
        assert np.array_equal(got_rgb[k].cpu().numpy(), er)
        assert np.array_equal(got_ir[k].cpu().numpy().astype(np.uint16), ei)
        assert np.array_equal(got_lab[k].cpu().numpy(), el)


def test_validation_loader_batch(dev):
    """augment.val_batch == ThermalTestDataLoader.__getitem__ after decoding (cm/thermal_loader.py:777-814), restated in
    oracle/thermal_resize_ref.py: 960x320, columns 148:852, thermal clip / scale, /255, (x - 0.5) / 0.5."""
    from heatnet_amd import augment
    from oracle import thermal_resize_ref as T
    rgb, ir, lab = _sample_batch(3, 650, 1920, seed=77)
    out = augment.val_batch(torch.from_numpy(rgb).to(dev), torch.from_numpy(ir).to(dev), torch.from_numpy(lab).to(dev))
    assert out["rgb"].shape == (3, 3, 320, 704) and out["ir"].shape == (3, 1, 320, 704) and out["label"].dtype == torch.int64
    for k in range(3):
        er, ei, el = T.prepare_sample(rgb[k], ir[k], lab[k], cols=(148, 852))
        r, r_org, i, i_org = T.val_tensors(er, ei)
        assert np.array_equal(out["label"][k].cpu().numpy(), el.astype(np.int64))
        assert np.array_equal(out["rgb"][k].cpu().numpy(), r)
        assert np.abs(out["rgb_org"][k].cpu().numpy() - r_org).max() <= 6e-8  # display copy: a stock-torch division
        assert np.abs(out["ir"][k].cpu().numpy() - i).max() <= 2.5e-7
        assert np.abs(out["ir_org"][k].cpu().numpy() - i_org).max() <= 1.2e-7


def test_resize_rejects_bad_windows(dev):
    from heatnet_amd import augment
    rgb, ir, lab = _sample_batch(1, 64, 96, seed=1)
    with pytest.raises(RuntimeError, match="hn_thermal_resize"):
        augment.resize_crop(torch.from_numpy(rgb).to(dev), torch.from_numpy(ir).to(dev), None, cols=(900, 1000))
