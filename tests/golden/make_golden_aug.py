"""Golden fixtures for the augmentation oracle (oracle/thermal_aug_ref.py), produced by PILLOW ITSELF -- the third-party
library torchvision's F.hflip / F.rotate forward to for the PIL images of thermal_loader.py:676-712.

    python tests/golden/make_golden_aug.py        (needs Pillow; run in the build container)

Stores seeded inputs and Pillow's outputs for: RGB uint8 bilinear rotation, float32 ('F') bilinear rotation, uint8 ('L')
nearest rotation, horizontal flip -- angles as the loader draws them, (random() - 0.5) * 40, plus the fast-path angles.
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ANGLES = [-20.0, -13.37, -4.25, -0.3, 0.0, 0.75, 7.5, 19.999, 90.0, 180.0, 270.0, 45.0]


def main():
    r = np.random.RandomState(2024)
    out = {"angles": np.array(ANGLES)}
    for tag, (h, w) in (("a", (37, 53)), ("sq", (40, 40))):
        rgb = r.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ir = r.uniform(0, 1, (h, w)).astype(np.float32)
        lab = r.randint(0, 13, (h, w)).astype(np.uint8)
        out["rgb_%s" % tag], out["ir_%s" % tag], out["lab_%s" % tag] = rgb, ir, lab
        out["rgb_%s_flip" % tag] = np.array(Image.fromarray(rgb).transpose(Image.FLIP_LEFT_RIGHT))
        out["ir_%s_flip" % tag] = np.array(Image.fromarray(ir).transpose(Image.FLIP_LEFT_RIGHT))
        for k, ang in enumerate(ANGLES):
            out["rgb_%s_rot%d" % (tag, k)] = np.array(Image.fromarray(rgb).rotate(ang, resample=Image.BILINEAR))
            out["ir_%s_rot%d" % (tag, k)] = np.array(Image.fromarray(ir).rotate(ang, resample=Image.BILINEAR))
            out["lab_%s_rot%d" % (tag, k)] = np.array(Image.fromarray(lab, mode="L").rotate(ang, resample=Image.NEAREST))
    np.savez_compressed(os.path.join(HERE, "aug_pillow.npz"), **out)
    print("wrote aug_pillow.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
