#!/usr/bin/env python3
"""Freezes the two natural photographs this container holds as a fixture: scikit-learn's sample images `china.jpg` and `flower.jpg`
(sklearn/datasets/images/, 427 x 640 RGB; Creative Commons BY 2.0, (c) Daniel Buechele / flickr user vultilion, retrieved 2011 by Robert Layton
— see sklearn/datasets/images/README.txt), decoded once with Pillow and stored as raw uint8 arrays so that no JPEG decoder is part of any test.

    python tests/golden/make_natural_images.py        ->  tests/golden/natural_images.npz   {china_rgb, flower_rgb}: [427, 640, 3] uint8

Every other image in tests/ is synthetic (orb_slam2_amd/synth.py); these two are what a camera would deliver: lens blur, JPEG block edges,
foliage, sky gradients, specular highlights.  tests/test_natural_images.py runs the extractor on them (CPU oracle vs the reference's own
ORBextractor.cc vs the HIP kernels) and tools/pin_opencv.py includes them in its OpenCV comparison."""
import os

import numpy as np


def main():
    import sklearn
    from PIL import Image
    src = os.path.join(os.path.dirname(sklearn.__file__), "datasets", "images")
    out = {}
    for name in ("china", "flower"):
        out[name + "_rgb"] = np.asarray(Image.open(os.path.join(src, name + ".jpg")).convert("RGB"), np.uint8)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "natural_images.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
