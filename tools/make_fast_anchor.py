#!/opt/conda/bin/python3.9
"""Third-party anchor for the FAST-9 segment decision: scikit-image's corner_fast (its own Cython FAST, n = 9) on the
synthetic textures -> tests/golden/fast9_skimage.npz.  Run with an interpreter that can import skimage (this container:
/opt/conda/bin/python3.9 with /opt/conda/lib/python3.9/site-packages on sys.path); the images come in as a .npz made by
the caller (tests/test_fast_anchor.py or `python tools/make_fast_anchor.py --make`), so this script needs nothing of
the repository.

skimage converts uint8 input to float (p / 255) and tests ring > centre + t and ring < centre - t in double; with
t = 20.5 / 255 that is `difference >= 21` = cv::FAST's strict `> 20` for every integer difference, rounding included."""
import sys

import numpy as np


def corner_sets(images):
    sys.path.insert(0, "/opt/conda/lib/python3.9/site-packages")
    from skimage.feature import corner_fast
    out = {}
    for name, img in images.items():
        resp = corner_fast(img, n=9, threshold=20.5 / 255.0)
        out[name] = np.packbits(resp > 0)
        out[name + "_shape"] = np.array(img.shape)
    return out


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    imgs = dict(np.load(src))
    import skimage
    res = corner_sets(imgs)
    res["skimage_version"] = np.array(skimage.__version__)
    np.savez_compressed(dst, **res)
