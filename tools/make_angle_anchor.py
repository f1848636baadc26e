#!/opt/conda/bin/python3.9
"""Third-party anchor for the ORB orientation: scikit-image's corner_orientations -- its own
Cython intensity-centroid routine -- with its ORB disc mask OFAST_MASK (749 pixels, the umax table of cv::ORB) at given
corner positions of given level images -> tests/golden/ic_angle_skimage.npz.  Run with an interpreter that can import
skimage (this container: /opt/conda/bin/python3.9); images and corner positions come in as a .npz made by the caller
(tests/test_angle_resize_anchor.py), so this script needs nothing of the repository.

skimage works on the float image p / 255 and returns atan2(m01, m10) in radians; the orientation does not depend on the
scale of the intensities."""
import sys

import numpy as np


def angles(data):
    sys.path.insert(0, "/opt/conda/lib/python3.9/site-packages")
    from skimage.feature import corner_orientations
    from skimage.feature.orb import OFAST_MASK
    out = {}
    for key in [k for k in data if k.endswith("_img")]:
        name = key[:-4]
        img = data[key].astype(np.float64) / 255.0
        rc = data[name + "_rc"].astype(np.intp)                      # (row, col) per corner
        out[name + "_rad"] = corner_orientations(img, rc, OFAST_MASK).astype(np.float64)
        out[name + "_rc"] = rc.astype(np.int32)
    out["mask_sum"] = np.array(OFAST_MASK.sum())
    return out


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    res = angles(dict(np.load(src)))
    import skimage
    res["skimage_version"] = np.array(skimage.__version__)
    np.savez_compressed(dst, **res)
