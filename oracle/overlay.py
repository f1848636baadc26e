"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- never imported by the product package.

Restatement of Tracking::CreateCurrentMatchImage (slam_pipeline/src/Tracking.cc:899-940): cvtColor(GRAY2RGB) of the two
frames side by side, then cv::circle(..., radius 3, FILLED) per match end point, green first, map-point matches second.
cv::circle is OpenCV's Circle() (imgproc/src/drawing.cpp), restated below from the upstream source as recalled (OpenCV is
absent from this image): the midpoint loop filling horizontal spans, clipped to the image.  Parity: unpinned."""
import numpy as np


def _circle_filled(img, cx, cy, radius, color):
    h, w = img.shape[:2]

    def hline(y, x0, x1):
        if 0 <= y < h:
            x0, x1 = max(x0, 0), min(x1, w - 1)
            if x0 <= x1:
                img[y, x0:x1 + 1] = color
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        y11, y12, y21, y22 = cy - dy, cy + dy, cy - dx, cy + dx
        x11, x12, x21, x22 = cx - dx, cx + dx, cx - dy, cx + dy
        hline(y11, x11, x12)
        hline(y12, x11, x12)
        hline(y21, x21, x22)
        hline(y22, x21, x22)
        dy += 1
        err += plus
        plus += 2
        mask = 0 if err <= 0 else -1
        err -= minus & mask
        dx += mask
        minus -= mask & 2


def create_current_match_image(f1, f2, matches, has_mp1, has_mp2):
    h, w = f1.shape
    out = np.zeros((h, 2 * w, 3), np.uint8)
    out[:, :w] = f1[:, :, None]          # cvtColor(GRAY2RGB)
    out[:, w:] = f2[:, :, None]
    img0, img1 = out[:, :w], out[:, w:]
    for want, color in ((False, (0, 255, 0)), (True, (255, 0, 0))):
        for i, (x1, y1, x2, y2) in enumerate(np.asarray(matches).reshape(-1, 4)):
            has = bool(has_mp1[i]) or bool(has_mp2[i])
            if has == want:
                _circle_filled(img0, int(x1), int(y1), 3, color)
                _circle_filled(img1, int(x2), int(y2), 3, color)
    return out
