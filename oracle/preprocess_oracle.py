"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of the preprocessing that precedes the network in every
caller of the reference: cv2.resize(img, new_size) -> cv2.cvtColor(BGR2RGB) -> float32 / 255.
(yolo_v3/test_single_image.py:25-28, yolo_v3/video_test.py:61-64; yolo_v2/Detector.py:37-39: `/255.0*2.0-1.0`).

PARITY UNPINNED: cv2 (OpenCV, un-vendored, no pinned version in the reference) is not installed here and the
reference holds no resized fixture, so this restates OpenCV's published 8-bit INTER_LINEAR algorithm
(imgproc/src/resize.cpp: fixed-point coefficients with INTER_RESIZE_COEF_BITS = 11, the uchar VResizeLinear
specialisation, and cv::resize's switch to the 2x2 box mean for an exact 2x reduction).  Pinned only by the
hand-computed cases in tests/test_preprocess.py.  Only tests/ may import this module.
"""
import numpy as np


def _coef(dst_n, src_n, clamp_weights):
    """-> (index int64 [dst_n], c0, c1 int64 [dst_n]) of resize.cpp's xofs/ialpha (clamp_weights) or yofs/ibeta."""
    scale = 1.0 / (float(dst_n) / float(src_n))                       # scale = 1. / inv_scale, in double
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)                  # fx = (float)((dx + 0.5) * scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if clamp_weights:
        lo = s < 0
        s[lo] = 0
        f[lo] = 0
        hi = s >= src_n - 1
        s[hi] = src_n - 1
        f[hi] = 0
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)   # saturate_cast<short>: round half to even
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, c0, c1


def resize_linear_u8(img, new_size):
    """img uint8 [h, w, 3]; new_size (width, height) as cv2.resize takes it -> uint8 [height, width, 3]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    W, H = int(new_size[0]), int(new_size[1])
    if w == 2 * W and h == 2 * H:                                     # INTER_LINEAR -> INTER_AREA fast path
        s = img.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _coef(W, w, True)
    sy, b0, b1 = _coef(H, h, False)
    sx1 = np.minimum(sx + 1, w - 1)
    y0 = np.clip(sy, 0, h - 1)
    y1 = np.clip(sy + 1, 0, h - 1)
    s = img.astype(np.int64)
    hrow = s[:, sx, :] * a0[None, :, None] + s[:, sx1, :] * a1[None, :, None]        # [h, W, 3]
    h0, h1 = hrow[y0], hrow[y1]
    v = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return (v & 255).astype(np.uint8)


def preprocess(img_bgr, new_size, v2=False):
    """-> float32 [1, H, W, 3], the array the reference feeds to the network."""
    img = resize_linear_u8(img_bgr, new_size)[:, :, ::-1]             # BGR -> RGB
    img = np.asarray(img, np.float32)
    if v2:
        return (img / 255.0 * 2.0 - 1.0)[np.newaxis]
    return img[np.newaxis, :] / 255.
