"""GPU counterpart of the preprocessing lines of the reference's drivers:

    img = cv2.resize(img_ori, tuple(new_size)); img = cv2.cvtColor(img, cv2.COLOR_BGR2RGB)
    img = np.asarray(img, np.float32); img = img[np.newaxis, :] / 255.          (test_single_image.py:25-28)

as one kernel (yolo_preprocess) over a batch of equally sized uint8 BGR frames.  Parity with cv2 itself is unpinned
(cv2 is not installed; see csrc/preprocess.hip)."""
import numpy as np
import torch

from . import _lib, _tensor


def preprocess(images_bgr, new_size, v2=False):
    """images_bgr: uint8 [h, w, 3] or [B, h, w, 3] (numpy or torch, host or device), BGR as cv2.imread returns it;
    new_size: (width, height) as cv2.resize takes it -> float32 device tensor [B, height, width, 3], RGB,
    in [0, 1] (v2=True: [-1, 1], yolo_v2/Detector.py:39)."""
    if isinstance(images_bgr, np.ndarray):
        if images_bgr.dtype != np.uint8:
            raise ValueError("preprocess: uint8 images expected, got %s" % images_bgr.dtype)
        t = torch.from_numpy(np.ascontiguousarray(images_bgr))
    else:
        t = images_bgr
        if t.dtype != torch.uint8:
            raise ValueError("preprocess: uint8 images expected, got %s" % t.dtype)
    if t.dim() == 3:
        t = t[None]
    if t.dim() != 4 or t.shape[3] != 3:
        raise ValueError("preprocess: images must be [h, w, 3] or [B, h, w, 3], got %s" % (tuple(t.shape),))
    t = t.to(_tensor.device(), non_blocking=True).contiguous()
    B, h, w = int(t.shape[0]), int(t.shape[1]), int(t.shape[2])
    W, H = int(new_size[0]), int(new_size[1])
    out = _tensor.empty((B, H, W, 3))
    _lib.check(_lib.lib().yolo_preprocess(t.data_ptr(), B, h, w, 0, out.data_ptr(), H, W, 1 if v2 else 0,
                                          _tensor.stream_ptr()))
    return out
