"""Frame-batch pipeline for callers whose images start in host memory (the per-frame loop of the reference's
video_test.py:58-71, many frames at a time): the uint8 batch i+1 is copied to the GPU on a copy stream while
yolo_preprocess + the detection step of batch i run on the compute stream, so the PCIe transfer (3 bytes per source
pixel) disappears behind the compute.  Host staging buffers are pinned and double-buffered; all ordering is by
HIP events, nothing synchronises the device inside the loop except the reuse of a staging buffer."""
import numpy as np
import torch

from . import _tensor
from .preprocess import preprocess


class DetectPipeline(object):

    def __init__(self, yolo_model, new_size=(416, 416), max_boxes=30, score_thresh=0.4, iou_thresh=0.5, depth=2):
        """Thresholds default to test_single_image.py:39; depth = staging slots (2 = copy one batch ahead)."""
        if depth < 2:
            raise ValueError("DetectPipeline needs at least 2 staging slots")
        self.model, self.new_size = yolo_model, (int(new_size[0]), int(new_size[1]))
        self.args = (int(max_boxes), float(score_thresh), float(iou_thresh))
        self.depth = int(depth)
        self._pinned = [None] * self.depth
        self._dev = [None] * self.depth
        self._copied = [None] * self.depth       # event: H2D copy into slot k finished
        self._consumed = [None] * self.depth     # event: the compute stream is done reading slot k
        self._copy_stream = None

    def _slot(self, k, shape):
        if self._copy_stream is None:
            _tensor.device()
            self._copy_stream = torch.cuda.Stream()
        if self._pinned[k] is None or tuple(self._pinned[k].shape) != tuple(shape):
            if self._copied[k] is not None:
                self._copied[k].synchronize()
            self._pinned[k] = torch.empty(shape, dtype=torch.uint8).pin_memory()
            old_consumed = self._consumed[k]
            self._dev[k] = torch.empty(shape, dtype=torch.uint8, device=_tensor.device())
            # the caching allocator may hand back a block that kernels already enqueued on the compute stream still read
            # (a just-freed preprocess / detect tensor, the previous buffer of this slot): the first copy into the new
            # buffer must order after all of them
            self._copy_stream.wait_stream(torch.cuda.current_stream())
            if old_consumed is not None:
                self._copy_stream.wait_event(old_consumed)
            self._copied[k] = torch.cuda.Event()
            self._consumed[k] = None

    def _upload(self, k, frames):
        """Start the H2D copy of one batch into slot k on the copy stream."""
        if isinstance(frames, torch.Tensor):
            if frames.dtype != torch.uint8:
                raise ValueError("DetectPipeline: uint8 frames expected, got %s" % frames.dtype)
            src = frames
        else:
            src = np.asarray(frames)
            if src.dtype != np.uint8:
                raise ValueError("DetectPipeline: uint8 frames expected, got %s" % src.dtype)
        if src.ndim != 4 or src.shape[3] != 3:
            raise ValueError("DetectPipeline: frames must be [B, h, w, 3], got %s" % (tuple(src.shape),))
        self._slot(k, tuple(src.shape))
        if isinstance(src, torch.Tensor) and src.is_pinned():
            host = src                                           # caller already staged it: no host copy
        else:
            self._copied[k].synchronize()                        # the previous copy out of this staging buffer is done
            host = self._pinned[k]
            host.copy_(src if isinstance(src, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(src)))
        cs = self._copy_stream
        if self._consumed[k] is not None:
            cs.wait_event(self._consumed[k])                     # the device buffer is free again
        with torch.cuda.stream(cs):
            self._dev[k].copy_(host, non_blocking=True)
            self._copied[k].record(cs)

    def _compute(self, k):
        cur = torch.cuda.current_stream()
        cur.wait_event(self._copied[k])
        x = preprocess(self._dev[k], self.new_size)
        ev = torch.cuda.Event()
        ev.record(cur)                                           # preprocess has read the uint8 buffer
        self._consumed[k] = ev
        return self.model.detect(x, *self.args)

    def run(self, batches):
        """batches: iterable of uint8 [B, h, w, 3] BGR frame batches (numpy or torch, host memory) -> yields, per batch and in
        order, the device tensors of yolov3.detect (boxes in network pixels; rescale as test_single_image.py:42-45)."""
        it = iter(batches)
        try:
            first = next(it)
        except StopIteration:
            return
        k = 0
        self._upload(k, first)
        for nxt in it:
            self._upload((k + 1) % self.depth, nxt)              # copy of batch i+1 ...
            yield self._compute(k)                               # ... under the compute of batch i
            k = (k + 1) % self.depth
        yield self._compute(k)
