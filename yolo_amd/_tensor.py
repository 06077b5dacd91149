"""torch-ROCm plumbing: device memory, current stream, pointers.  No arithmetic happens here."""
import numpy as np
import torch


def device():
    if not torch.cuda.is_available():
        raise RuntimeError("yolo_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def as_dev_f32(x):
    """numpy / torch (any device) -> contiguous float32 tensor on the current GPU."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(device=device(), dtype=torch.float32).contiguous()


def empty(shape, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=device())


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return t.data_ptr() if t is not None else None


def workspace(nbytes):
    """256-byte aligned scratch (torch's caching allocator aligns to >= 512 B)."""
    t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device())
    assert t.data_ptr() % 256 == 0
    return t
