"""Image-sharded multi-GPU use of the hot path: one process per GPU (torch.distributed), weights
replicated, contiguous batch shards, NO collective inside the hot path (images are independent).
The only exchange is the optional final all-gather of the padded detection tensors (RCCL when the
process group backend is "nccl"; "gloo" in the CPU tests)."""
import torch


def shard_bounds(n_images, rank, world):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one, earlier ranks get the extras."""
    if world <= 0 or not (0 <= rank < world) or n_images < 0:
        raise ValueError("bad shard request: n=%d rank=%d world=%d" % (n_images, rank, world))
    q, r = divmod(n_images, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_detections(boxes, scores, labels, counts):
    """[n,R,4], [n,R], [n,R] int32, [n] int32 -> one float32 wire tensor [n, R, 6] = (x0,y0,x1,y1,score,label);
    rows r >= counts[i] are zeroed so the wire format is deterministic."""
    n, R = scores.shape
    valid = (torch.arange(R, device=scores.device)[None, :] < counts[:, None].to(torch.int64))
    wire = torch.cat([boxes, scores[..., None], labels.to(torch.float32)[..., None]], dim=2)
    return wire * valid[..., None].to(wire.dtype)


def pack_detections_device(boxes, scores, labels, counts, scale_xy=None, image_base=0):
    """The device-side wire format (yolo_pack_detections): outputs of yolov3.detect / gpu_nms_batched ->
    (rows [m, 7] float32 = (image index, x0, y0, x1, y1, score, label), offsets [n+1] int32); rows of image i are
    rows[offsets[i]:offsets[i+1]].  scale_xy [n, 2] = (width_ori / new_w, height_ori / new_h) applies the box rescale of
    test_single_image.py:42-45 on the GPU; image_base = global index of the first local image (a shard's `lo`)."""
    from . import _lib, _tensor
    n, cap = int(scores.shape[0]), int(scores.shape[1])
    boxes, scores = _tensor.as_dev_f32(boxes), _tensor.as_dev_f32(scores)
    labels = labels.to(device=boxes.device, dtype=torch.int32).contiguous()
    counts = counts.to(device=boxes.device, dtype=torch.int32).contiguous()
    sc = None if scale_xy is None else _tensor.as_dev_f32(scale_xy).reshape(n, 2)
    rows = _tensor.empty((n * cap, 7))
    offsets = _tensor.empty((n + 1,), torch.int32)
    _lib.check(_lib.lib().yolo_pack_detections(boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(), counts.data_ptr(), n, cap,
                                               _tensor.ptr(sc), int(image_base), rows.data_ptr(), offsets.data_ptr(),
                                               _tensor.stream_ptr()))
    return rows[:int(offsets[-1].item())], offsets


def all_gather_rows(rows, offsets, n_images, group=None):
    """Gather every rank's packed rows (pack_detections_device with image_base = the shard's first image) into the global
    row list of the whole batch, in image order: -> (rows [M, 7], offsets [n_images + 1]).  Row counts differ per rank,
    so each rank pads to the largest count before the collective (RCCL when the backend is "nccl")."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    m = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    ms = [torch.empty_like(m) for _ in range(world)]
    dist.all_gather(ms, m, group=group)
    ms = [int(t.item()) for t in ms]
    cap = max(max(ms), 1)
    pad = torch.zeros((cap, 7), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    per = [shard_bounds(n_images, r, world) for r in range(world)]
    cnt = offsets[1:] - offsets[:-1]
    ncap = max(hi - lo for lo, hi in per)
    cpad = torch.zeros((ncap,), dtype=cnt.dtype, device=cnt.device)
    cpad[:cnt.shape[0]] = cnt
    cparts = [torch.empty_like(cpad) for _ in range(world)]
    dist.all_gather(cparts, cpad, group=group)
    all_rows = torch.cat([parts[r][:ms[r]] for r in range(world)], dim=0)
    all_cnt = torch.cat([cparts[r][:hi - lo] for r, (lo, hi) in enumerate(per)], dim=0)
    all_off = torch.zeros((n_images + 1,), dtype=torch.int64, device=rows.device)
    all_off[1:] = torch.cumsum(all_cnt.to(torch.int64), 0)
    return all_rows, all_off


def all_gather_detections(wire, counts, n_images, group=None):
    """Gather every rank's [n_local, R, 6] + counts into the global [n_images, R, 6] / [n_images] order.
    Shards may differ by one image, so each rank pads to the largest shard before the collective."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = [shard_bounds(n_images, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in per)
    R = wire.shape[1]
    w = torch.zeros((cap, R, 6), dtype=wire.dtype, device=wire.device)
    c = torch.zeros((cap,), dtype=counts.dtype, device=counts.device)
    w[:wire.shape[0]] = wire
    c[:counts.shape[0]] = counts
    ws = [torch.empty_like(w) for _ in range(world)]
    cs = [torch.empty_like(c) for _ in range(world)]
    dist.all_gather(ws, w, group=group)
    dist.all_gather(cs, c, group=group)
    out_w = torch.cat([ws[r][:hi - lo] for r, (lo, hi) in enumerate(per)], dim=0)
    out_c = torch.cat([cs[r][:hi - lo] for r, (lo, hi) in enumerate(per)], dim=0)
    return out_w, out_c
