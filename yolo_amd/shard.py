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
