"""yolo_nms_static (the entry point that allocates nothing: candidate arrays in a device array inside the library) against
yolo_nms with a caller workspace, through the C-ABI: two calls in a row, the second on a smaller input, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng, B, n, C):
    xy = rng.random((B, n, 2)) * 400
    wh = rng.random((B, n, 2)) * 120 + 1
    boxes = np.concatenate([xy, xy + wh], 2).astype(np.float32)
    boxes += rng.normal(0, 3, boxes.shape).astype(np.float32)
    scores = (rng.random((B, n, C)) ** 2).astype(np.float32)
    return boxes, scores


def _run(L, static, boxes, scores, C, mb, st, it, mode):
    import torch
    from yolo_amd import _lib, _tensor
    B, n = boxes.shape[:2]
    bd, sd = _tensor.as_dev_f32(boxes), _tensor.as_dev_f32(scores)
    cap = C * mb
    ob = torch.full((B, cap, 4), -7.0, device="cuda")
    osc = torch.full((B, cap), -7.0, device="cuda")
    ol = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    oi = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    oc = torch.full((B,), -7, dtype=torch.int32, device="cuda")
    if static:
        _lib.check(L.yolo_nms_static(bd.data_ptr(), sd.data_ptr(), B, n, C, mb, st, it, mode, ob.data_ptr(), osc.data_ptr(),
                                     ol.data_ptr(), oi.data_ptr(), oc.data_ptr(), _tensor.stream_ptr()))
    else:
        ws = _tensor.workspace(L.yolo_nms_workspace_bytes(B, n, C, mb))
        _lib.check(L.yolo_nms(bd.data_ptr(), sd.data_ptr(), B, n, C, mb, st, it, mode, ob.data_ptr(), osc.data_ptr(),
                              ol.data_ptr(), oi.data_ptr(), oc.data_ptr(), ws.data_ptr(), ws.numel(), _tensor.stream_ptr()))
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (ob, osc, ol, oi, oc)]


def test_nms_static_equals_nms_two_calls(torch_cuda):
    from yolo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(123)
    # first call: one 416-sized image (10647 x 80, the largest config-2 per-image problem); second: smaller in every dimension
    for (B, n, C, mb, st, it, mode) in ((1, 10647, 80, 50, 0.5, 0.5, 0), (2, 3000, 7, 20, 0.2, 0.45, 0), (1, 500, 3, 10, 0.1, 0.5, 1)):
        boxes, scores = _case(rng, B, n, C)
        got = _run(L, True, boxes, scores, C, mb, st, it, mode)
        ref = _run(L, False, boxes, scores, C, mb, st, it, mode)
        assert np.array_equal(got[4], ref[4]) and ref[4].sum() > 0
        for b in range(B):
            m = int(ref[4][b])
            for g, r in zip(got[:4], ref[:4]):
                assert np.array_equal(g[b, :m], r[b, :m])


def test_nms_static_allocates_nothing_and_replays_in_a_graph(torch_cuda):
    """After one warm call on the device (which resolves the device array), a call changes neither the driver's free-memory
    figure nor torch's allocator statistics, and it can be captured into a HIP graph and replayed: the replay on new inputs
    in the same buffers gives what yolo_nms gives."""
    import torch
    from yolo_amd import _lib, _tensor
    L = _lib.lib()
    rng = np.random.default_rng(77)
    B, n, C, mb = 1, 2000, 5, 10
    boxes, scores = _case(rng, B, n, C)
    _run(L, True, boxes, scores, C, mb, 0.3, 0.5, 0)                  # warm call
    bd, sd = _tensor.as_dev_f32(boxes), _tensor.as_dev_f32(scores)
    cap = C * mb
    outs = [torch.zeros((B, cap, 4), device="cuda"), torch.zeros((B, cap), device="cuda"),
            torch.zeros((B, cap), dtype=torch.int32, device="cuda"), torch.zeros((B, cap), dtype=torch.int32, device="cuda"),
            torch.zeros((B,), dtype=torch.int32, device="cuda")]

    def call(stream_ptr):
        _lib.check(L.yolo_nms_static(bd.data_ptr(), sd.data_ptr(), B, n, C, mb, 0.3, 0.5, 0, *[o.data_ptr() for o in outs], stream_ptr))
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    reserved0, allocs0 = torch.cuda.memory_reserved(), torch.cuda.memory_stats().get("allocation.all.allocated", 0)
    call(_tensor.stream_ptr())
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0 and torch.cuda.memory_reserved() == reserved0
    assert torch.cuda.memory_stats().get("allocation.all.allocated", 0) == allocs0
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g.capture_begin()
        call(torch.cuda.current_stream().cuda_stream)
        g.capture_end()
    torch.cuda.synchronize()
    boxes2, scores2 = _case(rng, B, n, C)                            # new inputs in the captured buffers
    bd.copy_(torch.from_numpy(boxes2))
    sd.copy_(torch.from_numpy(scores2))
    for o in outs:
        o.fill_(-3)
    g.replay()
    torch.cuda.synchronize()
    ref = _run(L, False, boxes2, scores2, C, mb, 0.3, 0.5, 0)
    m = int(ref[4][0])
    assert m > 0 and int(outs[4][0]) == m
    for o, r in zip(outs[:4], ref[:4]):
        assert np.array_equal(o[0, :m].cpu().numpy(), r[0, :m])


def test_nms_static_rejects_oversize(torch_cuda):
    import torch
    from yolo_amd import _lib
    L = _lib.lib()
    z = torch.zeros(8, device="cuda")
    # 2 images x 22743 boxes x 80 classes of keys = 29 MB > the 16 MiB built-in buffer: refused before any pointer is touched
    rc = L.yolo_nms_static(z.data_ptr(), z.data_ptr(), 2, 22743, 80, 50, 0.5, 0.5, 0, z.data_ptr(), z.data_ptr(), z.data_ptr(),
                           z.data_ptr(), z.data_ptr(), None)
    assert rc == -4 and b"built-in buffer" in L.yolo_last_error()
