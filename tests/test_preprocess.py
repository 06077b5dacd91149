"""Preprocessing.  Parity with cv2.resize itself is UNPINNED (cv2 absent, no resized fixture in the
reference): the oracle restates OpenCV's 8-bit INTER_LINEAR algorithm and is pinned here by hand-computed cases;
the GPU kernel is then compared with the oracle bit for bit."""
import numpy as np
import pytest


def _grey(vals):
    a = np.asarray(vals, np.uint8)
    return np.repeat(a[..., None], 3, axis=-1)


def test_oracle_hand_computed_cases():
    from oracle import preprocess_oracle as P
    # 2 -> 3 columns: f = -1/6 (clamped: left pixel), 0.5 (mean), 7/6 (clamped: right pixel)
    assert P.resize_linear_u8(_grey([[10, 200]]), (3, 1))[0, :, 0].tolist() == [10, 105, 200]
    # 2 -> 4 columns: weights (1536, 512) / 2048 -> ((255 * 512 >> 4) * 2048 >> 16) = 255; (255 + 2) >> 2 = 64
    assert P.resize_linear_u8(_grey([[0, 255]]), (4, 1))[0, :, 0].tolist() == [0, 64, 191, 255]
    # the same vertically: row indices are clamped, row weights are not (both rows = row 0 / row 1 at the ends)
    assert P.resize_linear_u8(_grey([[0], [255]]), (1, 4))[:, 0, 0].tolist() == [0, 64, 191, 255]
    # exact 2x reduction: 2x2 box mean with rounding, (1 + 2 + 3 + 5 + 2) >> 2 = 3
    assert P.resize_linear_u8(_grey([[1, 2], [3, 5]]), (1, 1))[0, 0, 0] == 3
    # same size: identity
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    assert np.array_equal(P.resize_linear_u8(img, (5, 7)), img)


def test_oracle_preprocess_channel_order_and_scaling():
    from oracle import preprocess_oracle as P
    img = np.zeros((2, 2, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 255, 128, 0               # B, G, R
    v3 = P.preprocess(img, (2, 2))
    assert v3.shape == (1, 2, 2, 3) and v3.dtype == np.float32
    assert np.array_equal(v3[0, 0, 0], np.array([0, 128, 255], np.float32) / np.float32(255))
    v2 = P.preprocess(img, (2, 2), v2=True)
    assert v2.dtype == np.float32 and v2[0, 0, 0, 0] == -1.0 and v2[0, 0, 0, 2] == 1.0


def test_oracle_bilinear_is_bounded_and_monotone():
    from oracle import preprocess_oracle as P
    ramp = _grey(np.tile(np.arange(0, 250, 5, dtype=np.uint8), (3, 1)))           # 50 columns, increasing
    out = P.resize_linear_u8(ramp, (131, 9)).astype(int)
    assert out.min() >= 0 and out.max() <= 245
    assert (np.diff(out[:, :, 0], axis=1) >= 0).all()
    const = np.full((11, 13, 3), 77, np.uint8)
    assert (P.resize_linear_u8(const, (29, 31)) == 77).all() and (P.resize_linear_u8(const, (5, 4)) == 77).all()


@pytest.mark.gpu
@pytest.mark.parametrize("B,h,w,H,W", [(1, 375, 500, 416, 416), (2, 100, 80, 416, 416), (3, 832, 832, 416, 416),
                                        (1, 416, 416, 416, 416), (2, 1080, 1920, 608, 608), (1, 1, 1, 32, 32),
                                        (1, 37, 2, 64, 96), (4, 720, 1280, 416, 416)])
def test_preprocess_kernel_bit_exact(torch_cuda, B, h, w, H, W):
    from oracle import preprocess_oracle as P
    from yolo_amd.preprocess import preprocess
    rng = np.random.default_rng(B * 1000 + h + w)
    imgs = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    for v2 in (False, True):
        got = preprocess(imgs, (W, H), v2=v2).cpu().numpy()
        assert got.shape == (B, H, W, 3) and got.dtype == np.float32
        for b in range(B):
            assert np.array_equal(got[b], P.preprocess(imgs[b], (W, H), v2=v2)[0]), (b, v2)


@pytest.mark.gpu
def test_preprocess_row_stride_and_errors(torch_cuda):
    import torch
    from oracle import preprocess_oracle as P
    from yolo_amd import _lib
    from yolo_amd.preprocess import preprocess
    L = _lib.lib()
    rng = np.random.default_rng(5)
    wide = rng.integers(0, 256, (2, 50, 70, 3), dtype=np.uint8)           # use the left 60 columns of 70-column rows
    t = torch.from_numpy(wide).cuda()
    out = torch.empty((2, 32, 64, 3), device="cuda")
    assert L.yolo_preprocess(t.data_ptr(), 2, 50, 60, 70 * 3, out.data_ptr(), 32, 64, 0, None) == 0
    for b in range(2):
        assert np.array_equal(out[b].cpu().numpy(), P.preprocess(wide[b, :, :60], (64, 32))[0])
    assert L.yolo_preprocess(t.data_ptr(), 2, 50, 60, 100, out.data_ptr(), 32, 64, 0, None) == -1     # stride < 3 * w
    assert L.yolo_preprocess(t.data_ptr(), 2, 50, 60, 0, out.data_ptr(), 32, 64, 2, None) == -1       # unknown mode
    assert L.yolo_preprocess(None, 2, 50, 60, 0, out.data_ptr(), 32, 64, 0, None) == -1
    with pytest.raises(ValueError):
        preprocess(np.zeros((4, 4, 3), np.float32), (8, 8))
    with pytest.raises(ValueError):
        preprocess(np.zeros((4, 4), np.uint8), (8, 8))


@pytest.mark.gpu
def test_single_image_driver_rescales_boxes(torch_cuda, weight_stream):
    """detect_image = preprocess + detect + the box rescale of test_single_image.py:42-45."""
    from oracle import preprocess_oracle as P
    from yolo_amd.model import yolov3
    from yolo_amd.test_single_image import detect_frames, detect_image
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    net = yolov3(80, DEFAULT_ANCHORS, dtype="fp32").load_weight_stream(weight_stream)
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (2, 120, 200, 3), dtype=np.uint8)
    res = detect_frames(net, frames, (96, 64), 30, 0.01, 0.5)
    assert len(res) == 2
    total = 0
    for b in range(2):
        boxes_, scores_, labels_ = res[b]
        x = P.preprocess(frames[b], (96, 64))
        ob, osc, ol, _, oc = net.detect(x, 30, 0.01, 0.5)
        m = int(oc[0])
        total += m
        ref = ob[0, :m].cpu().numpy().copy()
        ref[:, [0, 2]] *= 200 / 96.0
        ref[:, [1, 3]] *= 120 / 64.0
        assert np.array_equal(boxes_, ref) and np.array_equal(scores_, osc[0, :m].cpu().numpy())
        assert labels_.dtype == np.int32 and np.array_equal(labels_, ol[0, :m].cpu().numpy())
    single = detect_image(net, frames[1], (96, 64), 30, 0.01, 0.5)
    assert all(np.array_equal(a, b) for a, b in zip(single, res[1]))
    assert total > 0


@pytest.mark.gpu
def test_detect_pipeline_equals_sequential(torch_cuda, weight_stream):
    """DetectPipeline (H2D copy of batch i+1 under the compute of batch i) returns what preprocess + detect return."""
    import torch
    from yolo_amd.model import yolov3
    from yolo_amd.pipeline import DetectPipeline
    from yolo_amd.preprocess import preprocess
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(weight_stream)
    rng = np.random.default_rng(71)
    batches = [rng.integers(0, 256, (3, 90, 120, 3), dtype=np.uint8) for _ in range(5)]
    batches[3] = torch.from_numpy(batches[3]).pin_memory()           # a caller-pinned batch skips the staging copy
    pipe = DetectPipeline(net, (96, 64), 20, 0.01, 0.45)
    got = [[t.clone() for t in out] for out in pipe.run(batches)]
    assert len(got) == 5
    total = 0
    for frames, out in zip(batches, got):
        ref = net.detect(preprocess(frames, (96, 64)), 20, 0.01, 0.45)
        cnt = ref[4].cpu().numpy()
        total += int(cnt.sum())
        assert torch.equal(out[4], ref[4])
        for b in range(3):
            for j in range(4):
                assert torch.equal(out[j][b, :cnt[b]], ref[j][b, :cnt[b]])
    assert total > 0
    assert list(pipe.run([])) == []
    with pytest.raises(ValueError):
        list(pipe.run([np.zeros((2, 8, 8, 3), np.float32)]))
