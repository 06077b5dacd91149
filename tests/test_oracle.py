"""CPU: pin the oracle against the golden vectors captured from the reference (oracle/gen_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden, rel_rms_ok


def _cases(npz):
    names = sorted({k.split("/")[0] for k in npz.files})
    return {n: {k.split("/", 1)[1]: npz[k] for k in npz.files if k.startswith(n + "/")} for n in names}


def test_stream_is_reproducible(oracle, weight_stream):
    g = load_golden("g1_convstack_64.npz")
    assert len(weight_stream) == 62001757                      # body size of the public yolov3.weights
    np.testing.assert_array_equal(weight_stream[::1000003], g["stream_probe"])
    assert abs(float(weight_stream.astype(np.float64).sum()) - float(g["stream_sum"])) < 1e-6


@pytest.mark.parametrize("size", [64, 96])
def test_forward_matches_reference_darknet(oracle, oracle_params, size):
    """G1: oracle.forward == reference yolo-v3/002 Darknet (weights loaded by its own load_weights)."""
    g = load_golden("g1_convstack_%d.npz" % size)
    fms = oracle.forward(oracle_params, g["img"])
    for got, key in zip(fms, ("fm1", "fm2", "fm3")):
        ok, maxerr, rms = rel_rms_ok(got, g[key], 1e-4)
        assert ok, (key, maxerr, rms)


@pytest.mark.parametrize("size", [64, 96])
def test_decode_matches_reference_predict_transform(oracle, size):
    """G2: oracle.predict == reference predict_transform (xywh, no clip; fixture logits never clip)."""
    g = load_golden("g1_convstack_%d.npz" % size)
    boxes, confs, probs = oracle.predict([g["fm1"], g["fm2"], g["fm3"]], oracle.YOLO_ANCHORS, (size, size))
    det = g["det"][0]
    cx, cy, w, h = det[:, 0], det[:, 1], det[:, 2], det[:, 3]
    ref = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    np.testing.assert_allclose(boxes[0], ref, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(confs[0, :, 0], det[:, 4], rtol=0, atol=1e-6)
    np.testing.assert_allclose(probs[0], det[:, 5:], rtol=0, atol=1e-6)


def test_decode_clip_hand_case(oracle):
    """model.py:103 clip(exp(t), 1e-9, 50): hand-computed (the PyTorch lesson has no clip)."""
    C = 2
    fm = np.zeros((1, 1, 1, 3 * (5 + C)), np.float32)
    fm[0, 0, 0, 2] = 10.0        # exp -> 22026 -> 50
    fm[0, 0, 0, 3] = -30.0       # exp -> 9.4e-14 -> 1e-9
    anchors = np.asarray([[8, 16], [1, 1], [1, 1]], np.float32)
    _, boxes, _, _ = oracle.reorg_layer(fm, anchors, (32, 32), C)
    np.testing.assert_allclose(boxes[0, 0, 0, 0], [16.0, 16.0, 50 * 8.0, 1e-9 * 16.0], rtol=1e-6)


def test_upsample_align_corners_equals_floor_half(oracle):
    for n in (13, 26, 3, 19):
        x = np.arange(n * n, dtype=np.float32).reshape(1, n, n, 1)
        up = oracle.upsample_layer(x, (1, 2 * n, 2 * n, 1))
        ref = x[:, np.arange(2 * n) // 2][:, :, np.arange(2 * n) // 2]
        np.testing.assert_array_equal(up, ref)


def test_cpu_nms_matches_reference(oracle):
    """G3: oracle.cpu_nms == reference nms_utils.cpu_nms outputs (bit-exact)."""
    cases = _cases(load_golden("g3_cpu_nms.npz"))
    assert len(cases) >= 9
    for name, d in cases.items():
        if name.startswith("py_nms"):
            keep = oracle.py_nms(d["boxes"], d["scores"], int(d["max_boxes"]), float(d["iou_thresh"]))
            np.testing.assert_array_equal(np.asarray(keep), d["keep"], err_msg=name)
            continue
        b, s, l = oracle.cpu_nms(d["boxes"], d["scores"], int(d["C"]), int(d["max_boxes"]),
                                 float(d["score_thresh"]), float(d["iou_thresh"]))
        if bool(d["none"]):
            assert b is None and s is None and l is None, name
            continue
        np.testing.assert_array_equal(b, d["out_boxes"], err_msg=name)
        np.testing.assert_array_equal(s, d["out_scores"], err_msg=name)
        np.testing.assert_array_equal(l, d["out_labels"], err_msg=name)
        assert l.dtype == np.int32


def test_c_oracle_equals_numpy_oracle(oracle):
    """The C twin (used at size) agrees with the numpy restatements, both modes, incl. ties/zero-area."""
    rng = np.random.default_rng(7)
    n, C = 700, 4
    xy = rng.random((n, 2)) * 300
    wh = rng.random((n, 2)) * 80
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    boxes[::11, 2:] = boxes[::11, :2]                    # zero area
    boxes[5::13] = boxes[5::13][:, [2, 3, 0, 1]]         # flipped corners
    scores = np.round(rng.random((n, C)) * 20).astype(np.float32) / 20   # many exact ties
    a = oracle.gpu_nms(boxes[None], scores[None], C, 30, 0.3, 0.4, use_c=True, return_indices=True)
    b = oracle.gpu_nms(boxes[None], scores[None], C, 30, 0.3, 0.4, use_c=False, return_indices=True)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    for c in range(C):
        m = np.nonzero(scores[:, c] >= np.float32(0.3))[0]
        keep = oracle.py_nms(boxes[m], scores[m, c], 30, 0.4)
        idx = oracle.class_nms_c(boxes, scores, c, 30, 0.3, 0.4, 1)
        np.testing.assert_array_equal(m[keep], idx)


def test_calc_iou_fixture_consistency(oracle):
    """G3b: reference eval_utils.calc_iou == plain IoU; anchors the (x0,y0,x1,y1) box convention."""
    g = load_golden("g3b_calc_iou.npz")
    p, t = g["pred"], g["true"]
    lo = np.maximum(p[:, None, :2], t[None, :, :2])
    hi = np.minimum(p[:, None, 2:], t[None, :, 2:])
    wh = np.maximum(hi - lo, 0)
    inter = wh[..., 0] * wh[..., 1]
    ap = (p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 1])
    at = (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    np.testing.assert_allclose(inter / (ap[:, None] + at[None] - inter + 1e-10), g["iou"], rtol=1e-5, atol=1e-6)


# ---- YOLOv2 (config 4) -------------------------------------------------------------------------
def test_v2_calc_output_matches_reference():
    """G4: oracle decode + class-agnostic NMS == reference yolo_v2/Detector.py calc_output (bit-exact)."""
    from oracle import yolo2_oracle as O2
    cases = _cases(load_golden("g4_v2_decode.npz"))
    assert set(cases) == {"a", "b", "c", "empty"}
    for name, d in cases.items():
        boxes, probs = O2.decode(d["logits"])
        assert boxes.dtype == np.float64 and probs.dtype == np.float32
        cls, bx, pr, flat = O2.nms_from_decoded(boxes, probs)
        np.testing.assert_array_equal(cls, d["cls"], err_msg=name)
        np.testing.assert_array_equal(bx, d["boxes"].reshape(-1, 4), err_msg=name)
        np.testing.assert_array_equal(pr, d["probs"], err_msg=name)
        res = O2.calc_output(d["logits"])
        assert [r[0] for r in res] == [O2.CLASSES[c] for c in d["cls"]]
    assert len(cases["empty"]["cls"]) == 0 and len(cases["b"]["cls"]) > 100


def test_v2_network_restatement_shapes_and_reorg():
    """yolo_v2.py:97-113 channel order of reorg and the [reorg, net24] concat; 50.7 M variables."""
    from oracle import yolo2_oracle as O2
    specs = O2.conv_specs()
    assert sum(s["k"] ** 2 * s["cin"] * s["cout"] + (5 if s["bn"] else 1) * s["cout"] for s in specs) == 50686397
    x = np.arange(2 * 4 * 4 * 3, dtype=np.float32).reshape(2, 4, 4, 3)
    r = O2.reorg(x)
    assert r.shape == (2, 2, 2, 12)
    np.testing.assert_array_equal(r[1, 1, 0], np.concatenate([x[1, 2, 0], x[1, 2, 1], x[1, 3, 0], x[1, 3, 1]]))
    p = O2.pooling_layer(x)
    np.testing.assert_array_equal(p[0, 1, 1], x[0, 2:, 2:].max(axis=(0, 1)))
