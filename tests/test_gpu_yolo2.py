"""GPU parity tests of the YOLOv2 path (config 4): HIP through the C-ABI (yolo_amd.yolo_v2) against the CPU oracle
(oracle/yolo2_oracle.py) and the golden vectors of the reference's own numpy post-processing (g4_v2_decode.npz).
Tolerances: fp32 conv stack |a-b| <= 1e-3*|b| + 1e-3*rms(b); bf16 stack |a-b| <= 1e-1*(|b|+rms(b)) and
rms(a-b) <= 3e-2*rms(b); decode: float32 sigmoid/exp/softmax within 2e-6 relative (device expf vs numpy exp);
max-pool, reorg and the NMS (given identical decoded inputs): bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, rel_rms_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O2():
    from oracle import yolo2_oracle
    return yolo2_oracle


@pytest.fixture(scope="module")
def v2_stream(O2):
    return O2.synth_stream(O2.conv_specs(), seed=4321)


@pytest.fixture(scope="module")
def v2_params(O2, v2_stream):
    return O2.split_stream(O2.conv_specs(), v2_stream)


@pytest.fixture(scope="module")
def v2_f32(v2_stream):
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    from yolo_amd.yolo_v2.yolo.yolo_v2 import yolo_v2
    return yolo_v2(False, dtype="fp32").load_variables(v2_stream)


@pytest.fixture(scope="module")
def v2_bf16(v2_stream):
    from yolo_amd.yolo_v2.yolo.yolo_v2 import yolo_v2
    return yolo_v2(False, dtype="bf16").load_variables(v2_stream)


@pytest.fixture(scope="module")
def detector(v2_f32):
    from yolo_amd.yolo_v2.Detector import Detector
    return Detector(v2_f32)


def _cases(npz):
    names = sorted({k.split("/")[0] for k in npz.files})
    return {n: {k.split("/", 1)[1]: npz[k] for k in npz.files if k.startswith(n + "/")} for n in names}


def test_training_graph_is_rejected():
    from yolo_amd.yolo_v2.yolo.yolo_v2 import yolo_v2
    with pytest.raises(ValueError):
        yolo_v2()                          # reference default isTraining=True: not built


def test_pooling_and_reorg_bit_exact(v2_f32, O2):
    rng = np.random.default_rng(3)
    for shape in ((2, 8, 12, 32), (1, 26, 26, 64), (3, 2, 2, 4)):
        x = rng.normal(0, 1, shape).astype(np.float32)
        np.testing.assert_array_equal(v2_f32.pooling_layer(x).cpu().numpy(), O2.pooling_layer(x))
        np.testing.assert_array_equal(v2_f32.reorg(x).cpu().numpy(), O2.reorg(x))
    with pytest.raises(ValueError):
        v2_f32.pooling_layer(np.zeros((1, 3, 4, 4), np.float32))


@pytest.mark.parametrize("name,hw", [("0_conv", 32), ("5_conv", 16), ("26_conv", 26), ("29_conv", 13), ("30_conv", 13)])
def test_conv_layer_vs_oracle(v2_f32, v2_params, O2, name, hw):
    specs = O2.conv_specs()
    i = [s["name"] for s in specs].index(name)
    s = specs[i]
    rng = np.random.default_rng(100 + i)
    x = rng.normal(0, 1, (2, hw, hw, s["cin"])).astype(np.float32)
    ref = O2.conv_layer(x, v2_params[i], s)
    got = v2_f32.conv_layer(x, [s["k"], s["k"], s["cin"], s["cout"]], batch_norm=s["bn"], name=name).cpu().numpy()
    ok, maxerr, rms = rel_rms_ok(got, ref, 1e-3)
    assert ok, (name, maxerr, rms)
    with pytest.raises(ValueError):
        v2_f32.conv_layer(x, [s["k"], s["k"], s["cin"], s["cout"] + 8], batch_norm=s["bn"], name=name)


@pytest.mark.parametrize("B,size", [(2, 64), (1, 416), (3, 96)])
def test_build_networks_fp32_vs_oracle(v2_f32, v2_params, O2, B, size):
    rng = np.random.default_rng(size)
    x = (rng.random((B, size, size, 3), dtype=np.float32) * 2 - 1).astype(np.float32)     # Detector.py:39 range
    ref = O2.build_networks(v2_params, x)
    got = v2_f32.build_networks(x).cpu().numpy()
    assert got.shape == ref.shape == (B, size // 32, size // 32, 125)
    ok, maxerr, rms = rel_rms_ok(got, ref, 1e-3)
    assert ok, (maxerr, rms)
    assert rms > 1e-2


def test_build_networks_bf16_vs_oracle(v2_bf16, v2_params, O2):
    rng = np.random.default_rng(77)
    x = (rng.random((2, 128, 128, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    ref = O2.build_networks(v2_params, x)
    got = v2_bf16.build_networks(x).cpu().numpy()
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    err = np.abs(got - ref)
    assert float((err - 1e-1 * (np.abs(ref) + rms)).max()) <= 0, (float(err.max()), rms)
    assert float(np.sqrt(np.mean(err.astype(np.float64) ** 2))) <= 3e-2 * rms


def test_stem_pool_fusion_equals_separate_kernels(v2_bf16):
    """bf16 plan: the stem conv and the max-pool behind it run as one kernel (front.hip stem_pool_kernel), and so do the second conv
    (32 -> 64) and its pool (resblock.hip conv3x3_c32_pool_kernel); YOLO_NO_STEM_POOL_FUSION=1 keeps all four apart.  The pool of the bf16-rounded activations equals the rounded pool of the f32 ones (rounding is
    monotonic), so the logits agree bit for bit; also on a rectangular input and with lanes."""
    import os
    import torch
    rng = np.random.default_rng(41)
    for B, H, W in ((3, 96, 96), (2, 64, 160), (40, 64, 64)):
        x = (rng.random((B, H, W, 3), dtype=np.float32) * 2 - 1)
        fused = v2_bf16.build_networks(x).clone()
        os.environ["YOLO_NO_STEM_POOL_FUSION"] = "1"
        try:
            apart = v2_bf16.build_networks(x).clone()
        finally:
            del os.environ["YOLO_NO_STEM_POOL_FUSION"]
        assert torch.equal(fused, apart), float((fused - apart).abs().max())


def test_small_batch_plan(v2_bf16, v2_stream):
    """yolo_v2(False, dtype="bf16", small_batch_plan=True): the split-K plan of yolo_net_set_small_batch_plan applies to this net
    too; at batch 1 and 2 it stays within the whole-stack bf16 tolerance of the default plan, is deterministic, and does apply."""
    from conftest import rel_rms_ok
    from yolo_amd.yolo_v2.yolo.yolo_v2 import yolo_v2
    split = yolo_v2(False, dtype="bf16", small_batch_plan=True).load_variables(v2_stream)
    for B, H, W in ((1, 416, 416), (2, 160, 96)):
        x = np.random.default_rng(B).random((B, H, W, 3), dtype=np.float32) * 2 - 1
        b = v2_bf16.build_networks(x).cpu().numpy()
        a = split.build_networks(x).cpu().numpy()
        a2 = split.build_networks(x).cpu().numpy()
        assert np.array_equal(a, a2)
        ok, maxerr, rms = rel_rms_ok(a, b, 1e-1)
        rmserr = float(np.sqrt(np.mean((a - b) ** 2)))
        assert ok, (maxerr, rms)
        assert rmserr <= 3e-2 * rms, (rmserr, rms)
        assert rmserr > 0, "the plan did not apply: outputs are bit-identical to the default plan"


def test_layerwise_composition_equals_fused_plan(v2_f32, O2):
    """build_networks written with the layer-level drop-ins (the reference's own structure, yolo_v2.py:34-68)
    gives the fused plan's logits."""
    import torch
    rng = np.random.default_rng(5)
    x = (rng.random((1, 64, 64, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    m = v2_f32
    from yolo_amd.yolo_v2.yolo.yolo_v2 import conv_shapes
    shapes = {n: (s, bn) for n, s, bn in conv_shapes()}

    def conv(t, name):
        s, bn = shapes[name]
        return m.conv_layer(t, s, batch_norm=bn, name=name)

    net = m.pooling_layer(conv(x, "0_conv"))
    net = m.pooling_layer(conv(net, "2_conv"))
    for n in ("4_conv", "5_conv", "6_conv"):
        net = conv(net, n)
    net = m.pooling_layer(net)
    for n in ("8_conv", "9_conv", "10_conv"):
        net = conv(net, n)
    net = m.pooling_layer(net)
    for n in ("12_conv", "13_conv", "14_conv", "15_conv"):
        net = conv(net, n)
    net16 = conv(net, "16_conv")
    net = m.pooling_layer(net16)
    for n in ("18_conv", "19_conv", "20_conv", "21_conv", "22_conv", "23_conv"):
        net = conv(net, n)
    net24 = conv(net, "24_conv")
    net = m.reorg(conv(net16, "26_conv"))
    net = torch.cat([net, net24], 3)
    net = conv(conv(net, "29_conv"), "30_conv")
    fused = m.build_networks(x)
    ok, maxerr, rms = rel_rms_ok(net.cpu().numpy(), fused.cpu().numpy(), 1e-5)
    assert ok, (maxerr, rms)


def test_decode_vs_reference_golden(detector, O2):
    for name, d in _cases(load_golden("g4_v2_decode.npz")).items():
        rb, rp = O2.decode(d["logits"])
        boxes, probs = detector.decode(d["logits"])
        assert boxes.dtype.is_floating_point and boxes.element_size() == 8
        np.testing.assert_allclose(boxes[0].cpu().numpy(), rb, rtol=2e-6, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(probs[0].cpu().numpy(), rp, rtol=2e-6, atol=1e-12, err_msg=name)


def test_nms_bit_exact_vs_reference_golden(detector, O2):
    """Fed the reference-decoded boxes/probs, the kernel returns exactly calc_output's detections."""
    for name, d in _cases(load_golden("g4_v2_decode.npz")).items():
        rb, rp = O2.decode(d["logits"])
        counts, cls, ob, op, flat = detector.nms(rb[None], rp[None])
        m = int(counts[0])
        assert m == len(d["cls"]), name
        np.testing.assert_array_equal(cls[0, :m].cpu().numpy(), d["cls"], err_msg=name)
        np.testing.assert_array_equal(ob[0, :m].cpu().numpy(), d["boxes"].reshape(-1, 4), err_msg=name)
        np.testing.assert_array_equal(op[0, :m].cpu().numpy(), d["probs"], err_msg=name)
        _, _, _, rflat = O2.nms_from_decoded(rb, rp)
        np.testing.assert_array_equal(flat[0, :m].cpu().numpy(), rflat, err_msg=name)


def test_nms_ties_and_batch(detector, O2):
    """Exact prob ties (descending candidate order, the oracle's documented rule), identical boxes, a batch."""
    rng = np.random.default_rng(9)
    B, S, NB, C = 3, 13, 5, 20
    boxes = np.zeros((B, S, S, NB, 4), np.float64)
    boxes[..., :2] = rng.random((B, S, S, NB, 2)) * 416
    boxes[..., 2:] = rng.random((B, S, S, NB, 2)) * 150 + 5
    boxes[1, 3] = boxes[1, 2]                                            # duplicates
    probs = (np.round(rng.random((B, S, S, NB, C)) ** 6 * 16) / 16).astype(np.float32)   # many ties, few >= 0.3
    counts, cls, ob, op, flat = detector.nms(boxes, probs)
    for b in range(B):
        rc, rb, rp, rf = O2.nms_from_decoded(boxes[b], probs[b])
        m = int(counts[b])
        assert m == len(rc) and m > 10
        np.testing.assert_array_equal(flat[b, :m].cpu().numpy(), rf)
        np.testing.assert_array_equal(cls[b, :m].cpu().numpy(), rc)
        np.testing.assert_array_equal(ob[b, :m].cpu().numpy(), rb)
        np.testing.assert_array_equal(op[b, :m].cpu().numpy(), rp)


def test_calc_output_end_to_end_vs_reference_golden(detector):
    """logits -> HIP decode -> HIP NMS == the reference's calc_output list (class names exact, numbers to 2e-6)."""
    for name, d in _cases(load_golden("g4_v2_decode.npz")).items():
        res = detector.calc_output(d["logits"])
        assert len(res) == len(d["cls"]), name
        if not res:
            continue
        from yolo_amd.yolo_v2.yolo import config as cfg
        assert [r[0] for r in res] == [cfg.CLASSES[c] for c in d["cls"]], name
        np.testing.assert_allclose(np.asarray([r[1:5] for r in res]), d["boxes"].reshape(-1, 4), rtol=2e-6, err_msg=name)
        np.testing.assert_allclose(np.asarray([r[5] for r in res], np.float32), d["probs"], rtol=2e-6, err_msg=name)


def test_detect_batch64_properties(v2_bf16, O2):
    """Config 4 at full size (batch 64, 416, bf16): per-image independence and NMS invariants."""
    import torch
    from yolo_amd.yolo_v2.Detector import Detector
    rng = np.random.default_rng(3)
    x = torch.from_numpy((rng.random((64, 416, 416, 3), dtype=np.float32) * 2 - 1).astype(np.float32)).cuda()
    logits = v2_bf16.build_networks(x)
    assert logits.shape == (64, 13, 13, 125) and bool(torch.isfinite(logits).all())
    one = v2_bf16.build_networks(x[5:6])
    assert torch.equal(one[0], logits[5])                                # images are independent
    det = Detector(v2_bf16)
    boxes, probs = det.decode(logits)
    counts, cls, ob, op, flat = det.nms(boxes, probs)
    boxes, probs = boxes.cpu().numpy(), probs.cpu().numpy()
    for b in (0, 17, 63):
        rc, rb, rp, rf = O2.nms_from_decoded(boxes[b], probs[b])
        m = int(counts[b])
        assert m == len(rc)
        np.testing.assert_array_equal(flat[b, :m].cpu().numpy(), rf)
        p = op[b, :m].cpu().numpy()
        assert np.all(p[:-1] >= p[1:]) and np.all(p >= np.float32(0.3))


def test_detect_resizes_uint8_images_on_gpu(detector, v2_f32):
    """Detector.detect on a uint8 BGR image of another size = the reference's resize / BGR2RGB / normalise lines
    (Detector.py:37-39, restated by the preprocessing oracle) followed by the network, calc_output and the rescale."""
    from oracle import preprocess_oracle as P
    rng = np.random.default_rng(41)
    img = rng.integers(0, 256, (300, 500, 3), dtype=np.uint8)
    got = detector.detect(img)
    x = P.preprocess(img, (416, 416), v2=True)
    ref = detector.calc_output(v2_f32.build_networks(x))
    assert isinstance(got, list) and len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g[0] == r[0] and g[5] == r[5]
        assert g[1] == r[1] * (1.0 * 500 / 416) and g[2] == r[2] * (1.0 * 300 / 416)
        assert g[3] == r[3] * (1.0 * 500 / 416) and g[4] == r[4] * (1.0 * 300 / 416)
    with pytest.raises(ValueError):
        detector.detect(np.zeros((300, 500, 3), np.float32))


def test_nms_degenerate_boxes_survive_in_several_classes(detector, O2):
    """A zero-size box has IoU 0/0 = NaN with itself, and `NaN > 0.5` is false (Detector.py:76, :107-115): every class of
    it that passes the threshold survives, as in the reference, so one box can give several rows.  With the default of
    n rows that can overflow: the kernel sets out_truncated (counts stay <= the rows written) and Detector.nms repeats with n * C rows."""
    import warnings
    rng = np.random.default_rng(19)
    S, NB, C = 13, 5, 20
    boxes = np.zeros((1, S, S, NB, 4), np.float64)
    boxes[..., :2] = rng.random((1, S, S, NB, 2)) * 416
    boxes[..., 2:] = 0.0                                                 # every box degenerate
    probs = np.zeros((1, S, S, NB, C), np.float32)
    probs[0, :, :, :, 3] = 0.6
    probs[0, :, :, :, 11] = 0.5                                          # two classes per box pass -> 2 n survivors
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                                   # the oracle divides 0 by 0 like the reference
        rc, rb, rp, rf = O2.nms_from_decoded(boxes[0], probs[0])
    n = S * S * NB
    assert len(rc) == 2 * n
    counts, cls, ob, op, flat = detector.nms(boxes, probs, n)            # explicit n rows: cut there, and reported as cut
    assert int(counts[0]) == n and bool(detector.last_truncated[0])      # the count never exceeds the rows written
    np.testing.assert_array_equal(flat[0, :n].cpu().numpy(), rf[:n])
    counts, cls, ob, op, flat = detector.nms(boxes, probs, n * C)
    assert not bool(detector.last_truncated.any())
    m = int(counts[0])
    assert m == 2 * n
    c2, _, _, _, f2 = detector.nms(boxes, probs)                         # default: repeats itself with n * C rows
    assert int(c2[0]) == 2 * n and not bool(detector.last_truncated.any())
    np.testing.assert_array_equal(f2[0, :m].cpu().numpy(), rf)
    np.testing.assert_array_equal(flat[0, :m].cpu().numpy(), rf)
    np.testing.assert_array_equal(cls[0, :m].cpu().numpy(), rc)
    np.testing.assert_array_equal(op[0, :m].cpu().numpy(), rp)
