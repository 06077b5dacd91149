"""GPU parity tests: the HIP path (through the C-ABI via yolo_amd) against the CPU oracle and the
golden fixtures.  Run on the MI355X box:  python -u -m pytest tests -m gpu -x -v --timeout 120
Tolerances: fp32 conv stack / decode  |a-b| <= 1e-3*|b| + 1e-3*rms(b)   (north_star: 1e-3 relative);
bf16: single layer |a-b| <= 4e-2*|b| + 4e-2*rms(b); whole 75-layer stack |a-b| <= 1e-1*(|b| + rms(b)) and
rms(a-b) <= 3e-2*rms(b);   NMS: bit-exact indices, boxes, scores, labels."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden, rel_rms_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net_f32(torch_cuda, oracle, weight_stream):
    from yolo_amd.model import yolov3
    return yolov3(80, oracle.YOLO_ANCHORS, dtype="fp32").load_weight_stream(weight_stream)


@pytest.fixture(scope="module")
def net_bf16(torch_cuda, oracle, weight_stream):
    from yolo_amd.model import yolov3
    return yolov3(80, oracle.YOLO_ANCHORS, dtype="bf16").load_weight_stream(weight_stream)


# ---------------------------------------------------------------------------------------------
# single conv layers
# ---------------------------------------------------------------------------------------------
def _conv_case(torch, oracle, dtype, cin, cout, k, stride, bn, B, H, W, residual, seed):
    from yolo_amd import _lib, _tensor
    rng = np.random.default_rng(seed)
    w = rng.normal(0, np.sqrt(2.0 / (k * k * cin)), (k, k, cin, cout)).astype(np.float32)
    p = dict(w=w)
    if bn:
        p.update(beta=rng.normal(0, 0.1, cout).astype(np.float32), gamma=rng.uniform(0.5, 1.5, cout).astype(np.float32),
                 mean=rng.normal(0, 0.1, cout).astype(np.float32), var=rng.uniform(0.5, 1.5, cout).astype(np.float32))
        ps = [p["beta"], p["gamma"], p["mean"], p["var"]]
    else:
        p.update(bias=rng.normal(0, 0.1, cout).astype(np.float32))
        ps = [p["bias"], None, None, None]
    x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32) if cin != 3 else rng.random((B, H, W, 3), dtype=np.float32)
    spec = dict(cin=cin, cout=cout, k=k, stride=stride, bn=bn)
    ref = oracle.conv2d(x, p, spec)
    res = None
    if residual:
        res = rng.normal(0, 1, ref.shape).astype(np.float32)
        ref = ref + res
    L = _lib.lib()
    h = ctypes.c_void_p()
    ptr = lambda a: a.ctypes.data if a is not None else None
    _lib.check(L.yolo_conv_create(_lib.DTYPES[dtype], cin, cout, k, stride, int(bn), ptr(w), ptr(ps[0]), ptr(ps[1]),
                                  ptr(ps[2]), ptr(ps[3]), ctypes.byref(h)))
    try:
        xd = _tensor.as_dev_f32(x)
        rd = _tensor.as_dev_f32(res) if residual else None
        out = _tensor.empty(ref.shape)
        ws = _tensor.workspace(L.yolo_conv_workspace_bytes(h, B, H, W))
        _lib.check(L.yolo_conv_forward(h, xd.data_ptr(), B, H, W, _tensor.ptr(rd), out.data_ptr(), ws.data_ptr(),
                                       ws.numel(), _tensor.stream_ptr()))
        got = out.cpu().numpy()
    finally:
        L.yolo_conv_destroy(h)
    return got, ref


CONV_CASES = [
    # cin, cout, k, stride, bn, B, H, W, residual
    (64, 32, 1, 1, True, 2, 16, 16, False),       # S2-like 1x1, BM=256 x BN=32 tile
    (32, 64, 3, 1, True, 2, 16, 24, True),        # Cin=32: two taps per 128-byte K-step in bf16, residual
    (32, 64, 3, 2, True, 1, 32, 32, False),       # stride-2 with explicit symmetric pad
    (128, 256, 3, 1, True, 2, 13, 13, True),      # 128x128 tile, odd grid, M tail
    (256, 512, 3, 2, True, 1, 26, 26, False),
    (384, 128, 1, 1, True, 1, 20, 12, False),     # concat-sized Cin (not a power of two)
    (512, 255, 1, 1, False, 2, 13, 13, False),    # detection head: bias, no activation, N=255
    (3, 32, 3, 1, True, 2, 32, 32, False),        # stem
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "c%d-%d_k%d_s%d" % c[:4])
def test_conv_layer_fp32(torch_cuda, oracle, case):
    got, ref = _conv_case(torch_cuda, oracle, "fp32", *case, seed=100)
    ok, maxerr, rms = rel_rms_ok(got, ref, 1e-3)
    assert ok, (maxerr, rms)
    assert maxerr <= 2e-5 * max(rms, 1.0) * 10, (maxerr, rms)      # exact-f32 MFMA: really ~1e-6


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "c%d-%d_k%d_s%d" % c[:4])
def test_conv_layer_bf16(torch_cuda, oracle, case):
    got, ref = _conv_case(torch_cuda, oracle, "bf16", *case, seed=101)
    ok, maxerr, rms = rel_rms_ok(got, ref, 4e-2)
    assert ok, (maxerr, rms)


def test_conv_rejects_bad_arguments(torch_cuda):
    from yolo_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()
    w = np.zeros((3, 3, 5, 8), np.float32)
    b = np.zeros(8, np.float32)
    rc = L.yolo_conv_create(0, 5, 8, 3, 1, 0, w.ctypes.data, b.ctypes.data, None, None, None, ctypes.byref(h))
    assert rc == -2 and b"Cin=5" in L.yolo_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


# ---------------------------------------------------------------------------------------------
# whole conv stack
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [64, 96])
def test_forward_fp32_vs_reference_golden(net_f32, size):
    """G1 fixture = the reference's own PyTorch Darknet-53 on the same synthetic weights."""
    g = load_golden("g1_convstack_%d.npz" % size)
    fms = net_f32.forward(g["img"])
    for got, key in zip(fms, ("fm1", "fm2", "fm3")):
        assert tuple(got.shape) == g[key].shape
        ok, maxerr, rms = rel_rms_ok(got.cpu().numpy(), g[key], 1e-3)
        assert ok, (key, maxerr, rms)


def test_forward_fp32_416_batch2_vs_oracle(net_f32, oracle, oracle_params):
    img = np.random.default_rng(1).random((2, 416, 416, 3), dtype=np.float32)
    ref = oracle.forward(oracle_params, img)
    got = net_f32.forward(img)
    assert net_f32.img_size == (416, 416)
    for a, b in zip(got, ref):
        ok, maxerr, rms = rel_rms_ok(a.cpu().numpy(), b, 1e-3)
        assert ok, (b.shape, maxerr, rms)


def test_forward_fp32_rectangular_batch3(net_f32, oracle, oracle_params):
    img = np.random.default_rng(3).random((3, 96, 160, 3), dtype=np.float32)
    ref = oracle.forward(oracle_params, img)
    got = net_f32.forward(img)
    for a, b in zip(got, ref):
        ok, maxerr, rms = rel_rms_ok(a.cpu().numpy(), b, 1e-3)
        assert ok, (b.shape, maxerr, rms)


def test_forward_bf16_vs_oracle(net_bf16, oracle, oracle_params):
    """bf16 storage + bf16 MFMA inputs, f32 accumulate, 75 layers deep: stated tolerance
    max |a-b| <= 1e-1*(|b| + rms(b)) and rms(a-b) <= 3e-2*rms(b)."""
    img = np.random.default_rng(1).random((2, 416, 416, 3), dtype=np.float32)
    ref = oracle.forward(oracle_params, img)
    got = net_bf16.forward(img)
    for a, b in zip(got, ref):
        a = a.cpu().numpy()
        ok, maxerr, rms = rel_rms_ok(a, b, 1e-1)
        rmserr = float(np.sqrt(np.mean((a - b) ** 2)))
        print("bf16 fm", b.shape, "max|err| %.4g rms(err) %.4g rms(ref) %.4g" % (maxerr, rmserr, rms))
        assert ok, (b.shape, maxerr, rms)
        assert rmserr <= 3e-2 * rms, (rmserr, rms)


def test_forward_rejects_bad_shapes(net_f32):
    with pytest.raises(ValueError):
        net_f32.forward(np.zeros((1, 100, 96, 3), np.float32))
    with pytest.raises(ValueError):
        net_f32.forward(np.zeros((1, 96, 96, 4), np.float32))
    with pytest.raises(ValueError):
        net_f32.forward(np.zeros((1, 96, 96, 3), np.float32), is_training=True)


# ---------------------------------------------------------------------------------------------
# decode
# ---------------------------------------------------------------------------------------------
def _rand_fms(rng, B, H, W, C, scale=1.5):
    return [rng.normal(0, scale, (B, H // r, W // r, 3 * (5 + C))).astype(np.float32) for r in (32, 16, 8)]


@pytest.mark.parametrize("B,H,W,C", [(2, 416, 416, 80), (1, 608, 608, 80), (3, 96, 160, 20)])
def test_decode_vs_oracle(torch_cuda, oracle, B, H, W, C):
    from yolo_amd.model import yolov3
    rng = np.random.default_rng(5)
    fms = _rand_fms(rng, B, H, W, C)
    fms[0][0, 0, 0, 2] = 12.0       # exp clip high
    fms[0][0, 0, 0, 3] = -40.0      # exp clip low
    m = yolov3(C, oracle.YOLO_ANCHORS)
    m.img_size = (H, W)
    boxes, confs, probs, scores = m.predict(fms, return_scores=True)
    rb, rc, rp = oracle.predict(fms, oracle.YOLO_ANCHORS, (H, W), C)
    n = 3 * sum((H // r) * (W // r) for r in (32, 16, 8))
    assert tuple(boxes.shape) == (B, n, 4) and tuple(confs.shape) == (B, n, 1) and tuple(probs.shape) == (B, n, C)
    np.testing.assert_allclose(boxes.cpu().numpy(), rb, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(confs.cpu().numpy(), rc, rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(probs.cpu().numpy(), rp, rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(scores.cpu().numpy(), rc * rp, rtol=1e-3, atol=1e-6)
    b2, c2, p2 = m.predict(fms)
    assert torch_cuda.equal(b2, boxes) and torch_cuda.equal(c2, confs) and torch_cuda.equal(p2, probs)


def test_decode_vs_reference_golden(torch_cuda, oracle):
    """G2: reference predict_transform output (xywh) for the fixture feature maps."""
    from yolo_amd.model import yolov3
    g = load_golden("g1_convstack_96.npz")
    m = yolov3(80, oracle.YOLO_ANCHORS)
    m.img_size = (96, 96)
    boxes, confs, probs = m.predict([g["fm1"], g["fm2"], g["fm3"]])
    det = g["det"][0]
    cx, cy, w, h = det[:, 0], det[:, 1], det[:, 2], det[:, 3]
    ref = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    np.testing.assert_allclose(boxes[0].cpu().numpy(), ref, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(confs[0, :, 0].cpu().numpy(), det[:, 4], atol=1e-6)
    np.testing.assert_allclose(probs[0].cpu().numpy(), det[:, 5:], atol=1e-6)


def test_reorg_layer_vs_oracle(torch_cuda, oracle):
    from yolo_amd.model import yolov3
    rng = np.random.default_rng(6)
    fm = rng.normal(0, 1.5, (2, 6, 10, 255)).astype(np.float32)
    m = yolov3(80, oracle.YOLO_ANCHORS)
    m.img_size = (96, 160)
    xy, boxes, cl, pl = m.reorg_layer(fm, oracle.YOLO_ANCHORS[3:6])
    rxy, rb, rcl, rpl = oracle.reorg_layer(fm, oracle.YOLO_ANCHORS[3:6], (96, 160))
    np.testing.assert_array_equal(xy.cpu().numpy(), rxy)
    np.testing.assert_allclose(boxes.cpu().numpy(), rb, rtol=1e-3, atol=1e-3)
    np.testing.assert_array_equal(cl.cpu().numpy(), rcl)
    np.testing.assert_array_equal(pl.cpu().numpy(), rpl)


# ---------------------------------------------------------------------------------------------
# NMS: bit-exact
# ---------------------------------------------------------------------------------------------
def _boxes(rng, n, clustered=0, span=400.0, size=120.0, jitter=6.0):
    if clustered:
        xy = rng.random((clustered, 2)) * span
        wh = rng.random((clustered, 2)) * size + 1
        base = np.concatenate([xy, xy + wh], 1)
        return (base[rng.integers(0, clustered, n)] + rng.normal(0, jitter, (n, 4))).astype(np.float32)
    xy = rng.random((n, 2)) * span
    wh = rng.random((n, 2)) * size + 1
    return np.concatenate([xy, xy + wh], 1).astype(np.float32)


def _check_gpu_nms(oracle, boxes, scores, C, max_boxes, st, it):
    from yolo_amd.utils import nms_utils
    rb, rs, rl, ri = oracle.gpu_nms(boxes, scores, C, max_boxes, st, it, return_indices=True)
    ob, osc, ol, oi, oc = nms_utils.gpu_nms_batched(boxes.reshape(1, -1, 4), scores.reshape(1, -1, C), C, max_boxes, st, it)
    m = int(oc[0])
    assert m == len(ri), (m, len(ri))
    np.testing.assert_array_equal(oi[0, :m].cpu().numpy(), ri)          # keep-indices, bit-exact
    np.testing.assert_array_equal(ol[0, :m].cpu().numpy(), rl)
    np.testing.assert_array_equal(ob[0, :m].cpu().numpy(), rb)
    np.testing.assert_array_equal(osc[0, :m].cpu().numpy(), rs)
    gb, gs, gl = nms_utils.gpu_nms(boxes, scores, C, max_boxes, st, it)
    np.testing.assert_array_equal(gb.cpu().numpy(), rb)
    np.testing.assert_array_equal(gs.cpu().numpy(), rs)
    np.testing.assert_array_equal(gl.cpu().numpy(), rl)
    assert gl.cpu().numpy().dtype == np.int32
    return m


@pytest.mark.parametrize("n,C,clustered,mb,st,it", [
    (200, 3, 0, 50, 0.5, 0.5),
    (1500, 5, 10, 20, 0.3, 0.45),
    (10647, 80, 40, 50, 0.5, 0.5),       # config-2 shape
    (10647, 80, 40, 30, 0.4, 0.5),       # test_single_image.py:39 thresholds
    (3000, 2, 8, 7, 0.0, 0.6),           # k = n > 2048: best-alive path, max_boxes cut
    (5000, 3, 3, 200, 0.05, 0.3),        # long suppression chains, candidates exhausted before max_boxes
    (64, 1, 0, 50, 0.0, 0.5), (65, 1, 2, 50, 0.0, 0.5), (1, 1, 0, 50, 0.0, 0.5),
])
def test_gpu_nms_bit_exact(torch_cuda, oracle, n, C, clustered, mb, st, it):
    rng = np.random.default_rng(n + C)
    boxes = _boxes(rng, n, clustered)
    scores = rng.random((n, C)).astype(np.float32)
    _check_gpu_nms(oracle, boxes[None], scores[None], C, mb, st, it)


def test_gpu_nms_ties_degenerate_and_empty(torch_cuda, oracle):
    from yolo_amd.utils import nms_utils
    rng = np.random.default_rng(77)
    n, C = 4000, 4
    boxes = _boxes(rng, n, 12)
    boxes[::9, 2:] = boxes[::9, :2]                       # zero-area: never suppress / suppressed
    boxes[4::17] = boxes[4::17][:, [2, 3, 0, 1]]          # flipped corners (TF normalises with min/max)
    scores = (np.round(rng.random((n, C)) * 16) / 16).astype(np.float32)     # massive exact ties
    scores[:, 2] = 0.0                                    # empty class in the middle
    scores[5, 3] = -0.0
    m = _check_gpu_nms(oracle, boxes[None], scores[None], C, 60, 0.25, 0.5)
    assert m > 0
    # nothing passes -> empty arrays with the reference's shapes (gpu) / None triple (cpu)
    gb, gs, gl = nms_utils.gpu_nms(boxes[None], scores[None] * 0.1, C, 50, 0.9, 0.5)
    assert tuple(gb.shape) == (0, 4) and tuple(gs.shape) == (0,) and tuple(gl.shape) == (0,)
    assert nms_utils.cpu_nms(boxes[None], scores[None] * 0.1, C, 50, 0.9, 0.5) == (None, None, None)


def test_gpu_nms_batched_equals_per_image(torch_cuda, oracle):
    from yolo_amd.utils import nms_utils
    rng = np.random.default_rng(8)
    B, n, C = 4, 2500, 6
    boxes = np.stack([_boxes(rng, n, 10) for _ in range(B)])
    scores = rng.random((B, n, C)).astype(np.float32) ** 2
    ob, osc, ol, oi, oc = nms_utils.gpu_nms_batched(boxes, scores, C, 25, 0.3, 0.5)
    for i in range(B):
        rb, rs, rl, ri = oracle.gpu_nms(boxes[i], scores[i], C, 25, 0.3, 0.5, return_indices=True)
        m = int(oc[i])
        assert m == len(ri)
        np.testing.assert_array_equal(oi[i, :m].cpu().numpy(), ri)
        np.testing.assert_array_equal(ob[i, :m].cpu().numpy(), rb)
        np.testing.assert_array_equal(osc[i, :m].cpu().numpy(), rs)
        np.testing.assert_array_equal(ol[i, :m].cpu().numpy(), rl)


def test_gpu_nms_flattens_batch_like_reference(torch_cuda, oracle):
    """nms_utils.py:20-21: with N > 1 the reference merges the images; the drop-in keeps that."""
    from yolo_amd.utils import nms_utils
    rng = np.random.default_rng(9)
    boxes = _boxes(rng, 600, 6).reshape(2, 300, 4)
    scores = rng.random((2, 300, 3)).astype(np.float32)
    rb, rs, rl = oracle.gpu_nms(boxes, scores, 3, 10, 0.4, 0.5)
    gb, gs, gl = nms_utils.gpu_nms(boxes, scores, 3, 10, 0.4, 0.5)
    np.testing.assert_array_equal(gb.cpu().numpy(), rb)
    np.testing.assert_array_equal(gs.cpu().numpy(), rs)
    np.testing.assert_array_equal(gl.cpu().numpy(), rl)


def test_cpu_nms_and_py_nms_vs_reference_golden(torch_cuda):
    """G3: outputs of the reference's own numpy cpu_nms / py_nms (bit-exact)."""
    from yolo_amd.utils import nms_utils
    g = load_golden("g3_cpu_nms.npz")
    names = sorted({k.split("/")[0] for k in g.files})
    for name in names:
        d = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(name + "/")}
        if name.startswith("py_nms"):
            keep = nms_utils.py_nms(d["boxes"], d["scores"], int(d["max_boxes"]), float(d["iou_thresh"]))
            np.testing.assert_array_equal(np.asarray(keep), d["keep"], err_msg=name)
            continue
        b, s, l = nms_utils.cpu_nms(d["boxes"], d["scores"], int(d["C"]), int(d["max_boxes"]),
                                    float(d["score_thresh"]), float(d["iou_thresh"]))
        if bool(d["none"]):
            assert b is None and s is None and l is None, name
            continue
        np.testing.assert_array_equal(b, d["out_boxes"], err_msg=name)
        np.testing.assert_array_equal(s, d["out_scores"], err_msg=name)
        np.testing.assert_array_equal(l, d["out_labels"], err_msg=name)
        assert l.dtype == np.int32 and s.dtype == np.float32


def test_nms_stress_608_properties(torch_cuda, oracle):
    """Config 5 shape (22743 x 80, score_thresh 0.001) on one image: oracle equality on a class subset
    plus size-independent properties (idempotence, score order, pairwise IoU <= thresh)."""
    from yolo_amd.utils import nms_utils
    rng = np.random.default_rng(55)
    fms = _rand_fms(rng, 1, 608, 608, 80, 1.0)
    boxes, confs, probs = oracle.predict(fms, oracle.YOLO_ANCHORS, (608, 608))
    scores = (confs * probs).astype(np.float32)
    ob, osc, ol, oi, oc = nms_utils.gpu_nms_batched(boxes, scores, 80, 50, 0.001, 0.5)
    m = int(oc[0])
    ob, osc, ol, oi = [t[0, :m].cpu().numpy() for t in (ob, osc, ol, oi)]
    assert m == 80 * 50
    for c in (0, 17, 79):
        ref = oracle.class_nms_c(boxes[0], scores[0], c, 50, 0.001, 0.5, 0)
        np.testing.assert_array_equal(oi[ol == c], ref)
    for c in range(80):
        s = osc[ol == c]
        assert np.all(s[:-1] >= s[1:])
    # idempotence: NMS of the kept boxes keeps them all
    for c in (3, 40):
        kb, ks = ob[ol == c], osc[ol == c]
        again = oracle.tf_non_max_suppression(kb, ks, 50, 0.5)
        np.testing.assert_array_equal(again, np.arange(len(kb)))


# ---------------------------------------------------------------------------------------------
# layer-level drop-in functions (yolo_v3/utils/layer_utils.py signatures)
# ---------------------------------------------------------------------------------------------
def test_layer_utils_compose_like_forward(net_f32, torch_cuda, oracle, weight_stream):
    """darknet53_body + yolo_block + conv2d + upsample_layer assembled exactly like model.py:37-65
    reproduce the fused whole-net plan (and therefore the reference golden)."""
    from yolo_amd import weights
    from yolo_amd.utils import layer_utils as lu
    g = load_golden("g1_convstack_64.npz")
    params = weights.split_stream(weight_stream, 80)
    torch = torch_cuda
    with lu.variable_scope(params, "fp32") as scope:
        route_1, route_2, route_3 = lu.darknet53_body(g["img"])
        inter1, net = lu.yolo_block(route_3, 512)
        fm1 = lu.conv2d(net, 255, 1)
        inter1 = lu.conv2d(inter1, 256, 1)
        inter1 = lu.upsample_layer(inter1, list(route_2.shape))
        concat1 = torch.cat([inter1, route_2], dim=3)
        inter2, net = lu.yolo_block(concat1, 256)
        fm2 = lu.conv2d(net, 255, 1)
        inter2 = lu.conv2d(inter2, 128, 1)
        inter2 = lu.upsample_layer(inter2, list(route_1.shape))
        concat2 = torch.cat([inter2, route_1], dim=3)
        _, net = lu.yolo_block(concat2, 128)
        fm3 = lu.conv2d(net, 255, 1)
        assert scope.next == 75
    for got, key in zip((fm1, fm2, fm3), ("fm1", "fm2", "fm3")):
        ok, maxerr, rms = rel_rms_ok(got.cpu().numpy(), g[key], 1e-3)
        assert ok, (key, maxerr, rms)
    x = np.random.default_rng(2).normal(size=(2, 5, 7, 8)).astype(np.float32)
    up = lu.upsample_layer(x, [2, 10, 14, 8])
    np.testing.assert_array_equal(up.cpu().numpy(), oracle.upsample_layer(x, (2, 10, 14, 8)))
    # any output size (layer_utils.py:67-72 takes an arbitrary out_shape): TF 1.x align_corners rounding, bit-exact vs the oracle
    for shape, out_hw in (((2, 5, 7, 8), (15, 21)), ((2, 5, 7, 8), (5, 7)), ((1, 13, 13, 6), (19, 31)), ((3, 4, 9, 5), (1, 1)),
                          ((1, 1, 1, 4), (3, 2)), ((2, 26, 26, 12), (13, 40)), ((1, 7, 3, 3), (100, 99))):
        xx = np.random.default_rng(sum(shape)).normal(size=shape).astype(np.float32)
        tgt = [shape[0], out_hw[0], out_hw[1], shape[3]]
        np.testing.assert_array_equal(lu.upsample_layer(xx, tgt).cpu().numpy(), oracle.upsample_layer(xx, tuple(tgt)))
    with pytest.raises(ValueError):
        lu.upsample_layer(x, [2, 0, 21, 8])


def test_detect_end_to_end_vs_oracle(net_f32, oracle, oracle_params):
    """Counterpart of test_single_image.py:31-40 (30 / 0.4 / 0.5): fp32 HIP chain vs oracle chain.
    Scores differ in the last bits, so compare as sets with tolerance; exact NMS parity is tested above."""
    img = np.random.default_rng(4).random((1, 416, 416, 3), dtype=np.float32)
    ob, osc, ol, oi, oc = net_f32.detect(img)
    m = int(oc[0])
    rb, rs, rl = oracle.detect(oracle_params, oracle.YOLO_ANCHORS, img)
    assert abs(m - len(rl)) <= max(2, len(rl) // 50)
    got = {(int(l), int(i)) for l, i in zip(ol[0, :m].cpu().numpy(), oi[0, :m].cpu().numpy())}
    fms = oracle.forward(oracle_params, img)
    b, c, p = oracle.predict(fms, oracle.YOLO_ANCHORS, (416, 416))
    _, _, rl2, ri2 = oracle.gpu_nms(b, c * p, 80, 30, 0.4, 0.5, return_indices=True)
    ref = {(int(l), int(i)) for l, i in zip(rl2, ri2)}
    assert len(got & ref) >= 0.97 * len(ref), (len(got), len(ref), len(got & ref))


def test_detect_post_fused_equals_separate_calls(net_bf16, net_f32):
    """yolo_detect_post (decode + score + mask + NMS in one call) == predict -> conf*prob -> gpu_nms_batched,
    bit for bit, at batch 8 / 416 and at a second shape, several thresholds (incl. the eval-style low one)."""
    import torch
    rng = np.random.default_rng(17)
    for net, B, size in ((net_bf16, 8, 416), (net_f32, 2, 96)):
        x = rng.random((B, size, size, 3), dtype=np.float32)
        fms = net.forward(x)
        for mb, st, it in ((50, 0.5, 0.5), (30, 0.4, 0.5), (20, 0.01, 0.45)):
            from yolo_amd.utils.nms_utils import gpu_nms_batched
            boxes, _, _, scores = net.predict(fms, return_scores=True)
            ref = gpu_nms_batched(boxes, scores, net.class_num, mb, st, it)
            got = net.postprocess(fms, mb, st, it)
            assert torch.equal(ref[4], got[4])
            assert st > 0.1 or int(ref[4].sum()) > 0
            for b in range(B):
                m = int(ref[4][b])
                for r, g in zip(ref[:4], got[:4]):
                    assert torch.equal(r[b, :m], g[b, :m])


def test_detect_with_more_than_250_classes(torch_cuda, oracle):
    """The fused detect path holds <= 250 classes; the reference has no limit (nms_utils.py:7): a 300-class model goes through
    predict + gpu_nms_batched inside detect() and gives the oracle's detections (fp32, same boxes / labels, scores to 1e-5)."""
    from yolo_amd import weights
    from yolo_amd.model import yolov3
    C = 300
    stream = weights.synth_darknet_stream(C, seed=5)
    m = yolov3(C, oracle.YOLO_ANCHORS, dtype="fp32").load_weight_stream(stream)
    img = np.random.default_rng(8).random((2, 64, 96, 3), dtype=np.float32)
    ob, osc, ol, oi, oc = [t.cpu().numpy() for t in m.detect(img, 7, 0.02, 0.5)]
    params = oracle.split_darknet_stream(oracle.yolov3_conv_specs(C), stream)
    fms = oracle.forward(params, img, C)
    rbx, rcf, rpr = oracle.predict(fms, oracle.YOLO_ANCHORS, (64, 96), C)
    hit = tot = 0
    for b in range(2):
        _, _, rl, ri = oracle.gpu_nms(rbx[b], rcf[b] * rpr[b], C, 7, 0.02, 0.5, return_indices=True)
        n = int(oc[b])
        got = {(int(l), int(i)) for l, i in zip(ol[b, :n], oi[b, :n])}
        ref = {(int(l), int(i)) for l, i in zip(rl, ri)}
        assert int(ol[b, :n].max()) > 250 and abs(n - len(ref)) <= max(2, len(ref) // 50)
        hit += len(got & ref)
        tot += len(ref)
    assert tot > 50 and hit >= 0.97 * tot, (hit, tot)      # scores differ in the last bits: sets with tolerance, as above


def test_detect_post_argument_errors(net_f32):
    with pytest.raises(ValueError):
        net_f32.postprocess([np.zeros((1, 2, 2, 255), np.float32), np.zeros((1, 4, 4, 255), np.float32),
                             np.zeros((1, 8, 8, 255), np.float32)], max_boxes=0)


def test_front_fusion_equals_separate_kernels(net_bf16):
    """bf16: the fused stem + first stride-2 conv kernel (front.hip) and the fused first residual block
    (resblock.hip) give the feature maps of the separate kernels bit for bit (same products, same accumulation
    order, same roundings)."""
    import os
    import torch
    rng = np.random.default_rng(23)
    for B, H, W in ((2, 416, 416), (1, 64, 96), (3, 32, 32)):
        x = rng.random((B, H, W, 3), dtype=np.float32)
        fused = [f.clone() for f in net_bf16.forward(x)]
        os.environ["YOLO_NO_FRONT_FUSION"] = "1"
        os.environ["YOLO_NO_RES_FUSION"] = "1"
        try:
            plain = [f.clone() for f in net_bf16.forward(x)]
        finally:
            del os.environ["YOLO_NO_FRONT_FUSION"]
            del os.environ["YOLO_NO_RES_FUSION"]
        for a, b in zip(fused, plain):
            assert torch.equal(a, b), float((a - b).abs().max())


def test_lanes_equal_single_launch(net_bf16, net_f32):
    """forward splits the batch into lanes (sub-batches on streams of their own, net.hip forward_lanes): the feature
    maps are those of one launch per layer over the whole batch, bit for bit, also for uneven splits."""
    import os
    import torch
    rng = np.random.default_rng(29)
    for net, B, H, W in ((net_bf16, 13, 96, 96), (net_bf16, 5, 64, 32), (net_f32, 3, 32, 64)):
        x = rng.random((B, H, W, 3), dtype=np.float32)
        outs = {}
        try:
            for lanes in (1, 2, 3, 4):
                os.environ["YOLO_LANES"] = str(lanes)
                outs[lanes] = [f.clone() for f in net.forward(x)]
        finally:
            del os.environ["YOLO_LANES"]
        outs[0] = [f.clone() for f in net.forward(x)]              # default lane count
        for lanes in (0, 2, 3, 4):
            for a, b in zip(outs[1], outs[lanes]):
                assert torch.equal(a, b), (lanes, float((a - b).abs().max()))


def test_lanes_workspace_checked(net_bf16):
    """a workspace sized for fewer images than the call is refused, with lanes as without."""
    import ctypes
    import torch
    from yolo_amd import _lib
    L = _lib.lib()
    B, H, W = 16, 64, 64
    need = L.yolo_net_workspace_bytes(net_bf16._net, B, H, W)
    assert need >= L.yolo_net_workspace_bytes(net_bf16._net, B // 2, H, W) * 2 - 4096
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    x = torch.zeros((B, H, W, 3), device="cuda")
    fms = [torch.empty((B, H // r, W // r, 255), device="cuda") for r in (32, 16, 8)]
    rc = L.yolo_net_forward(net_bf16._net, x.data_ptr(), B, H, W, ws.data_ptr(), need // 4, fms[0].data_ptr(),
                            fms[1].data_ptr(), fms[2].data_ptr(), None)
    assert rc == -5                                            # YOLO_ERR_WORKSPACE
    rc = L.yolo_net_forward(net_bf16._net, x.data_ptr(), B, H, W, ws.data_ptr(), need, fms[0].data_ptr(),
                            fms[1].data_ptr(), fms[2].data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()


def test_detect_one_call_equals_forward_then_postprocess(net_bf16, net_f32):
    """yolo_net_detect (forward + post-processing per lane in one call) = yolo_net_forward, then yolo_detect_post."""
    import os
    import torch
    rng = np.random.default_rng(31)
    for net, B, H, W, st in ((net_bf16, 13, 96, 96, 0.01), (net_f32, 3, 64, 96, 0.02), (net_bf16, 2, 416, 416, 0.3)):
        x = rng.random((B, H, W, 3), dtype=np.float32)
        ref = [t.clone() for t in net.postprocess(net.forward(x), 20, st, 0.45)]
        outs = []
        try:
            for lanes in ("1", "3"):
                os.environ["YOLO_LANES"] = lanes
                outs.append([t.clone() for t in net.detect(x, 20, st, 0.45)])
        finally:
            del os.environ["YOLO_LANES"]
        outs.append([t.clone() for t in net.detect(x, 20, st, 0.45)])
        cnt = ref[4].cpu().numpy()
        assert cnt.sum() > 0
        for got in outs:
            assert torch.equal(got[4], ref[4])
            for b in range(B):
                for k in range(4):
                    assert torch.equal(got[k][b, :cnt[b]], ref[k][b, :cnt[b]]), (b, k)
    with pytest.raises(ValueError):
        net_f32.detect(np.zeros((1, 40, 64, 3), np.float32))
    with pytest.raises(ValueError):
        net_f32.detect(np.zeros((1, 64, 64, 3), np.float32), max_boxes=0)


def test_load_state_equals_load_weight_stream(net_f32, weight_stream, tmp_path):
    """a checkpoint in the reference's TF variable naming (convert_weight counterpart) loads to the same network."""
    import torch
    from yolo_amd import weights
    from yolo_amd.model import yolov3
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    ck = str(tmp_path / "ck.npz")
    np.savez(ck, **weights.stream_to_state(weight_stream, 80))
    net = yolov3(80, DEFAULT_ANCHORS, dtype="fp32").load_state(ck)
    x = np.random.default_rng(37).random((1, 64, 64, 3), dtype=np.float32)
    for a, b in zip(net.forward(x), net_f32.forward(x)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("cfg", ["5", "26", "61", "75", "81", "88", "96", "97", "98", "99", "115"])
def test_alternative_conv_tilings_vs_oracle(net_bf16, oracle, oracle_params, cfg):
    """The alternates the shipped library keeps beside its default tile choice (yolo_debug_set_conv_cfg: 5 = im2col kernel with
    the LDS epilogue, 26 = 16-wave halo tile, 61 = 256 x 128 halo tile on every 3x3 stride-1 layer, 75 = halo tiles with the
    register epilogue, 81 / 88 = im2col kernel on 16x16x32 MFMAs with the wave-private / register epilogue, 96 / 97 = direct-A 1x1
    kernel with 8 / 4 waves) meet the bf16 tolerance of test_forward_bf16_vs_oracle against the oracle.  (The measured-and-rejected
    tilings live in the XP build only: make -C yolo_amd/csrc XP=1.)"""
    from yolo_amd import _lib
    img = np.random.default_rng(43).random((3, 96, 160, 3), dtype=np.float32)
    ref = oracle.forward(oracle_params, img)
    _lib.set_conv_cfg(cfg)
    try:
        got = [f.cpu().numpy() for f in net_bf16.forward(img)]
    finally:
        _lib.set_conv_cfg(None)
    for a, b in zip(got, ref):
        ok, maxerr, rms = rel_rms_ok(a, b, 1e-1)
        assert ok, (cfg, b.shape, maxerr, rms)
        assert float(np.sqrt(np.mean((a - b) ** 2))) <= 3e-2 * rms


def test_direct_1x1_kernel_equals_im2col_kernel(net_bf16):
    """bf16: the direct-A 1x1 kernel (activations loaded straight into MFMA operand registers, 96 / 97) multiplies the same
    operand pieces in the same k order as the im2col kernel with the register epilogue (88): bit-identical feature maps, also on
    tile tails (96 x 160, batch 3) and at 416 x 416 (where the default choice mixes 97 and 88)."""
    import torch
    from yolo_amd import _lib
    rng = np.random.default_rng(72)
    for B, H, W in ((3, 96, 160), (2, 416, 416)):
        x = rng.random((B, H, W, 3), dtype=np.float32)
        outs = {}
        try:
            for cfg in (88, 96, 97, None):
                _lib.set_conv_cfg(cfg)
                outs[cfg] = [f.clone() for f in net_bf16.forward(x)]
        finally:
            _lib.set_conv_cfg(None)
        for cfg in (96, 97, None):
            for fa, fb in zip(outs[88], outs[cfg]):
                assert torch.equal(fa, fb), (cfg, float((fa - fb).abs().max()))


@pytest.mark.parametrize("mode", ["tf", "py"])
def test_nms_short_and_long_lists_side_by_side(torch_cuda, oracle, mode):
    """the selection kernel settles lists of <= 64 candidates with one wave each (four lists per workgroup) and longer ones
    with a workgroup each: classes with 0, 1, 63, 64, 65 and 300 candidates in ONE call, clustered boxes (long suppression
    chains) and tied scores, give the oracle's keep-indices on both sides of the boundary."""
    from yolo_amd.utils import nms_utils
    rng = np.random.default_rng(64)
    n, counts = 400, (0, 1, 63, 64, 65, 300, 64, 2)
    C = len(counts)
    boxes = _boxes(rng, n, clustered=12, jitter=4.0)
    scores = np.zeros((n, C), np.float32)
    for c, k in enumerate(counts):
        pick = rng.permutation(n)[:k]
        scores[pick, c] = (0.5 + 0.5 * rng.integers(0, 8, k) / 8).astype(np.float32)      # eight distinct values: many ties
    if mode == "tf":
        ob, osc, ol, oi, oc = nms_utils.gpu_nms_batched(np.stack([boxes, boxes[::-1]]), np.stack([scores, scores[::-1]]), C, 20, 0.4, 0.45)
        for b, (bx, sc) in enumerate(((boxes, scores), (boxes[::-1], scores[::-1]))):
            m = int(oc[b])
            _, _, rl, ri = oracle.gpu_nms(np.ascontiguousarray(bx), np.ascontiguousarray(sc), C, 20, 0.4, 0.45, return_indices=True)
            assert m == len(ri) and m > 0
            assert np.array_equal(oi[b, :m].cpu().numpy(), ri) and np.array_equal(ol[b, :m].cpu().numpy(), rl)
    else:
        for c, k in enumerate(counts):
            if k == 0:
                continue
            sel = np.nonzero(scores[:, c] > 0)[0]
            got = nms_utils.py_nms(boxes[sel], scores[sel, c], max_boxes=20, iou_thresh=0.45)
            assert got == [int(v) for v in oracle.py_nms(boxes[sel], scores[sel, c], 20, 0.45)], (c, k)


@pytest.mark.parametrize("mode", ["tf", "py"])
def test_nms_many_candidates_with_equal_scores(torch_cuda, oracle, mode):
    """more than 2048 candidates per class whose scores collide: the key ranges of the selection rounds are then cut
    inside the index bits of the keys (deep radix levels), and several rounds are needed when the best boxes
    suppress each other."""
    from yolo_amd.utils import nms_utils
    rng = np.random.default_rng(61)
    n, C = 6000, 3
    boxes = _boxes(rng, n, clustered=15, jitter=3.0)
    scores = np.empty((n, C), np.float32)
    scores[:, 0] = 0.5                                              # all equal
    scores[:, 1] = rng.integers(1, 4, n).astype(np.float32) / 4      # three distinct values
    scores[:, 2] = rng.random(n).astype(np.float32)
    if mode == "tf":
        ob, osc, ol, oi, oc = nms_utils.gpu_nms_batched(boxes[None], scores[None], C, 40, 0.2, 0.3)
        m = int(oc[0])
        _, _, rl, ri = oracle.gpu_nms(boxes, scores, C, 40, 0.2, 0.3, return_indices=True)
        assert m == len(ri) and m > 0
        assert np.array_equal(oi[0, :m].cpu().numpy(), ri) and np.array_equal(ol[0, :m].cpu().numpy(), rl)
    else:
        for c in range(C):
            got = nms_utils.py_nms(boxes, scores[:, c], max_boxes=40, iou_thresh=0.3)
            assert got == [int(v) for v in oracle.py_nms(boxes, scores[:, c], 40, 0.3)], c


def test_detect_batch64_416_properties(net_bf16, oracle):
    """BASELINE config 2 at full size (batch 64, 416 x 416, bf16, thresholds 50 / 0.5 / 0.5), through size-independent
    properties: images are independent (any sub-batch gives the same rows, bit for bit), and every image's detections
    are a fixed point of the oracle's NMS: per class sorted by score, all >= score_thresh, no kept pair above the IoU
    threshold, and re-running the oracle NMS on the kept boxes keeps them all."""
    import torch
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.random((64, 416, 416, 3), dtype=np.float32)).cuda()
    ob, osc, ol, oi, oc = [t.clone() for t in net_bf16.detect(x, 50, 0.5, 0.5)]
    cnt = oc.cpu().numpy()
    assert cnt.sum() > 1000 and cnt.max() <= 80 * 50
    for lo, hi in ((5, 6), (40, 43), (63, 64)):
        sb, ss, sl, si, sc = net_bf16.detect(x[lo:hi], 50, 0.5, 0.5)
        assert torch.equal(sc, oc[lo:hi])
        for j in range(hi - lo):
            m = int(cnt[lo + j])
            for a, b in ((sb, ob), (ss, osc), (sl, ol), (si, oi)):
                assert torch.equal(a[j, :m], b[lo + j, :m])
    fms = net_bf16.forward(x)
    boxes, _, _, scores = net_bf16.predict(fms, return_scores=True)
    for b in (0, 17, 63):
        m = int(cnt[b])
        kb, ks, kl, ki = [t[b, :m].cpu().numpy() for t in (ob, osc, ol, oi)]
        assert np.all(ks >= np.float32(0.5))
        assert np.array_equal(kb, boxes[b].cpu().numpy()[ki])                       # rows are the decoded boxes they index
        assert np.array_equal(ks, scores[b].cpu().numpy()[ki, kl])
        assert np.all(np.diff(kl) >= 0)                                             # classes ascending (nms_utils.py:23)
        for c in np.unique(kl):
            sel = kl == c
            s_c, b_c = ks[sel], kb[sel]
            assert np.all(s_c[:-1] >= s_c[1:]) and len(s_c) <= 50
            again = oracle.tf_non_max_suppression(b_c, s_c, 50, 0.5)
            assert np.array_equal(again, np.arange(len(s_c)))                       # idempotent: nothing kept suppresses anything kept
        # and the whole image equals the oracle's per-class NMS run on the SAME decoded boxes / scores
        _, _, rl, ri = oracle.gpu_nms(boxes[b].cpu().numpy(), scores[b].cpu().numpy(), 80, 50, 0.5, 0.5, return_indices=True)
        assert np.array_equal(ki, ri) and np.array_equal(kl, rl)


def test_register_epilogue_equals_lds_epilogue(net_bf16):
    """bf16: the register epilogue (swapped 16x16x32 MFMA operands, v_permlane16_swap, no LDS: configuration 75 for the
    3x3 stride-1 layers, 88 for the im2col kernel) computes every output element with the same products, accumulation order
    and roundings as the LDS-staged epilogues: the feature maps are bit-identical (also on M / N tile tails: 96 x 160, batch 3)."""
    import torch
    from yolo_amd import _lib
    rng = np.random.default_rng(71)
    for B, H, W in ((3, 96, 160), (2, 416, 416)):
        x = rng.random((B, H, W, 3), dtype=np.float32)
        outs = {}
        try:
            for cfg in ("61", "75", "81", "88"):
                _lib.set_conv_cfg(cfg)
                outs[cfg] = [f.clone() for f in net_bf16.forward(x)]
        finally:
            _lib.set_conv_cfg(None)
        for a, b in (("61", "75"), ("81", "88")):
            for fa, fb in zip(outs[a], outs[b]):
                assert torch.equal(fa, fb), (a, b, float((fa - fb).abs().max()))
