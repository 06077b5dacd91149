"""GPU tests of the ingestion row: a Darknet .cfg + .weights FILE pair through `yolov3.from_cfg` /
`load_darknet_weights` against the reference's own `Darknet(cfgfile)` + `load_weights` result (G1 goldens,
yolo-v3/002/darknet.py:150-153,212-265), and a Darknet-order YOLOv2 stream through `darknet_stream_to_variables` ->
`load_variables` -> `build_networks` against the oracle on the converted parameters.  fp32 tolerance 1e-3 (conftest.rel_rms_ok)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_rms_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg_and_weights_files(tmp_path_factory, weight_stream):
    """The reference's cfg/yolov3.cfg re-rendered from its parsed block list (G6) + the seeded stream as a .weights file."""
    from yolo_amd import cfg, weights
    d = tmp_path_factory.mktemp("ingest")
    blocks = json.load(open(os.path.join(GOLDEN, "g6_yolov3_cfg_blocks.json")))
    cfg_path, w_path = str(d / "yolov3.cfg"), str(d / "yolov3.weights")
    with open(cfg_path, "w") as f:
        f.write("# rendered from tests/golden/g6_yolov3_cfg_blocks.json\n" + cfg.blocks_to_text(blocks))
    weights.write_darknet_weights(w_path, weight_stream)
    assert os.path.getsize(w_path) == 248007048            # the size of the public yolov3.weights
    return cfg_path, w_path


def test_from_cfg_and_weights_file_vs_reference_golden(torch_cuda, cfg_and_weights_files):
    from yolo_amd.model import yolov3
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    cfg_path, w_path = cfg_and_weights_files
    m = yolov3.from_cfg(cfg_path, w_path, dtype="fp32")
    assert m.class_num == 80 and m.net_size == (320, 320)
    np.testing.assert_array_equal(m.anchors, DEFAULT_ANCHORS)
    for size in (64, 96):
        g = load_golden("g1_convstack_%d.npz" % size)
        fms = m.forward(g["img"])
        for got, key in zip(fms, ("fm1", "fm2", "fm3")):
            assert tuple(got.shape) == g[key].shape
            ok, maxerr, rms = rel_rms_ok(got.cpu().numpy(), g[key], 1e-3)
            assert ok, (size, key, maxerr, rms)


def test_load_darknet_weights_file_equals_stream(torch_cuda, cfg_and_weights_files, weight_stream):
    """The file path of the loader (header skipped, float32 body) gives the handle the in-memory stream gives, and a
    truncated file is refused with the architecture's float count."""
    from yolo_amd.model import yolov3
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    _, w_path = cfg_and_weights_files
    g = load_golden("g1_convstack_64.npz")
    a = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_darknet_weights(w_path)
    b = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(weight_stream)
    for x, y in zip(a.forward(g["img"]), b.forward(g["img"])):
        assert torch_cuda.equal(x, y)
    short = w_path + ".short"
    with open(w_path, "rb") as f, open(short, "wb") as o:
        o.write(f.read(20 + 4 * 1000))
    with pytest.raises(ValueError):
        yolov3(80, DEFAULT_ANCHORS).load_darknet_weights(short)


def test_cfg_with_other_class_count_builds_that_net(torch_cuda, tmp_path, oracle):
    """A 20-class cfg (three detection convs of 75 filters) builds a 20-class handle: feature maps of the oracle on the
    same seeded 20-class stream."""
    import copy
    from yolo_amd import cfg, weights
    from yolo_amd.model import yolov3
    blocks = json.load(open(os.path.join(GOLDEN, "g6_yolov3_cfg_blocks.json")))
    voc = copy.deepcopy(blocks)
    for i, b in enumerate(voc):
        if b["type"] == "yolo":
            b["classes"] = "20"
            voc[i - 1]["filters"] = "75"
    voc[0]["width"], voc[0]["height"] = "416", "352"
    p = str(tmp_path / "voc.cfg")
    open(p, "w").write(cfg.blocks_to_text(voc))
    stream = weights.synth_darknet_stream(20, seed=77)
    wf = str(tmp_path / "voc.weights")
    weights.write_darknet_weights(wf, stream)
    m = yolov3.from_cfg(p, wf, dtype="fp32")
    assert m.class_num == 20 and m.net_size == (416, 352)
    img = np.random.default_rng(9).random((1, 64, 96, 3), dtype=np.float32)
    specs = oracle.yolov3_conv_specs(20)
    ref = oracle.forward(oracle.split_darknet_stream(specs, stream), img)
    for got, want in zip(m.forward(img), ref):
        assert tuple(got.shape) == want.shape and want.shape[-1] == 75
        ok, maxerr, rms = rel_rms_ok(got.cpu().numpy(), want, 1e-3)
        assert ok, (want.shape, maxerr, rms)


def test_yolov2_darknet_stream_through_the_gpu_net(torch_cuda):
    """Darknet-order yolov2 stream -> darknet_stream_to_variables -> load_variables -> build_networks equals the oracle's
    build_networks on the converted variables (fp32), and the first conv equals a direct evaluation of the DARKNET
    parameterisation (conv -> (x - mean) * gamma / sqrt(var + 1e-5) + beta -> leaky 0.1), which pins the mapping itself."""
    import torch
    from oracle import yolo2_oracle as O2
    from yolo_amd.yolo_v2.yolo import yolo_v2 as Y
    rng = np.random.default_rng(31)
    parts, first = [], None
    for name, (k, _, ci, co), bn in Y.conv_shapes(20, 5):
        if bn:
            head = [rng.normal(0, 0.1, co), rng.uniform(0.5, 1.5, co) * 0.85, rng.normal(0, 0.1, co), rng.uniform(0.5, 1.5, co)]
        else:
            head = [rng.normal(0, 0.1, co)]
        w = rng.normal(0, np.sqrt((2.0 if bn else 0.3) / (k * k * ci)), (co, ci, k, k))
        if first is None:
            first = [np.asarray(h, np.float32) for h in head] + [w.astype(np.float32)]
        parts += head + [w.ravel()]
    dark = np.concatenate(parts).astype(np.float32)
    var = Y.darknet_stream_to_variables(dark, 20, 5)
    net = Y.yolo_v2(False, dtype="fp32").load_variables(var)
    x = (rng.random((2, 96, 96, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    specs = O2.conv_specs()
    ref = O2.build_networks(O2.split_stream(specs, var), x)
    got = net.build_networks(x).cpu().numpy()
    assert got.shape == ref.shape == (2, 3, 3, 125)
    ok, maxerr, rms = rel_rms_ok(got, ref, 1e-3)
    assert ok and rms > 1e-2, (maxerr, rms)
    beta, gamma, mean, variance, w = first
    y = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), padding=1)
    y = (y - torch.from_numpy(mean)[None, :, None, None]) * torch.from_numpy(gamma / np.sqrt(variance + 1e-5))[None, :, None, None] \
        + torch.from_numpy(beta)[None, :, None, None]
    y = torch.maximum(0.1 * y, y).permute(0, 2, 3, 1).numpy()
    got0 = net.conv_layer(x, [3, 3, 3, 32], batch_norm=True, name="0_conv").cpu().numpy()
    ok, maxerr, rms = rel_rms_ok(got0, y, 1e-3)
    assert ok, (maxerr, rms)
