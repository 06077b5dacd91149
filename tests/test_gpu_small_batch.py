"""GPU tests of the opt-in small-batch plan (yolo_net_set_small_batch_plan): layers whose grid has
fewer than 128 tiles split their K loop over several workgroups and a second kernel sums the f32
partial slices in a fixed order.  The summation order differs from the default plan's; a last-bit
difference in one f32 sum flips bf16 roundings (2^-8 relative) that 75 layers then spread, so two
equally valid bf16 runs differ by about as much as either differs from fp32 (measured rms 0.7 - 0.9 %
of the map's rms).  The checks: the whole-stack bf16 tolerance against the default plan
(|a-b| <= 1e-1*(|b| + rms(b)), rms(a-b) <= 3e-2*rms(b)), the same bound against the fp32 oracle that
the default plan is held to (tests/test_gpu_parity.py::test_forward_bf16_vs_oracle), and an error
against the oracle no larger than 1.25x the default plan's."""
import numpy as np
import pytest

from conftest import rel_rms_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets(torch_cuda, oracle, weight_stream):
    from yolo_amd.model import yolov3
    plain = yolov3(80, oracle.YOLO_ANCHORS, dtype="bf16").load_weight_stream(weight_stream)
    split = yolov3(80, oracle.YOLO_ANCHORS, dtype="bf16", small_batch_plan=True).load_weight_stream(weight_stream)
    return plain, split


@pytest.mark.parametrize("shape", [(1, 416, 416), (2, 416, 416), (4, 416, 416), (1, 96, 160), (3, 160, 96), (1, 608, 608)])
def test_small_batch_plan_vs_default_plan(nets, shape):
    plain, split = nets
    B, H, W = shape
    img = np.random.default_rng(B * 1000 + H).random((B, H, W, 3), dtype=np.float32)
    ref = [f.cpu().numpy() for f in plain.forward(img)]
    got = [f.cpu().numpy() for f in split.forward(img)]
    again = [f.cpu().numpy() for f in split.forward(img)]
    for a, a2, b in zip(got, again, ref):
        assert a.shape == b.shape
        assert np.array_equal(a, a2), "split-K sums its slices in a fixed order: two runs must agree bit for bit"
        ok, maxerr, rms = rel_rms_ok(a, b, 1e-1)
        rmserr = float(np.sqrt(np.mean((a - b) ** 2)))
        print("split-K vs default", b.shape, "max|err| %.4g rms(err) %.4g rms(ref) %.4g" % (maxerr, rmserr, rms))
        assert ok, (b.shape, maxerr, rms)
        assert rmserr <= 3e-2 * rms, (rmserr, rms)


def test_small_batch_plan_vs_oracle(nets, oracle, oracle_params):
    plain, split = nets
    img = np.random.default_rng(5).random((1, 416, 416, 3), dtype=np.float32)
    ref = oracle.forward(oracle_params, img)
    for a, d, b in zip(split.forward(img), plain.forward(img), ref):
        a, d = a.cpu().numpy(), d.cpu().numpy()
        ok, maxerr, rms = rel_rms_ok(a, b, 1e-1)
        rmserr = float(np.sqrt(np.mean((a - b) ** 2)))
        rmsdef = float(np.sqrt(np.mean((d - b) ** 2)))
        print("vs oracle", b.shape, "split-K rms(err) %.4g default rms(err) %.4g rms(ref) %.4g" % (rmserr, rmsdef, rms))
        assert ok, (b.shape, maxerr, rms)
        assert rmserr <= 3e-2 * rms, (rmserr, rms)
        assert rmserr <= 1.25 * rmsdef, (rmserr, rmsdef)


def test_small_batch_plan_detect_and_large_batch(nets, torch_cuda):
    """detect() runs through the same plan; so do the lanes of a large batch."""
    torch = torch_cuda
    plain, split = nets
    img = np.random.default_rng(6).random((1, 416, 416, 3), dtype=np.float32)
    _, s0, _, _, n0 = plain.detect(img, score_thresh=0.3)
    b1, s1, l1, _, n1 = split.detect(img, score_thresh=0.3)
    assert b1.shape[-1] == 4 and s1.shape == l1.shape
    n0, n1 = int(n0[0]), int(n1[0])
    assert abs(n0 - n1) <= max(2, n0 // 10), (n0, n1)          # threshold crossings of a few candidates only
    # batch 64 runs as three lanes of 21 - 22 images: their 13 x 13 launches still have fewer than 128 tiles, so the plan
    # applies there too (each lane sums into the scratch area of its own workspace slice)
    img64 = torch.rand(64, 416, 416, 3, device="cuda")
    for a, b in zip(split.forward(img64), plain.forward(img64)):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        ok, maxerr, rms = rel_rms_ok(a, b, 1e-1)
        assert ok, (b.shape, maxerr, rms)
        assert float(np.sqrt(np.mean((a - b) ** 2))) <= 3e-2 * rms
