"""Eval bookkeeping: yolo_amd/utils/eval_utils.py against vectors generated from the reference's own
evaluate_on_cpu (tests/golden/g5_eval.npz, oracle/gen_golden.py eval)."""
import numpy as np
import pytest

from conftest import load_golden

CASES = ("a", "b", "dup")


def _case(g, name):
    C = int(g[name + "/C"])
    y_true = [g["%s/y_true%d" % (name, j)] for j in range(3)]
    y_pred = [g[name + "/pred_boxes"], g[name + "/pred_confs"], g[name + "/pred_probs"]]
    want = tuple(g[name + "/" + k] for k in ("true_positive", "true_labels", "pred_labels"))
    return C, y_pred, y_true, float(g[name + "/score_thresh"]), float(g[name + "/iou_thresh"]), want, g[name + "/recall_precision"]


def _as_arrays(dicts, C):
    return tuple(np.asarray([d[c] for c in range(C)], np.int64) for d in dicts)


@pytest.mark.parametrize("name", CASES)
def test_bookkeeping_matches_reference_with_oracle_nms(oracle, name):
    """host-side counting alone: the NMS is the oracle's numpy restatement of cpu_nms (pinned by g3)."""
    from yolo_amd.utils import eval_utils
    C, y_pred, y_true, st, it, want, rp = _case(load_golden("g5_eval.npz"), name)
    got = eval_utils.evaluate_on_cpu(y_pred, y_true, C, calc_now=False, score_thresh=st, iou_thresh=it, _nms=oracle.cpu_nms)
    for a, b in zip(_as_arrays(got, C), want):
        assert np.array_equal(a, b)
    rec, prec = eval_utils.evaluate_on_cpu(y_pred, y_true, C, calc_now=True, score_thresh=st, iou_thresh=it, _nms=oracle.cpu_nms)
    assert rec == rp[0] and prec == rp[1]


def test_calc_iou_matches_reference_fixture():
    from yolo_amd.utils import eval_utils
    g = load_golden("g3b_calc_iou.npz")
    got = eval_utils.calc_iou(g["pred"], g["true"])
    assert got.dtype == g["iou"].dtype and np.array_equal(got, g["iou"])


def test_image_without_ground_truth_raises_like_the_reference(oracle):
    from yolo_amd.utils import eval_utils
    C, y_pred, y_true, st, it, _, _ = _case(load_golden("g5_eval.npz"), "a")
    y_true = [a.copy() for a in y_true]
    for a in y_true:
        a[1] = 0
    with pytest.raises(IndexError):
        eval_utils.evaluate_on_cpu(y_pred, y_true, C, _nms=oracle.cpu_nms)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_evaluate_on_cpu_gpu_kernels_match_reference(torch_cuda, name):
    """evaluate_on_cpu with the numpy-semantics NMS kernel (yolo_nms, mode PY) reproduces the reference's counts."""
    from yolo_amd.utils import eval_utils
    C, y_pred, y_true, st, it, want, rp = _case(load_golden("g5_eval.npz"), name)
    got = eval_utils.evaluate_on_cpu(y_pred, y_true, C, calc_now=False, score_thresh=st, iou_thresh=it)
    for a, b in zip(_as_arrays(got, C), want):
        assert np.array_equal(a, b)
    rec, prec = eval_utils.evaluate_on_cpu(y_pred, y_true, C, score_thresh=st, iou_thresh=it)
    assert rec == rp[0] and prec == rp[1]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_evaluate_on_gpu_batched_equals_per_image_oracle(torch_cuda, oracle, name):
    """evaluate_on_gpu (one batched TF-semantics NMS call) = the same counting over the oracle's per-image gpu_nms."""
    from yolo_amd.utils import eval_utils
    C, y_pred, y_true, st, it, _, _ = _case(load_golden("g5_eval.npz"), name)
    op = eval_utils.gpu_nms_op(C, max_boxes=20, score_thresh=st, iou_thresh=0.45)
    got = eval_utils.evaluate_on_gpu(None, op, None, None, y_pred, y_true, C, calc_now=False, score_thresh=0.9, iou_thresh=it)

    def per_image(i):
        return oracle.gpu_nms(y_pred[0][i], (y_pred[1][i] * y_pred[2][i]).astype(np.float32), C, 20, st, 0.45)
    want = eval_utils._evaluate(per_image, y_pred, y_true, C, False, it)
    for a, b in zip(_as_arrays(got, C), _as_arrays(want, C)):
        assert np.array_equal(a, b)
    assert sum(got[2].values()) > 0
    dflt = eval_utils.evaluate_on_gpu(None, None, None, None, y_pred, y_true, C)
    assert len(dflt) == 2 and 0.0 <= dflt[0] <= 1.0 and 0.0 <= dflt[1] <= 1.0
    with pytest.raises(ValueError):
        eval_utils.evaluate_on_gpu(None, eval_utils.gpu_nms_op(C + 1), None, None, y_pred, y_true, C)
