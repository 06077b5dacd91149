"""GPU tests of the rows either side of the hot path and of the bf16 detection tolerance:
sharded batch = unsharded batch, device-side box rescale + wire rows against the reference's host arithmetic,
bf16 detections against the fp32 oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net_bf16(torch_cuda, oracle, weight_stream):
    from yolo_amd.model import yolov3
    return yolov3(80, oracle.YOLO_ANCHORS, dtype="bf16").load_weight_stream(weight_stream)


def test_sharded_batch_equals_unsharded(net_bf16):
    """BASELINE config 3 on one GPU: a 128-image batch cut with shard_bounds into G = 2 / 4 / 8 contiguous shards (as G
    processes would), each shard through detect -> pack_detections_device(image_base = lo); the concatenated row lists
    equal the unsharded run's rows bit for bit.  Uneven shards (G = 3, 7 of 13 images) as well."""
    import torch
    from yolo_amd.shard import pack_detections_device, shard_bounds
    rng = np.random.default_rng(2)
    for n, size, groups in ((128, 416, (2, 4, 8)), (13, 96, (3, 7))):
        x = torch.from_numpy(rng.random((n, size, size, 3), dtype=np.float32)).cuda()
        thr = 0.5 if size == 416 else 0.05
        outs = []
        for j in range(0, n, 64):                                     # local steps of 64 like bench.py's sharded pass
            ob, osc, ol, _, oc = net_bf16.detect(x[j:j + 64], 50, thr, 0.5)
            rows, off = pack_detections_device(ob, osc, ol, oc, None, j)
            outs.append((rows.clone(), off.clone()))
        ref_rows = torch.cat([r for r, _ in outs])
        assert ref_rows.shape[0] > 50
        assert bool((ref_rows[1:, 0] >= ref_rows[:-1, 0]).all()) and int(ref_rows[-1, 0]) < n      # image order
        for G in groups:
            parts = []
            for rank in range(G):
                lo, hi = shard_bounds(n, rank, G)
                for j in range(lo, hi, 64):
                    k = min(hi, j + 64)
                    ob, osc, ol, _, oc = net_bf16.detect(x[j:k], 50, thr, 0.5)
                    rows, off = pack_detections_device(ob, osc, ol, oc, None, j)
                    assert int(off[-1]) == rows.shape[0]
                    parts.append(rows.clone())
            got = torch.cat(parts)
            assert got.shape == ref_rows.shape and torch.equal(got, ref_rows), G


def test_pack_detections_rescale_vs_reference_arithmetic(torch_cuda, oracle):
    """yolo_pack_detections = the reference's host-side rescale (test_single_image.py:42-45: float32 boxes *= python float
    ratio, in place) + row format, bit-exact against that arithmetic on the oracle's NMS output."""
    from yolo_amd.shard import pack_detections_device
    from yolo_amd.utils import nms_utils
    rng = np.random.default_rng(12)
    B, n, C = 3, 2000, 5
    xy = rng.random((B, n, 2)) * 380
    wh = rng.random((B, n, 2)) * 100 + 1
    boxes = np.concatenate([xy, xy + wh], 2).astype(np.float32)
    scores = (rng.random((B, n, C)) ** 3).astype(np.float32)
    scores[1] = 0.0                                                   # an image without detections in the middle
    ob, osc, ol, oi, oc = nms_utils.gpu_nms_batched(boxes, scores, C, 15, 0.5, 0.45)
    sizes = [(768, 576), (1920, 1080), (333, 517)]                    # (width_ori, height_ori) per image
    new_size = (416, 416)
    scale = np.asarray([[w / float(new_size[0]), h / float(new_size[1])] for w, h in sizes], np.float32)
    rows, off = pack_detections_device(ob, osc, ol, oc, scale, image_base=40)
    rows, off = rows.cpu().numpy(), off.cpu().numpy()
    assert off[0] == 0 and off[2] == off[1] and off[-1] == rows.shape[0] > 10
    for b in range(B):
        rb, rs, rl = oracle.gpu_nms(boxes[b], scores[b], C, 15, 0.5, 0.45)
        boxes_ = np.array(rb, np.float32).reshape(-1, 4)
        w, h = sizes[b]
        boxes_[:, 0] *= (w / float(new_size[0]))                      # test_single_image.py:42-45
        boxes_[:, 2] *= (w / float(new_size[0]))
        boxes_[:, 1] *= (h / float(new_size[1]))
        boxes_[:, 3] *= (h / float(new_size[1]))
        r = rows[off[b]:off[b + 1]]
        assert len(r) == len(rl)
        np.testing.assert_array_equal(r[:, 0], np.float32(40 + b))
        np.testing.assert_array_equal(r[:, 1:5], boxes_)
        np.testing.assert_array_equal(r[:, 5], rs)
        np.testing.assert_array_equal(r[:, 6].astype(np.int32), rl)
    rows2, off2 = pack_detections_device(ob, osc, ol, oc)             # no scale: boxes stay in network pixels
    assert np.array_equal(off2.cpu().numpy(), off)
    m0 = int(oc[0])
    np.testing.assert_array_equal(rows2[:m0, 1:5].cpu().numpy(), ob[0, :m0].cpu().numpy())


def test_pack_detections_rejects_misaligned_boxes(torch_cuda):
    """yolo_pack_detections loads one box per 16-byte access: a boxes pointer that is not 16-byte aligned (a view into a
    larger buffer) is refused with an argument error instead of being read with misaligned vector loads."""
    import torch
    from yolo_amd import _lib, _tensor
    n, cap = 2, 8
    buf = torch.zeros(n * cap * 4 + 4, device="cuda")
    boxes = buf[1:]                                                  # 4 bytes past a 16-byte boundary
    scores = torch.zeros(n, cap, device="cuda")
    labels = torch.zeros(n, cap, dtype=torch.int32, device="cuda")
    counts = torch.ones(n, dtype=torch.int32, device="cuda")
    rows = torch.zeros(n * cap, 7, device="cuda")
    offsets = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    with pytest.raises((ValueError, RuntimeError), match="16-byte"):
        _lib.check(_lib.lib().yolo_pack_detections(boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(), counts.data_ptr(), n, cap,
                                                   None, 0, rows.data_ptr(), offsets.data_ptr(), _tensor.stream_ptr()))


def _iou(a, b):
    iw = np.maximum(0.0, np.minimum(a[2], b[:, 2]) - np.maximum(a[0], b[:, 0]))
    ih = np.maximum(0.0, np.minimum(a[3], b[:, 3]) - np.maximum(a[1], b[:, 1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter)


def _iou_pairs(a, b):
    """IoU of box a[i] with box b[i]."""
    iw = np.maximum(0.0, np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]))
    ih = np.maximum(0.0, np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]))
    inter = iw * ih
    return inter / np.maximum((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter, 1e-12)


# stated bf16 tolerance (north_star "bf16 output within a stated tolerance"); measured values are in DESIGN.md section 2
BF16_SCORE_TOL = 0.03       # |score difference| of matched detections (measured 0.015)
BF16_MATCH_ALL = 0.80       # of ALL detections of one path, the share the other path reproduces with IoU >= 0.9 (measured 0.85 - 0.88)
BF16_MATCH_ALL_50 = 0.90    # of ALL detections, the share reproduced at IoU >= 0.5 in the other side's list taken at score_thresh - BF16_SCORE_TOL (measured 0.927 - 0.95)
BF16_MATCH_CLEAR = 0.95     # of the detections scoring >= score_thresh + BF16_SCORE_TOL, the share reproduced with IoU >= 0.5


def test_bf16_detections_vs_fp32_oracle(net_bf16, oracle, oracle_params):
    """bf16 conv stack (bf16 storage + MFMA inputs, f32 accumulate) against the fp32 CPU oracle, 2 x 416 x 416.
    (a) Candidate level, ALL 2 x 10647 boxes x 80 classes: |score - oracle score| <= BF16_SCORE_TOL everywhere (mean <= 1e-3),
        decoded box vs oracle box of the same index: IoU >= 0.9 for >= 98 % of the boxes and >= 0.7 for every box.
    (b) Detection level, gpu_nms defaults (50, 0.5, 0.5), ALL detections, both directions.  A detection is reproduced when the
        other side holds one of the same class whose score differs by <= BF16_SCORE_TOL and whose box overlaps it with
          * IoU >= 0.5 - the overlap above which NMS itself treats two boxes as one object - where the other side's list is taken
            at score_thresh - BF16_SCORE_TOL, so that a score that merely crosses the threshold does not count as a miss:
            asserted >= BF16_MATCH_ALL_50 of ALL detections of every image and direction, >= 0.95 pooled (measured 0.927 / 0.978);
          * IoU >= 0.9 against the other side's list at the same threshold: asserted >= BF16_MATCH_ALL.  The seeded random-init
            weights give every threshold hundreds of candidates within the bf16 score error of it and long chains of
            near-identical overlapping boxes, so crossings and flipped greedy choices bound this share well below 1 for ANY
            8-bit-mantissa path; widening the head's logit spread or moving the threshold into a gap of the oracle's score
            histogram does not change that (gaps are ~1e-3 wide, the score error 2e-2; measured 0.89 - 0.92 for head gains
            0.55 - 1.5, round 3) - which is why (a) and the IoU 0.5 share are the statements that carry weight.
          * of the detections that clear the threshold by more than the tolerance, >= BF16_MATCH_CLEAR at IoU >= 0.5."""
    thr = 0.5
    img = np.random.default_rng(1).random((2, 416, 416, 3), dtype=np.float32)
    # ---- (a) candidates
    fms_ref = oracle.forward(oracle_params, img)
    rbx, rcf, rpr = oracle.predict(fms_ref, oracle.YOLO_ANCHORS, (416, 416))
    rsc = (rcf * rpr).astype(np.float32)
    gbx, _, _, gsc = [t.cpu().numpy() for t in net_bf16.predict(net_bf16.forward(img), return_scores=True)]
    dsc = np.abs(gsc - rsc)
    biou = _iou_pairs(gbx.reshape(-1, 4), rbx.reshape(-1, 4))
    cand = {"max_dscore": float(dsc.max()), "mean_dscore": float(dsc.mean()), "iou_ge_0.9": float((biou >= 0.9).mean()),
            "iou_min": float(biou.min())}
    print("bf16 vs fp32 oracle candidates:", cand)
    assert cand["max_dscore"] <= BF16_SCORE_TOL and cand["mean_dscore"] <= 1e-3, cand
    assert cand["iou_ge_0.9"] >= 0.98 and cand["iou_min"] >= 0.7, cand
    # ---- (b) detections
    det = {t: [x.cpu().numpy() for x in net_bf16.detect(img, 50, t, 0.5)] for t in (thr, thr - BF16_SCORE_TOL)}
    worst = {"all@0.9": 1.0, "all@0.5_crossing_proof": 1.0, "clear@0.5": 1.0, "dscore": 0.0}
    pooled = {"n": 0, "hit50": 0, "hit90": 0}
    for b in range(2):
        ref = {t: oracle.gpu_nms(rbx[b], rsc[b], 80, 50, t, 0.5) for t in (thr, thr - BF16_SCORE_TOL)}
        got = {}
        for t, (ob, osc, ol, _, oc) in det.items():
            m = int(oc[b])
            got[t] = (ob[b, :m], osc[b, :m], ol[b, :m])
        assert len(ref[thr][2]) > 20 and len(got[thr][2]) > 20

        def matched(A, Bq, iou_min, only_clear):
            ab, as_, al = A
            bb, bs, bl = Bq
            hit, n, dmax = 0, 0, 0.0
            for i in range(len(al)):
                if only_clear and as_[i] < thr + BF16_SCORE_TOL:
                    continue
                n += 1
                cand_i = np.nonzero(bl == al[i])[0]
                if len(cand_i) == 0:
                    continue
                iou = _iou(ab[i], bb[cand_i])
                ok = (iou >= iou_min) & (np.abs(bs[cand_i] - as_[i]) <= BF16_SCORE_TOL)
                if ok.any():
                    hit += 1
                    dmax = max(dmax, float(np.abs(bs[cand_i][ok] - as_[i]).min()))
            return hit, n, dmax
        lo = thr - BF16_SCORE_TOL
        for A, B_same, B_low in ((ref[thr], got[thr], got[lo]), (got[thr], ref[thr], ref[lo])):
            h9, n, d1 = matched(A, B_same, 0.9, False)
            h5, _, d2 = matched(A, B_low, 0.5, False)
            hc, nclear, d3 = matched(A, B_same, 0.5, True)
            assert nclear > 20
            worst["all@0.9"] = min(worst["all@0.9"], h9 / n)
            worst["all@0.5_crossing_proof"] = min(worst["all@0.5_crossing_proof"], h5 / n)
            worst["clear@0.5"] = min(worst["clear@0.5"], hc / nclear)
            worst["dscore"] = max(worst["dscore"], d1, d2, d3)
            pooled["n"] += n; pooled["hit50"] += h5; pooled["hit90"] += h9
        print("image %d: oracle %d detections, bf16 %d" % (b, len(ref[thr][2]), len(got[thr][2])))
    print("bf16 vs fp32 oracle detections (worst image / direction):", worst, "pooled:", pooled)
    assert worst["all@0.5_crossing_proof"] >= BF16_MATCH_ALL_50, worst
    assert pooled["hit50"] >= 0.95 * pooled["n"], pooled           # over both images and directions (measured 0.978)
    assert worst["all@0.9"] >= BF16_MATCH_ALL and worst["clear@0.5"] >= BF16_MATCH_CLEAR, worst


def test_pipeline_shape_change_with_pending_work(net_bf16):
    """DetectPipeline.run twice with different frame shapes while the compute stream still has work queued: the staging
    buffers are re-allocated, and the first copy into a new buffer is ordered behind everything the compute stream
    has been given (the caching allocator may hand back a block those kernels still read)."""
    import torch
    from yolo_amd.pipeline import DetectPipeline
    from yolo_amd.preprocess import preprocess
    rng = np.random.default_rng(5)
    pipe = DetectPipeline(net_bf16, (96, 96), 20, 0.02, 0.45)
    a = [rng.integers(0, 256, (4, 120, 160, 3), dtype=np.uint8) for _ in range(3)]
    b = [rng.integers(0, 256, (2, 64, 80, 3), dtype=np.uint8) for _ in range(3)]
    big = torch.from_numpy(rng.random((16, 416, 416, 3), dtype=np.float32)).cuda()
    got_a = [[t.clone() for t in o] for o in pipe.run(a)]
    net_bf16.detect(big, 20, 0.5, 0.5)                               # several ms of queued compute-stream work ...
    got_b = [[t.clone() for t in o] for o in pipe.run(b)]            # ... while the slots are re-allocated for a new shape
    for frames, got in zip(a + b, got_a + got_b):
        ref = net_bf16.detect(preprocess(frames, (96, 96)), 20, 0.02, 0.45)
        cnt = ref[4].cpu().numpy()
        assert torch.equal(got[4], ref[4])
        for i in range(len(cnt)):
            for j in range(4):
                assert torch.equal(got[j][i, :cnt[i]], ref[j][i, :cnt[i]])


def test_detect_batch8_608_eval_threshold_properties(net_bf16, oracle):
    """BASELINE config 5 end to end: forward + predict + conf*prob + per-class NMS at 608 x 608 (model.py:22: the size comes
    from the input), batch 8, score_thresh 0.001 / max_boxes 50 (the eval-style call of eval.py:93-95): 22743 boxes x 80
    classes per image, nearly all of them candidates.  Size-independent properties: sub-batches give the same rows, every
    class list is score-sorted and of length <= 50, and for a sample of (image, class) pairs the kept indices equal the
    oracle's NMS run on the SAME decoded boxes / scores."""
    import torch
    x = torch.from_numpy(np.random.default_rng(6).random((8, 608, 608, 3), dtype=np.float32)).cuda()
    ob, osc, ol, oi, oc = [t.clone() for t in net_bf16.detect(x, 50, 0.001, 0.5)]
    cnt = oc.cpu().numpy()
    assert net_bf16.img_size == (608, 608) and cnt.min() > 80 * 10 and cnt.max() <= 80 * 50
    sb, ss, sl, si, sc = net_bf16.detect(x[5:7], 50, 0.001, 0.5)
    assert torch.equal(sc, oc[5:7])
    for j in range(2):
        m = int(cnt[5 + j])
        for a, b in ((sb, ob), (ss, osc), (sl, ol), (si, oi)):
            assert torch.equal(a[j, :m], b[5 + j, :m])
    fms = net_bf16.forward(x)
    boxes, _, _, scores = net_bf16.predict(fms, return_scores=True)
    assert tuple(boxes.shape) == (8, 22743, 4)
    for b in (0, 7):
        m = int(cnt[b])
        ks, kl, ki = [t[b, :m].cpu().numpy() for t in (osc, ol, oi)]
        assert np.all(np.diff(kl) >= 0) and np.all(ks >= np.float32(0.001))
        bx, sx = boxes[b].cpu().numpy(), scores[b].cpu().numpy()
        for c in (0, 33, 79):
            sel = kl == c
            assert np.all(ks[sel][:-1] >= ks[sel][1:]) and sel.sum() <= 50
            ref = oracle.class_nms_c(bx, sx, c, 50, 0.001, 0.5, 0)
            assert np.array_equal(ki[sel], ref), (b, c)
