/* CPU oracle for per-class NMS.  TEST INFRASTRUCTURE ONLY - never linked into the product.
 *
 * mode 0: `gpu_nms` (reference yolo_v3/utils/nms_utils.py:22-33): score >= thresh filter, then
 *         tf.image.non_max_suppression (TensorFlow 1.x non_max_suppression_op.cc, not vendored in the
 *         reference -> PARITY UNPINNED; normative tie rule: score desc, index asc).
 * mode 1: `cpu_nms`/`py_nms` (nms_utils.py:40-90): areas without min/max, +1 on intersection w/h,
 *         keep iff ovr <= thresh, keep[:max_boxes]; ties in DESCENDING index order
 *         (stable ascending argsort reversed, nms_utils.py:49).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (every op is a single IEEE binary32 operation).
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { float s; int32_t i; } cand_t;

static int g_mode;

static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a;
    const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    if (g_mode == 0) return (x->i < y->i) ? -1 : (x->i > y->i);
    return (x->i > y->i) ? -1 : (x->i < y->i);
}

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

/* TF 1.x IOUGreaterThanThreshold, boxes as (c0, c1, c2, c3) with pairs (0,2) and (1,3). */
static int tf_suppresses(const float* a, const float* b, float thr) {
    float a_lo0 = fminf_(a[0], a[2]), a_hi0 = fmaxf_(a[0], a[2]);
    float a_lo1 = fminf_(a[1], a[3]), a_hi1 = fmaxf_(a[1], a[3]);
    float b_lo0 = fminf_(b[0], b[2]), b_hi0 = fmaxf_(b[0], b[2]);
    float b_lo1 = fminf_(b[1], b[3]), b_hi1 = fmaxf_(b[1], b[3]);
    float area_a = (a_hi0 - a_lo0) * (a_hi1 - a_lo1);
    float area_b = (b_hi0 - b_lo0) * (b_hi1 - b_lo1);
    if (area_a <= 0.0f || area_b <= 0.0f) return 0;
    float d0 = fmaxf_(fminf_(a_hi0, b_hi0) - fmaxf_(a_lo0, b_lo0), 0.0f);
    float d1 = fmaxf_(fminf_(a_hi1, b_hi1) - fmaxf_(a_lo1, b_lo1), 0.0f);
    float inter = d0 * d1;
    float iou = inter / ((area_a + area_b) - inter);
    return iou > thr;
}

/* nms_utils.py:48,54-62; `sel` is the picked box i, `b` a remaining one. Returns 1 when b is dropped. */
static int py_drops(const float* sel, const float* b, float thr) {
    float area_s = (sel[2] - sel[0]) * (sel[3] - sel[1]);
    float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    float xx1 = fmaxf_(sel[0], b[0]);
    float yy1 = fmaxf_(sel[1], b[1]);
    float xx2 = fminf_(sel[2], b[2]);
    float yy2 = fminf_(sel[3], b[3]);
    float w = fmaxf_(0.0f, (xx2 - xx1) + 1.0f);
    float h = fmaxf_(0.0f, (yy2 - yy1) + 1.0f);
    float inter = w * h;
    float ovr = inter / ((area_s + area_b) - inter);
    return !(ovr <= thr);
}

int oracle_class_nms(const float* boxes, const float* scores, int n, int C, int c, int max_boxes,
                     float score_thresh, float iou_thresh, int mode, int32_t* out_idx, int32_t* out_count) {
    cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    int k = 0;
    for (int i = 0; i < n; ++i) {
        float s = scores[(size_t)i * C + c];
        if (s >= score_thresh) { cand[k].s = s; cand[k].i = i; ++k; }
    }
    g_mode = mode;
    qsort(cand, (size_t)k, sizeof(cand_t), cmp_cand);
    int m = 0;
    if (mode == 0) {
        for (int t = 0; t < k && m < max_boxes; ++t) {
            const float* bi = boxes + 4 * (size_t)cand[t].i;
            int ok = 1;
            for (int j = 0; j < m && ok; ++j)
                if (tf_suppresses(bi, boxes + 4 * (size_t)out_idx[j], iou_thresh)) ok = 0;
            if (ok) out_idx[m++] = cand[t].i;
        }
    } else {
        /* greedy with a dead mask == the reference's shrinking `order` array */
        char* dead = (char*)calloc((size_t)(k > 0 ? k : 1), 1);
        for (int t = 0; t < k && m < max_boxes; ++t) {
            if (dead[t]) continue;
            const float* bi = boxes + 4 * (size_t)cand[t].i;
            out_idx[m++] = cand[t].i;
            for (int u = t + 1; u < k; ++u)
                if (!dead[u] && py_drops(bi, boxes + 4 * (size_t)cand[u].i, iou_thresh)) dead[u] = 1;
        }
        free(dead);
    }
    *out_count = m;
    free(cand);
    return 0;
}
