// Shared by decode.hip (yolo_decode / yolo_reorg_boxes) and nms.hip (the fused decode + candidate compaction of
// yolo_detect_post): the decode parameters and the per-element arithmetic of yolov3.reorg_layer / predict
// (reference yolo_v3/model.py:69-153).  Both translation units are built with -ffp-contract=off, so an element
// decoded by either kernel has the same bits.
#pragma once
#include "common.h"

namespace yolo {

struct DecodeK {
    const float* fm[3];
    float* boxes;
    float* confs;
    float* probs;
    float* scores;
    int n, C, A;          // boxes per image, classes, attrs = 5 + C
    int corners;          // 1: x_min,y_min,x_max,y_max (predict) ; 0: cx,cy,w,h (reorg_layer)
    int off[4];           // first box of each scale, off[3] = n
    int gw[3], gh[3];
    float rw[3], rh[3];   // ratio (model.py:78): img / grid
    float aw[3][3], ah[3][3];   // rescaled anchors (model.py:80): anchor / ratio
};

// Geometry of the three scales for an H x W input and the 9 anchors (HOST float32 [9,2]), model.py:76-80,116-118.
static inline void decode_setup(DecodeK& k, int H, int W, int class_num, const float* anchors) {
    k.C = class_num; k.A = 5 + class_num;
    int off = 0;
    for (int s = 0; s < 3; ++s) {
        const int r = 32 >> s;
        k.gh[s] = H / r; k.gw[s] = W / r;
        k.off[s] = off;
        off += k.gh[s] * k.gw[s] * 3;
        // model.py:78: ratio = cast(img_size / grid_size, float32)  (true division in float64, then cast)
        k.rh[s] = (float)((double)H / k.gh[s]);
        k.rw[s] = (float)((double)W / k.gw[s]);
        const float* anc = anchors + 2 * 3 * (2 - s);           // model.py:116-118: fm1 <- anchors[6:9] ...
        for (int a = 0; a < 3; ++a) {
            k.aw[s][a] = anc[2 * a] / k.rw[s];                  // model.py:80 (float32 / float32)
            k.ah[s][a] = anc[2 * a + 1] / k.rh[s];
        }
    }
    k.off[3] = off; k.n = off; k.corners = 1;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// conf / prob path (95 % of the elements): v_exp_f32 + v_rcp_f32, ~2 ulp - far inside the 1e-3 budget
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.0f + __expf(-x)); }

// Box attribute attr (0..3) of box `local` of scale s from the box's raw logits t0..t3 (row[0..3]).
__device__ __forceinline__ float decode_box_attr(const DecodeK& k, int s, int local, const float* row, int attr) {
    const int cell = local / 3;
    const int a = local - cell * 3;
    const int y = cell / k.gw[s];
    const int x = cell - y * k.gw[s];
    const bool isx = (attr & 1) == 0;                            // attr 0,2 -> x ; 1,3 -> y
    const float t_c = isx ? row[0] : row[1];
    const float t_s = isx ? row[2] : row[3];
    const float ratio = isx ? k.rw[s] : k.rh[s];
    const float anc = isx ? k.aw[s][a] : k.ah[s][a];
    const float g = isx ? (float)x : (float)y;
    const float c = __fmul_rn(__fadd_rn(sigmoidf_(t_c), g), ratio);                  // model.py:88,99,101
    const float ex = fminf(fmaxf(expf(t_s), 1e-9f), 50.0f);                          // model.py:103
    const float sz = __fmul_rn(__fmul_rn(ex, anc), ratio);                           // model.py:103,105
    if (k.corners) {
        const float half = sz / 2.0f;                                                // model.py:148-151
        return attr < 2 ? __fsub_rn(c, half) : __fadd_rn(c, half);
    }
    return attr < 2 ? c : sz;                                                        // model.py:107
}

}  // namespace yolo
