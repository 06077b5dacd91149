// Anchor/grid decode: yolov3.reorg_layer + yolov3.predict (reference yolo_v3/model.py:69-153) and the
// caller's scores = confs * probs (yolo_v3/test_single_image.py:38) as ONE coalesced elementwise
// pass.  HBM-bound: reads the three feature maps once (n*(5+C) floats per image), writes boxes /
// confs / probs (+ scores) once.
//
// A feature map [B,g,g,3*(5+C)] is, per image, exactly the array [g*g*3][5+C] in the reference's
// (y, x, anchor) box order (model.py:81,123-125), so thread e of image b maps 1:1 to input element
// e of that image's scale-concatenated [n][5+C] view: reads are perfectly contiguous.
#include <cstring>
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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// conf / prob path (95 % of the elements): v_exp_f32 + v_rcp_f32, ~2 ulp - far inside the 1e-3 budget
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void decode_kernel(DecodeK k) {
    const int b = blockIdx.y;
    const int per_img = k.n * k.A;
    const float inv_a = 1.0f / (float)k.A;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per_img; e += gridDim.x * 256) {
        // j = e / A without an integer division (e < 2^24 is checked on the host)
        int j = (int)((float)e * inv_a);
        int attr = e - j * k.A;
        if (attr < 0) { attr += k.A; --j; }
        if (attr >= k.A) { attr -= k.A; ++j; }
        const int s = (j >= k.off[1]) + (j >= k.off[2]);
        const int local = j - k.off[s];
        const int ns = k.off[s + 1] - k.off[s];
        const float* row = k.fm[s] + ((long long)b * ns + local) * k.A;
        const long long jb = (long long)b * k.n + j;
        if (attr >= 4 && !k.probs) continue;                             // reorg_layer: boxes only
        if (attr >= 5) {
            const float pr = sigmoid_fast(row[attr]);                    // model.py:135
            const long long o = jb * k.C + (attr - 5);
            k.probs[o] = pr;
            if (k.scores) k.scores[o] = sigmoid_fast(row[4]) * pr;       // test_single_image.py:38
        } else if (attr == 4) {
            k.confs[jb] = sigmoid_fast(row[4]);                          // model.py:134
        } else {
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
                k.boxes[jb * 4 + attr] = attr < 2 ? __fsub_rn(c, half) : __fadd_rn(c, half);
            } else {
                k.boxes[jb * 4 + attr] = attr < 2 ? c : sz;                                  // model.py:107
            }
        }
    }
}

}  // namespace yolo

extern "C" int yolo_decode(const float* fm1, const float* fm2, const float* fm3, int B, int H, int W, int class_num,
                           const float* anchors, float* boxes, float* confs, float* probs, float* scores, void* stream) {
    using namespace yolo;
    YOLO_REQUIRE(fm1 && fm2 && fm3 && anchors && boxes && confs && probs, YOLO_ERR_ARG, "decode: null pointer");
    YOLO_REQUIRE(B > 0 && class_num > 0 && B <= 65535, YOLO_ERR_ARG, "decode: bad B/class_num");
    YOLO_REQUIRE(H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, YOLO_ERR_SHAPE, "decode: H, W must be positive multiples of 32 (got %d x %d)", H, W);
    DecodeK k;
    k.fm[0] = fm1; k.fm[1] = fm2; k.fm[2] = fm3;
    k.boxes = boxes; k.confs = confs; k.probs = probs; k.scores = scores;
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
    const long long per_img = (long long)k.n * k.A;
    YOLO_REQUIRE(per_img < (1ll << 24), YOLO_ERR_SHAPE, "decode: more than 2^24 elements per image");
    unsigned gx = (unsigned)((per_img + 255) / 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(decode_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, k);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// reorg_layer (model.py:69-108) for ONE feature map: boxes [B,gh,gw,3,4] as (cx, cy, w, h) in input pixels.
// conf/prob logits are plain slices of the feature map and are not recomputed here.
extern "C" int yolo_reorg_boxes(const float* fm, int B, int gh, int gw, int H, int W, int class_num,
                                const float* anchors3, float* boxes, void* stream) {
    using namespace yolo;
    YOLO_REQUIRE(fm && anchors3 && boxes, YOLO_ERR_ARG, "reorg: null pointer");
    YOLO_REQUIRE(B > 0 && B <= 65535 && gh > 0 && gw > 0 && H > 0 && W > 0 && class_num > 0, YOLO_ERR_ARG, "reorg: bad size");
    DecodeK k;
    memset(&k, 0, sizeof(k));
    k.fm[0] = k.fm[1] = k.fm[2] = fm;
    k.boxes = boxes;
    k.C = class_num; k.A = 5 + class_num;
    k.n = gh * gw * 3;
    k.off[0] = 0; k.off[1] = k.off[2] = k.off[3] = k.n;      // a single scale
    k.gh[0] = gh; k.gw[0] = gw;
    k.rh[0] = (float)((double)H / gh);
    k.rw[0] = (float)((double)W / gw);
    for (int a = 0; a < 3; ++a) {
        k.aw[0][a] = anchors3[2 * a] / k.rw[0];
        k.ah[0][a] = anchors3[2 * a + 1] / k.rh[0];
    }
    const long long per_img = (long long)k.n * k.A;
    YOLO_REQUIRE(per_img < (1ll << 24), YOLO_ERR_SHAPE, "reorg: more than 2^24 elements per image");
    unsigned gx = (unsigned)((per_img + 255) / 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(decode_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, k);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}
