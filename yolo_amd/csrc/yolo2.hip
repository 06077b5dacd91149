// YOLOv2 post-processing (config 4): replaces Detector.calc_output / get_boxes / calc_iou
// (reference yolo_v2/Detector.py:51-115): sigmoid/exp/softmax decode of the [S,S,5,5+C] logits and the
// class-agnostic greedy NMS over every (box, class) pair with prob >= threshold.
// Arithmetic mirrors the reference's numpy types: sigmoid / exp / softmax / probs in float32, box
// coordinates and IoU in float64 (the int64 `offset` array and the Python-list anchors promote them).
// The NMS uses only +,-,*,/ and compares in IEEE double (_rn intrinsics, -ffp-contract=off): given the
// same boxes/probs it is bit-identical to the CPU oracle.
#include "common.h"

namespace yolo {

typedef unsigned long long u64;
constexpr u64 kDead2 = ~0ull;

struct Decode2K {
    const float* logits;   // [B, S, S, NB, 5 + C]
    double* boxes;         // [B, S*S*NB, 4]  (cx, cy, w, h) in network-input pixels
    float* probs;          // [B, S*S*NB, C]
    int S, NB, C, image_size;
    double aw[16], ah[16]; // anchors: widths then heights (config.py:13)
};

__global__ __launch_bounds__(256) void decode2_kernel(Decode2K k, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (image, cell, box)
    if (i >= total) return;
    const int per_img = k.S * k.S * k.NB;
    const int j = i % per_img;
    const int a = j % k.NB;
    const int cell = j / k.NB;
    const int y = cell / k.S, x = cell - y * k.S;
    const float* t = k.logits + (long long)i * (5 + k.C);
    // Detector.py:100-103 (float32 sigmoid/exp, then float64), :55 (* image_size)
    const float sx = 1.0f / (1.0f + expf(-1.0f * t[0]));
    const float sy = 1.0f / (1.0f + expf(-1.0f * t[1]));
    double* b = k.boxes + (long long)i * 4;
    b[0] = ((double)sx + (double)x) / (double)k.S * (double)k.image_size;
    b[1] = ((double)sy + (double)y) / (double)k.S * (double)k.image_size;
    b[2] = (double)expf(t[2]) * k.aw[a] / (double)k.S * (double)k.image_size;
    b[3] = (double)expf(t[3]) * k.ah[a] / (double)k.S * (double)k.image_size;
    const float conf = 1.0f / (1.0f + expf(-1.0f * t[4]));  // :57
    float sum = 0.f;
    for (int c = 0; c < k.C; ++c) sum += expf(t[5 + c]);     // :61 (naive softmax, no max subtraction)
    float* p = k.probs + (long long)i * k.C;
    for (int c = 0; c < k.C; ++c) p[c] = expf(t[5 + c]) / sum * conf;   // :61-62
}

// Detector.py:107-115 in float64
__device__ __forceinline__ double iou2(const double* a, const double* b) {
    const double ax0 = __dsub_rn(a[0], __dmul_rn(0.5, a[2])), ax1 = __dadd_rn(a[0], __dmul_rn(0.5, a[2]));
    const double bx0 = __dsub_rn(b[0], __dmul_rn(0.5, b[2])), bx1 = __dadd_rn(b[0], __dmul_rn(0.5, b[2]));
    const double ay0 = __dsub_rn(a[1], __dmul_rn(0.5, a[3])), ay1 = __dadd_rn(a[1], __dmul_rn(0.5, a[3]));
    const double by0 = __dsub_rn(b[1], __dmul_rn(0.5, b[3])), by1 = __dadd_rn(b[1], __dmul_rn(0.5, b[3]));
    const double width = __dsub_rn(fmin(ax1, bx1), fmax(ax0, bx0));
    const double height = __dsub_rn(fmin(ay1, by1), fmax(ay0, by0));
    const double inter = (width <= 0.0 || height <= 0.0) ? 0.0 : __dmul_rn(width, height);
    return __ddiv_rn(inter, __dsub_rn(__dadd_rn(__dmul_rn(a[2], a[3]), __dmul_rn(b[2], b[3])), inter));
}

// One workgroup per image.  Candidates = (box, class) pairs with prob >= threshold in np.nonzero order
// (flat index f = box*C + c); processed best-first (prob descending, ties in DESCENDING f: the reference's
// argsort()[::-1]); a kept candidate zeroes every later one with IoU > iou_thresh (:72-78).
__global__ __launch_bounds__(256) void nms2_kernel(const double* __restrict__ boxes, const float* __restrict__ probs,
                                                   int n, int C, float thr, double iou_thr, int max_det,
                                                   u64* __restrict__ keys_ws, int* __restrict__ out_count,
                                                   int* __restrict__ out_cls, double* __restrict__ out_boxes,
                                                   float* __restrict__ out_probs, int* __restrict__ out_flat,
                                                   int* __restrict__ out_truncated) {
    __shared__ int s_count;
    __shared__ u64 wbest[4];
    __shared__ u64 bcast;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long nc = (long long)n * C;
    const float* pr = probs + b * nc;
    const double* bx = boxes + (long long)b * n * 4;
    u64* keys = keys_ws + b * nc;
    if (tid == 0) s_count = 0;
    __syncthreads();
    for (int f = tid; f < nc; f += 256) {
        const float p = pr[f];
        if (p >= thr) {                                                      // :63
            const unsigned bits = __float_as_uint(p + 0.0f);
            const unsigned u = bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u);
            keys[atomicAdd(&s_count, 1)] = ((u64)(~u) << 32) | (unsigned)(~(unsigned)f);
        }
    }
    __syncthreads();
    const int k = s_count;
    int ndet = 0;
    bool have = false;
    u64 selkey = 0;
    double sb[4] = {0, 0, 0, 0};
    bool truncated = false;
    while (true) {
        u64 best = kDead2;
        for (int i = tid; i < k; i += 256) {
            const u64 key = keys[i];
            if (key == kDead2) continue;
            if (have) {
                if (key == selkey) { keys[i] = kDead2; continue; }
                const unsigned f = ~(unsigned)key;
                if (iou2(sb, bx + (long long)(f / C) * 4) > iou_thr) { keys[i] = kDead2; continue; }   // :76
            }
            best = key < best ? key : best;
        }
        for (int d = 32; d > 0; d >>= 1) {
            const u64 o = __shfl_xor(best, d);
            best = o < best ? o : best;
        }
        if (lane == 0) wbest[wave] = best;
        __syncthreads();
        if (tid == 0) {
            u64 m = wbest[0];
            for (int w = 1; w < 4; ++w) m = wbest[w] < m ? wbest[w] : m;
            bcast = m;
        }
        __syncthreads();
        selkey = bcast;
        if (selkey == kDead2) break;
        if (ndet == max_det) { truncated = true; break; }      // a survivor that has no output row left
        const unsigned f = ~(unsigned)selkey;
        const int box = (int)(f / C);
        for (int e = 0; e < 4; ++e) sb[e] = bx[(long long)box * 4 + e];
        have = true;
        if (tid == 0) {
            const long long o = (long long)b * max_det + ndet;
            int cls = 0;
            while (cls < C && !(pr[(long long)box * C + cls] >= thr)) ++cls;   // :67 argmax of the bool mask
            out_cls[o] = cls;
            for (int e = 0; e < 4; ++e) out_boxes[o * 4 + e] = sb[e];
            out_probs[o] = pr[f];
            if (out_flat) out_flat[o] = (int)f;
        }
        ++ndet;
    }
    // degenerate boxes (w*h == 0, inf, NaN: IoU is NaN, never > thr) let several classes of ONE box survive, so n rows are
    // not a bound; the reference (Detector.py:72-83) has no cap.  The count never exceeds the rows written; that rows were cut
    // is reported through out_truncated
    if (tid == 0) {
        out_count[b] = ndet;
        if (out_truncated) out_truncated[b] = truncated ? 1 : 0;
    }
}

}  // namespace yolo

using namespace yolo;

// logits [B,S,S,NB*(5+C)] f32 -> boxes f64 [B,S*S*NB,4] (cx,cy,w,h px), probs f32 [B,S*S*NB,C].
// anchors: HOST doubles, NB widths then NB heights (reference config.py:13 layout).
extern "C" int yolo2_decode(const float* logits, int B, int S, int NB, int C, int image_size, const double* anchors,
                            double* boxes, float* probs, void* stream) {
    YOLO_REQUIRE(logits && anchors && boxes && probs, YOLO_ERR_ARG, "yolo2_decode: null pointer");
    YOLO_REQUIRE(B > 0 && S > 0 && NB > 0 && NB <= 16 && C > 0 && image_size > 0, YOLO_ERR_ARG, "yolo2_decode: bad sizes");
    Decode2K k;
    k.logits = logits; k.boxes = boxes; k.probs = probs; k.S = S; k.NB = NB; k.C = C; k.image_size = image_size;
    for (int a = 0; a < NB; ++a) { k.aw[a] = anchors[a]; k.ah[a] = anchors[NB + a]; }
    const long long total = (long long)B * S * S * NB;
    YOLO_REQUIRE(total < (1ll << 31), YOLO_ERR_SHAPE, "yolo2_decode: too many boxes");
    hipLaunchKernelGGL(decode2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k, (int)total);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

extern "C" size_t yolo2_nms_workspace_bytes(int B, int n, int C) {
    if (B <= 0 || n <= 0 || C <= 0) return 0;
    return align_up((size_t)B * n * C * 8, 256);
}

// Class-agnostic NMS per image (Detector.py:63-83).  Outputs at rows [i*max_det, ...): class id (first class
// of the box with prob >= threshold, :67), box (cx,cy,w,h) f64, prob f32, flat candidate index box*C+c.
extern "C" int yolo2_nms(const double* boxes, const float* probs, int B, int n, int C, float threshold, double iou_thresh,
                         int max_det, int32_t* out_counts, int32_t* out_cls, double* out_boxes, float* out_probs,
                         int32_t* out_flat, int32_t* out_truncated, void* workspace, size_t workspace_bytes, void* stream) {
    YOLO_REQUIRE(boxes && probs && out_counts && out_cls && out_boxes && out_probs && workspace, YOLO_ERR_ARG, "yolo2_nms: null pointer");
    YOLO_REQUIRE(B > 0 && n > 0 && C > 0 && max_det > 0 && (long long)n * C < (1ll << 31), YOLO_ERR_ARG, "yolo2_nms: bad sizes");
    YOLO_REQUIRE(workspace_bytes >= yolo2_nms_workspace_bytes(B, n, C), YOLO_ERR_WORKSPACE, "yolo2_nms: workspace too small");
    hipLaunchKernelGGL(nms2_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, boxes, probs, n, C, threshold, iou_thresh,
                       max_det, (u64*)workspace, out_counts, out_cls, out_boxes, out_probs, out_flat, out_truncated);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}
