// Anchor/grid decode: yolov3.reorg_layer + yolov3.predict (reference yolo_v3/model.py:69-153) and the
// caller's scores = confs * probs (yolo_v3/test_single_image.py:38) as ONE coalesced elementwise
// pass.  HBM-bound: reads the three feature maps once (n*(5+C) floats per image), writes boxes /
// confs / probs (+ scores) once.
//
// A feature map [B,g,g,3*(5+C)] is, per image, exactly the array [g*g*3][5+C] in the reference's
// (y, x, anchor) box order (model.py:81,123-125), so thread e of image b maps 1:1 to input element
// e of that image's scale-concatenated [n][5+C] view: reads are perfectly contiguous.
#include <cstring>
#include "decode_common.h"

namespace yolo {

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
            k.boxes[jb * 4 + attr] = decode_box_attr(k, s, local, row, attr);
        }
    }
}

// ---- predict at the HBM rate: one workgroup per tile of kDecTile boxes of one scale of one image ---------------------------------
// The tile's logits are one contiguous run of kDecTile * (5 + C) floats: they are read once with coalesced loads into LDS,
// the box conf is sigmoid-ed once per box, and every output tensor of the tile is again one contiguous run - probs / scores
// (16-byte aligned: the row length C * 4 bytes times kDecTile boxes) go out as 16-byte stores, boxes as one float4 per box.
// Arithmetic per element is exactly decode_kernel's (sigmoid_fast / decode_box_attr): same bits.
constexpr int kDecTile = 64;

__global__ __launch_bounds__(256) void decode_tile_kernel(DecodeK k, int tiles0, int tiles1) {
    extern __shared__ float dlog[];                 // [kDecTile * A] logits, then [kDecTile] conf
    const int b = blockIdx.y;
    int t = blockIdx.x, s = 0;
    if (t >= tiles0 + tiles1) { s = 2; t -= tiles0 + tiles1; }
    else if (t >= tiles0) { s = 1; t -= tiles0; }
    const int ns = k.off[s + 1] - k.off[s];
    const int j0 = t * kDecTile;                    // first box of the tile inside its scale
    const int nj = min(kDecTile, ns - j0);
    const int A = k.A, C = k.C;
    const float* src = k.fm[s] + ((long long)b * ns + j0) * A;
    const int nel = nj * A;
    for (int e = threadIdx.x; e < nel; e += 256) dlog[e] = src[e];
    float* conf = dlog + kDecTile * A;
    __syncthreads();
    const long long jb = (long long)b * k.n + k.off[s] + j0;       // first box of the tile in the [B, n] outputs
    if (threadIdx.x < nj) {
        const int j = threadIdx.x;
        const float* row = dlog + j * A;
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        f32x4 box;
#pragma unroll
        for (int a = 0; a < 4; ++a) box[a] = decode_box_attr(k, s, j0 + j, row, a);       // model.py:76-108,147-152
        ((f32x4*)k.boxes)[jb + j] = box;
        const float cf = sigmoid_fast(row[4]);                                              // model.py:134
        conf[j] = cf;
        k.confs[jb + j] = cf;
    }
    __syncthreads();
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    const int nout = nj * C;                        // multiple of 4 when C % 4 == 0 (checked on the host)
    f32x4* pp = (f32x4*)(k.probs + jb * C);
    f32x4* sp = k.scores ? (f32x4*)(k.scores + jb * C) : nullptr;
    const float inv_c = 1.0f / (float)C;
    for (int q = threadIdx.x; q < nout / 4; q += 256) {
        f32x4 pr, sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * q + i;
            int j = (int)((float)e * inv_c);        // e / C without an integer division, fixed up
            int c = e - j * C;
            if (c < 0) { c += C; --j; }
            if (c >= C) { c -= C; ++j; }
            const float p = sigmoid_fast(dlog[j * A + 5 + c]);                              // model.py:135
            pr[i] = p;
            sc[i] = conf[j] * p;                                                            // test_single_image.py:38
        }
        pp[q] = pr;
        if (sp) sp[q] = sc;
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
    decode_setup(k, H, W, class_num, anchors);
    const long long per_img = (long long)k.n * k.A;
    YOLO_REQUIRE(per_img < (1ll << 24), YOLO_ERR_SHAPE, "decode: more than 2^24 elements per image");
    const size_t lds = (size_t)(kDecTile * k.A + kDecTile) * sizeof(float);
    if (class_num % 4 == 0 && lds <= 64 * 1024 && ((uintptr_t)probs & 15) == 0 && ((uintptr_t)scores & 15) == 0 &&
        ((uintptr_t)boxes & 15) == 0) {
        // tiled kernel (16-byte stores need C % 4 == 0; COCO 80 / VOC 20 both qualify)
        int tiles[3];
        for (int s = 0; s < 3; ++s) tiles[s] = (k.off[s + 1] - k.off[s] + kDecTile - 1) / kDecTile;
        static std::atomic<bool> attr_done[16];   // zero-initialised; setting the attribute twice is harmless, the flag only must not tear
        int dev = 0;
        YOLO_HIP_CHECK(hipGetDevice(&dev));
        if (dev >= 16 || !attr_done[dev]) {
            YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)decode_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            if (dev < 16) attr_done[dev] = true;
        }
        hipLaunchKernelGGL(decode_tile_kernel, dim3((unsigned)(tiles[0] + tiles[1] + tiles[2]), B), dim3(256), lds, (hipStream_t)stream,
                           k, tiles[0], tiles[1]);
        YOLO_HIP_CHECK(hipGetLastError());
        return YOLO_OK;
    }
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
