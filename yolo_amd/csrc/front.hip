// Fused front of Darknet-53 (bf16 path): conv 3->32 3x3 s1 + BN + leaky (the stem) followed by conv 32->64 3x3 s2
// + BN + leaky, the first two conv2d calls of darknet53_body (reference yolo_v3/utils/layer_utils.py:31-32; the
// stride-2 conv pads 1/1 explicitly, :11-20).  As two kernels this pair moves 133 MB (image) + 709 MB (stem output,
// written) + 709 MB (read again) + 354 MB at batch 64 and is bound by HBM; fused, the 416x416x32 stem output never
// leaves the CU: 133 MB in, 354 MB out.
//
// One workgroup of 64*TH threads walks over output tiles of TH x 16 pixels (of the H/2 x W/2 map):
//   1. image patch (2TH+3) x 35 x 3 f32 -> LDS (zero outside the image = the stem's SAME padding); the loads of the
//      NEXT tile's patch are issued before the compute phases and written to LDS after them.
//   2. stem on MFMA for the (2TH+1) x 33 stem pixels the tile needs (K = 27 padded to 32, two
//      v_mfma_f32_32x32x16_bf16 per 32 pixels, weights as the A operand so that a lane ends up with 4 x 4
//      consecutive channels of ONE pixel), bias + leaky, zero where the stem pixel lies outside the image (the
//      stride-2 conv's explicit padding), bf16 -> LDS stem patch.
//      Stem patch layout [channel chunk of 8][x parity][y][x / 2][16 B]: the stride-2 conv reads pixels 2 apart in x,
//      which are 16 B apart here, so every fragment read is a conflict-free ds_read_b128.
//   3. stride-2 conv as an implicit GEMM out of LDS: per wave 32 pixels x 32 channels, 18 k-slices (9 taps x 2 channel
//      halves); weights [64][288] bf16 staged once per workgroup (row stride 592 B: conflict-free fragment reads).
//   4. bias + leaky -> bf16 -> LDS staging -> 16-byte stores, 128 contiguous bytes per pixel.
#include <cstdlib>
#include "common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

// image patch (bf16) | stem patch (the output staging of a tile reuses its first TH * 16 * 144 bytes) | stride-2 weights | biases
template <int TH>
constexpr int front_lds_bytes() { return (((2 * TH + 3) * 108 * 2 + 15) & ~15) + 8 * (2 * TH + 1) * 17 * 16 + 64 * 592 + 96 * 4; }

#ifdef YOLO_CONV_STAMPS
// diagnostic build: cycles wave 0 of workgroup 0 spends in each phase, summed over its tiles (scripts/front_stamps.py)
__device__ unsigned long long g_front_stamps[8];
#define FSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); fst_[i] += t_ - ftl_; ftl_ = t_; } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

struct FrontK {
    const float* img;      // [B,H,W,3] f32
    const u32x4* w0;       // stem weights [32 cout][32 k] bf16, k = (ky*3+kx)*3 + c, BN folded
    const float* b0;       // [32]
    const char* w1;        // [>=64 cout][w1_row bytes] bf16, k = (ky*3+kx)*32 + ci, BN folded
    const float* b1;       // [64]
    char* out;             // [B,H/2,W/2,out_ld] bf16 (+ out_coff channels)
    int B, H, W, w1_row, out_ld, out_coff, act0, act1;
    int tiles_x, tiles_y, ntiles;
};

template <int ACT>
__device__ __forceinline__ float act(float v) {
    if constexpr (ACT == 1) return leaky_v3(v);                                  // model.py:8-12
    else if constexpr (ACT == 2) return __builtin_fmaxf(0.1f * v, v);            // yolo_v2.py:86
    else return v;
}

template <int TH, int ACT>
__global__ __launch_bounds__(64 * TH, TH == 8 ? 4 : 1) void front_kernel(FrontK p) {
    constexpr int NT = 64 * TH, NW = TH;
    constexpr int SH = 2 * TH + 1, SWH = 17;           // stem patch: SH x 33 pixels, x stored as (parity, x/2)
    constexpr int IH = SH + 2, IROW = 105, IWP = 108;  // image patch: IH x 35 pixels x 3 floats, rows padded to 108
    constexpr int IMG_BYTES = (IH * IWP * 2 + 15) & ~15;
    constexpr int PLANE = SH * SWH * 16;               // one (chunk, parity) plane of the stem patch
    constexpr int STEM_BYTES = 8 * PLANE;
    constexpr int W1_ROW = 592, W1_BYTES = 64 * W1_ROW;
    constexpr int ST_ROW = 144, ST_BYTES = TH * 16 * ST_ROW;
    constexpr int NE = (IH * IROW + NT - 1) / NT;      // image floats per thread
    constexpr int NG = SH + 1;                         // stem groups of 32 pixels: SH rows of x = 0..31, one of x = 32

    static_assert(IMG_BYTES % 16 == 0 && ST_BYTES <= STEM_BYTES && front_lds_bytes<TH>() == IMG_BYTES + STEM_BYTES + W1_BYTES + 384 &&
                  front_lds_bytes<TH>() <= 80 * 1024, "LDS budget: two workgroups per CU");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    unsigned short* const img = (unsigned short*)lds;       // bf16: rounded once when the patch lands, not per gathered tap
    char* const stem = lds + IMG_BYTES;
    char* const w1s = stem + STEM_BYTES;
    float* const bias = (float*)(w1s + W1_BYTES);           // [32] stem, [64] stride-2 conv
    char* const stage = stem;                               // the stem patch is dead once the tile's MFMAs have read it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int Ho = p.H >> 1, Wo = p.W >> 1;

    // ---- once per workgroup: stride-2 conv weights -> LDS, stem weights / biases -> registers --------------
    for (int i = tid; i < 64 * 36; i += NT) {          // 36 x 16 B per row of 288 bf16
        const int row = i / 36, ch = i - row * 36;
        *(u32x4*)(w1s + row * W1_ROW + ch * 16) = *(const u32x4*)(p.w1 + (long long)row * p.w1_row + ch * 16);
    }
    bf16x8 wf0, wf1;                                   // A[row = cout fr][slot (m, fh, j)] = w0[cout][tap] (common.h stem_slot_tap)
    {
        const unsigned short* wh = (const unsigned short*)p.w0;
        u16x8 h0, h1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t0 = stem_slot_tap(0, fh, j), t1 = stem_slot_tap(1, fh, j);
            h0[j] = t0 >= 0 ? wh[fr * 32 + t0] : (unsigned short)0;
            h1[j] = t1 >= 0 ? wh[fr * 32 + t1] : (unsigned short)0;
        }
        wf0 = __builtin_bit_cast(bf16x8, h0);
        wf1 = __builtin_bit_cast(bf16x8, h1);
    }
    if (tid < 32) bias[tid] = p.b0[tid];
    else if (tid < 96) bias[tid] = p.b1[tid - 32];
    const int wm = wave >> 1, wn = wave & 1;           // conv phase: wave tile = 32 pixels (2 tile rows) x 32 channels
    // accumulator register 4 q + i of a lane is channel 8 q + 4 fh + i: its biases are four consecutive floats per q, read from
    // LDS where they are used (keeping 32 of them in registers cost the second workgroup per CU)
    const f32x4* const b0p = (const f32x4*)(bias + 4 * fh);
    const f32x4* const b1p = (const f32x4*)(bias + 32 + wn * 32 + 4 * fh);
    // stem gather (stem_slot_tap): lanes fh = 0 take the first eight values of window rows 0 and 1, lanes fh = 1 those of row 2 and
    // the ninth value of each row; a row's nine values are contiguous bf16 in the image patch, so eight of them are ONE 16-byte LDS
    // read at a 2-byte-aligned address (gfx950 reads LDS unaligned) instead of eight ds_read_u16 + packing
    struct __attribute__((packed, aligned(2))) Run8 { u16x8 v; };

    const int per_img = p.tiles_x * p.tiles_y;
    const float inv_per_img = 1.0f / (float)per_img, inv_tx = 1.0f / (float)p.tiles_x;
    auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {     // tile < 2^24: float quotient + fix-up
        b = (int)((float)tile * inv_per_img);
        int r = tile - b * per_img;
        if (r < 0) { --b; r += per_img; } else if (r >= per_img) { ++b; r -= per_img; }
        int ty = (int)((float)r * inv_tx);
        int tx = r - ty * p.tiles_x;
        if (tx < 0) { --ty; tx += p.tiles_x; } else if (tx >= p.tiles_x) { ++ty; tx -= p.tiles_x; }
        oy0 = ty * TH;
        ox0 = tx * 16;
    };
    // image patch element e of this thread: tile-invariant LDS index, offset from the patch origin and (row, pixel)
    float pre[NE];
    int lidx[NE], goff[NE], ipos[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * NT;
        const bool v = idx < IH * IROW;
        const int iy = v ? idx / IROW : 0;
        const int rx = v ? idx - iy * IROW : 0;
        lidx[e] = v ? iy * IWP + rx : -1;
        goff[e] = iy * p.W * 3 + rx;
        ipos[e] = iy | ((rx / 3) << 8) | ((rx % 3) << 16);
    }
    auto load_image_patch = [&](int b, int oy0, int ox0) {      // global -> registers
        const int gy0 = 2 * oy0 - 2, gx0 = 2 * ox0 - 2;
        const bool interior = gy0 >= 0 && gy0 + IH <= p.H && gx0 >= 0 && gx0 + 35 <= p.W;      // wave-uniform
        if (interior) {
            const float* org = p.img + (((long long)b * p.H + gy0) * p.W + gx0) * 3;
#pragma unroll
            for (int e = 0; e < NE; ++e) pre[e] = org[lidx[e] >= 0 ? goff[e] : 0];
        } else {                                                  // out-of-image elements read a clamped pixel
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                int gy = gy0 + (ipos[e] & 255), gx = gx0 + ((ipos[e] >> 8) & 255);
                gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
                pre[e] = p.img[(((long long)b * p.H + gy) * p.W + gx) * 3 + (ipos[e] >> 16)];
            }
        }
    };
    auto store_image_patch = [&](int oy0, int ox0) {             // registers -> LDS, zeros outside the image (SAME padding)
        const int gy0 = 2 * oy0 - 2, gx0 = 2 * ox0 - 2;
        const bool interior = gy0 >= 0 && gy0 + IH <= p.H && gx0 >= 0 && gx0 + 35 <= p.W;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            bool inside = true;
            if (!interior) {
                const int gy = gy0 + (ipos[e] & 255), gx = gx0 + ((ipos[e] >> 8) & 255);
                inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            }
            if (lidx[e] >= 0) img[lidx[e]] = __builtin_bit_cast(unsigned short, (__bf16)(inside ? pre[e] : 0.f));
        }
    };

    int tile = blockIdx.x;
    int nb = 0, noy0 = 0, nox0 = 0;                    // origin of the tile whose image patch is in flight / in LDS
    if (tile < p.ntiles) {
        tile_origin(tile, nb, noy0, nox0);
        load_image_patch(nb, noy0, nox0);
        store_image_patch(noy0, nox0);
    }
    __syncthreads();

#ifdef YOLO_CONV_STAMPS
    unsigned long long fst_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ftl_ = __builtin_amdgcn_s_memtime();
#endif
    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = nb, oy0 = noy0, ox0 = nox0;
        const int next = tile + gridDim.x;
        if (next < p.ntiles) {
            tile_origin(next, nb, noy0, nox0);
            load_image_patch(nb, noy0, nox0);          // in flight during the two compute phases
        }
        FSTAMP(0);                                     // next patch: addresses + load issue

        // ---- stem: groups of 32 stem pixels; a wave takes its first two groups together (two independent read -> MFMA ->
        // activation -> write chains in flight instead of one), waves 0 and 1 then a third ----------------------------
        {
            auto gather = [&](int g, bf16x8& a0, bf16x8& a1, int& sy, int& sx, bool& pv) {
                pv = g < SH || fr < SH;                          // gather side: pixel = lane & 31
                sy = g < SH ? g : (pv ? fr : 0);                 // lanes past the column group read row 0, unused
                sx = g < SH ? fr : 32;
                const unsigned short* base = img + sy * IWP + sx * 3;
                const u16x8 g0 = ((const Run8*)(base + (fh ? 2 * IWP : 0)))->v;
                u16x8 g1;
                if (fh == 0) {
                    g1 = ((const Run8*)(base + IWP))->v;
                } else {
                    g1 = u16x8{base[8], base[IWP + 8], base[2 * IWP + 8], 0, 0, 0, 0, 0};
                }
                a0 = __builtin_bit_cast(bf16x8, g0);
                a1 = __builtin_bit_cast(bf16x8, g1);
            };
            auto finish = [&](const f32x16& acc, int sy, int sx, bool pv) {
                // stem pixels outside the image are the stride-2 conv's zero padding (only row/column -1 can occur)
                const bool outside = (oy0 == 0 && sy == 0) || (ox0 == 0 && sx == 0);
                if (pv) {
                    char* dst = stem + (((sx & 1) * SH + sy) * SWH + (sx >> 1)) * 16 + fh * 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {          // channels 8q + 4fh .. +3 -> chunk q, bytes fh*8..
                        const f32x4 bq = b0p[2 * q];
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (__bf16)(outside ? 0.f : act<ACT>(acc[4 * q + i] + bq[i]));
                        *(bf16x4*)(dst + q * 2 * PLANE) = o;
                    }
                }
            };
            static_assert(NG > NW && NG <= 3 * NW, "every wave has two groups, some a third");
            bf16x8 a0, a1, c0, c1;
            int sya, sxa, syc, sxc;
            bool pva, pvc;
            gather(wave, a0, a1, sya, sxa, pva);
            gather(wave + NW, c0, c1, syc, sxc, pvc);
            f32x16 acca, accc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acca[r] = 0.f; accc[r] = 0.f; }
            acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, a0, acca, 0, 0, 0);     // rows = channels, cols = pixels
            accc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, c0, accc, 0, 0, 0);
            acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, a1, acca, 0, 0, 0);
            accc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, c1, accc, 0, 0, 0);
            finish(acca, sya, sxa, pva);
            finish(accc, syc, sxc, pvc);
            if (wave + 2 * NW < NG) {                            // wave-uniform
                gather(wave + 2 * NW, a0, a1, sya, sxa, pva);
#pragma unroll
                for (int r = 0; r < 16; ++r) acca[r] = 0.f;
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, a0, acca, 0, 0, 0);
                acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, a1, acca, 0, 0, 0);
                finish(acca, sya, sxa, pva);
            }
        }
        FSTAMP(1);                                     // stem groups of this wave
        __syncthreads();
        FSTAMP(2);                                     // barrier after the stem

        // ---- stride-2 conv out of the stem patch: A = weights (32 couts of this wave), B = 32 pixels ----------
        {
            const int ty = wm * 2 + (fr >> 4), tx = fr & 15;
            const char* wrow = w1s + (wn * 32 + fr) * W1_ROW + fh * 16;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // fragments of tap t+1 are read while the two MFMAs of tap t run (static register sets)
            bf16x8 af[2][2], bfr[2][2];
            auto read_tap = [&](int t, int set) {
                const int ky = t / 3, kx = t % 3;
                // pixel (2ty + ky, 2tx + kx): parity kx & 1, x/2 = tx + (kx >> 1)
                const char* px = stem + (((kx & 1) * SH + 2 * ty + ky) * SWH + tx + (kx >> 1)) * 16;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {       // channel halves: chunk = 2 hh + fh
                    bfr[set][hh] = *(const bf16x8*)(px + (2 * hh + fh) * 2 * PLANE);
                    af[set][hh] = *(const bf16x8*)(wrow + (t * 2 + hh) * 32);
                }
            };
            read_tap(0, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) read_tap(t + 1, (t + 1) & 1);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][hh], bfr[t & 1][hh], acc, 0, 0, 0);
            }
            // bias + leaky -> staging [pixel][64 ch] bf16 (row stride 144 B) in the stem patch's place: every wave has read its
            // last stem fragment before anybody overwrites it
            bf16x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = b1p[2 * q];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[q][i] = (__bf16)act<ACT>(acc[4 * q + i] + bq[i]);
            }
            FSTAMP(3);                                 // conv MFMAs + activation
            __syncthreads();
            FSTAMP(4);                                 // barrier before staging
            char* dst = stage + (wm * 32 + fr) * ST_ROW + (wn * 32 + 4 * fh) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(bf16x4*)(dst + q * 16) = o[q];
        }
        __syncthreads();                               // staging complete; the image patch is free again
        FSTAMP(5);                                     // staging writes + barrier

        if (next < p.ntiles) store_image_patch(noy0, nox0);
        // ---- stores: 8 x 16 B per pixel, pixels of a tile row contiguous in memory ----------------------------
#pragma unroll
        for (int it = 0; it < (TH * 16 * 8) / NT; ++it) {
            const int idx = tid + it * NT;
            const int pix = idx >> 3, ch = idx & 7;
            const int py = pix >> 4, pxx = pix & 15;
            const u32x4 v = *(const u32x4*)(stage + pix * ST_ROW + ch * 16);
            *(u32x4*)(p.out + ((((long long)b * Ho + oy0 + py) * Wo + ox0 + pxx) * p.out_ld + p.out_coff) * 2 + ch * 16) = v;
        }
        FSTAMP(6);                                     // image patch -> LDS (waits for its loads), output stores
        __syncthreads();                               // next image patch visible; staging (= stem patch) free
        FSTAMP(7);                                     // last barrier
    }
#ifdef YOLO_CONV_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) g_front_stamps[i] = fst_[i];
#endif
}

#ifdef YOLO_CONV_STAMPS
extern "C" int yolo_debug_read_front_stamps(unsigned long long* host) {
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_front_stamps)) != hipSuccess) return YOLO_ERR_HIP;
    return hipMemcpy(host, dptr, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost) == hipSuccess ? YOLO_OK : YOLO_ERR_HIP;
}
#endif

// ---- stem conv 3 -> 32 + BN + leaky + 2x2 max-pool (YOLOv2's first conv + pool, yolo_v2.py:34-35; bf16) out of an LDS image patch ----
// The stem phase of front_kernel with the pool taken in registers.  One workgroup of 8 waves walks over tiles of 8 x 16 POOLED
// pixels = 16 x 32 stem pixels = 16 groups of 32: wave w computes stem rows 2w and 2w + 1 of the tile (two independent gather ->
// MFMA -> activation chains), takes the maximum of the two rows in registers and of horizontally adjacent pixels through a lane
// exchange, and the even lanes store 4 x 4 channels of one pooled pixel each.  The pool sees the bf16-rounded activations, like the
// separate max-pool kernel (bit-identical).  LDS: the 18 x 34 x 3 image patch in bf16 (3.9 KB) + 32 biases; no staging.
struct StemPoolK {
    const float* img;      // [B,H,W,3] f32
    const u32x4* w0;       // stem weights [32 cout][32 k] bf16, k = (ky*3+kx)*3 + c, BN folded
    const float* b0;       // [32]
    char* out;             // [B,H/2,W/2,out_ld] bf16 (+ out_coff channels)
    int B, H, W, out_ld, out_coff;
    int tiles_x, tiles_y, ntiles;
};

template <int ACT>
__global__ __launch_bounds__(512, 4) void stem_pool_kernel(StemPoolK p) {
    constexpr int NT = 512;
    constexpr int IH = 18, IROW = 102, IWP = 108;      // image patch: 18 x 34 pixels x 3 values, rows padded to 108
    constexpr int NE = (IH * IROW + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned short img[IH * IWP + 8];
    __shared__ __attribute__((aligned(16))) float bias[32];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int Ho = p.H >> 1, Wo = p.W >> 1;

    bf16x8 wf0, wf1;                                   // A[row = cout fr][slot (m, fh, j)] = w0[cout][tap] (common.h stem_slot_tap)
    {
        const unsigned short* wh = (const unsigned short*)p.w0;
        u16x8 h0, h1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t0 = stem_slot_tap(0, fh, j), t1 = stem_slot_tap(1, fh, j);
            h0[j] = t0 >= 0 ? wh[fr * 32 + t0] : (unsigned short)0;
            h1[j] = t1 >= 0 ? wh[fr * 32 + t1] : (unsigned short)0;
        }
        wf0 = __builtin_bit_cast(bf16x8, h0);
        wf1 = __builtin_bit_cast(bf16x8, h1);
    }
    if (tid < 32) bias[tid] = p.b0[tid];
    const f32x4* const b0p = (const f32x4*)(bias + 4 * fh);
    struct __attribute__((packed, aligned(2))) Run8 { u16x8 v; };

    const int per_img = p.tiles_x * p.tiles_y;
    const float inv_per_img = 1.0f / (float)per_img, inv_tx = 1.0f / (float)p.tiles_x;
    auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {     // tile < 2^24: float quotient + fix-up
        b = (int)((float)tile * inv_per_img);
        int r = tile - b * per_img;
        if (r < 0) { --b; r += per_img; } else if (r >= per_img) { ++b; r -= per_img; }
        int ty = (int)((float)r * inv_tx);
        int tx = r - ty * p.tiles_x;
        if (tx < 0) { --ty; tx += p.tiles_x; } else if (tx >= p.tiles_x) { ++ty; tx -= p.tiles_x; }
        oy0 = ty * 8;
        ox0 = tx * 16;
    };
    float pre[NE];
    int lidx[NE], goff[NE], ipos[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * NT;
        const bool v = idx < IH * IROW;
        const int iy = v ? idx / IROW : 0;
        const int rx = v ? idx - iy * IROW : 0;
        lidx[e] = v ? iy * IWP + rx : -1;
        goff[e] = iy * p.W * 3 + rx;
        ipos[e] = iy | ((rx / 3) << 8) | ((rx % 3) << 16);
    }
    auto load_image_patch = [&](int b, int oy0, int ox0) {      // global -> registers
        const int gy0 = 2 * oy0 - 1, gx0 = 2 * ox0 - 1;
        const bool interior = gy0 >= 0 && gy0 + IH <= p.H && gx0 >= 0 && gx0 + 34 <= p.W;      // wave-uniform
        if (interior) {
            const float* org = p.img + (((long long)b * p.H + gy0) * p.W + gx0) * 3;
#pragma unroll
            for (int e = 0; e < NE; ++e) pre[e] = org[lidx[e] >= 0 ? goff[e] : 0];
        } else {                                                  // out-of-image elements read a clamped pixel
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                int gy = gy0 + (ipos[e] & 255), gx = gx0 + ((ipos[e] >> 8) & 255);
                gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
                pre[e] = p.img[(((long long)b * p.H + gy) * p.W + gx) * 3 + (ipos[e] >> 16)];
            }
        }
    };
    auto store_image_patch = [&](int oy0, int ox0) {             // registers -> LDS, zeros outside the image (SAME padding)
        const int gy0 = 2 * oy0 - 1, gx0 = 2 * ox0 - 1;
        const bool interior = gy0 >= 0 && gy0 + IH <= p.H && gx0 >= 0 && gx0 + 34 <= p.W;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            bool inside = true;
            if (!interior) {
                const int gy = gy0 + (ipos[e] & 255), gx = gx0 + ((ipos[e] >> 8) & 255);
                inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            }
            if (lidx[e] >= 0) img[lidx[e]] = __builtin_bit_cast(unsigned short, (__bf16)(inside ? pre[e] : 0.f));
        }
    };

    int tile = blockIdx.x;
    int nb = 0, noy0 = 0, nox0 = 0;
    if (tile < p.ntiles) {
        tile_origin(tile, nb, noy0, nox0);
        load_image_patch(nb, noy0, nox0);
        store_image_patch(noy0, nox0);
    }
    __syncthreads();

    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = nb, oy0 = noy0, ox0 = nox0;
        const int next = tile + gridDim.x;
        if (next < p.ntiles) {
            tile_origin(next, nb, noy0, nox0);
            load_image_patch(nb, noy0, nox0);          // in flight during the compute phase
        }
        // stem rows 2 wave and 2 wave + 1 of the tile, pixel fr of the row: window origin = image patch (row, 3 fr)
        float va[16];
        {
            auto gather = [&](int sy, bf16x8& a0, bf16x8& a1) {
                const unsigned short* base = img + sy * IWP + fr * 3;
                const u16x8 g0 = ((const Run8*)(base + (fh ? 2 * IWP : 0)))->v;
                u16x8 g1;
                if (fh == 0) {
                    g1 = ((const Run8*)(base + IWP))->v;
                } else {
                    g1 = u16x8{base[8], base[IWP + 8], base[2 * IWP + 8], 0, 0, 0, 0, 0};
                }
                a0 = __builtin_bit_cast(bf16x8, g0);
                a1 = __builtin_bit_cast(bf16x8, g1);
            };
            bf16x8 a0, a1, c0, c1;
            gather(2 * wave, a0, a1);
            gather(2 * wave + 1, c0, c1);
            f32x16 acca, accc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acca[r] = 0.f; accc[r] = 0.f; }
            acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, a0, acca, 0, 0, 0);     // rows = channels, cols = pixels
            accc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, c0, accc, 0, 0, 0);
            acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, a1, acca, 0, 0, 0);
            accc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, c1, accc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {              // register 4 q + i = channel 8 q + 4 fh + i
                const f32x4 bq = b0p[2 * q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x0 = (float)(__bf16)act<ACT>(acca[4 * q + i] + bq[i]);
                    const float x1 = (float)(__bf16)act<ACT>(accc[4 * q + i] + bq[i]);
                    va[4 * q + i] = __builtin_fmaxf(x0, x1);                            // the two rows of the window
                }
            }
        }
        // the two columns of the window: lanes fr and fr ^ 1
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float o = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, va[r]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            va[r] = __builtin_fmaxf(va[r], o);
        }
        if ((fr & 1) == 0) {
            char* op = p.out + ((((long long)b * Ho + oy0 + wave) * Wo + ox0 + (fr >> 1)) * p.out_ld + p.out_coff + 4 * fh) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)va[4 * q + i];
                *(bf16x4*)(op + q * 16) = o;
            }
        }
        __syncthreads();                               // every wave has read the image patch
        if (next < p.ntiles) store_image_patch(noy0, nox0);
        __syncthreads();                               // next image patch visible
    }
}

// images [B,H,W,3] f32 -> out [B,H/2,W/2,32] bf16 = pooling_layer(conv_layer(x, 32, 3)) (yolo_v2.py:34-35).  c0: the conv's launch
// description with `out` / `out_ld` / `out_coff` describing the POOLED tensor.
int launch_stem_pool(const ConvLaunch& c0, hipStream_t stream) {
    YOLO_REQUIRE(c0.Cin == 3 && c0.Cout == 32 && c0.ksize == 3 && c0.stride == 1 && c0.in_ld == 3 && c0.in_coff == 0 &&
                 (c0.leaky == 1 || c0.leaky == 2), YOLO_ERR_UNSUPPORTED, "stem + pool: only 3->32 (3x3 s1) is implemented");
    YOLO_REQUIRE(c0.H % 16 == 0 && c0.W % 32 == 0 && c0.out_ld % 4 == 0 && c0.out_coff % 4 == 0, YOLO_ERR_SHAPE,
                 "stem + pool: H must be a multiple of 16, W of 32");
    StemPoolK k;
    k.img = (const float*)c0.in; k.w0 = (const u32x4*)c0.w; k.b0 = c0.bias; k.out = (char*)c0.out;
    k.B = c0.B; k.H = c0.H; k.W = c0.W; k.out_ld = c0.out_ld; k.out_coff = c0.out_coff;
    k.tiles_x = (c0.W / 2) / 16; k.tiles_y = (c0.H / 2) / 8;
    const long long nt = (long long)k.B * k.tiles_x * k.tiles_y;
    YOLO_REQUIRE(nt < (1ll << 24), YOLO_ERR_SHAPE, "stem + pool: too many tiles");
    k.ntiles = (int)nt;
    int grid = 512;                                    // two workgroups of eight waves per CU (<= 128 VGPRs)
    if (grid > k.ntiles) grid = k.ntiles;
    note_kernel("yolo::stem_pool_kernel<%d>", c0.leaky);
    if (c0.leaky == 1) hipLaunchKernelGGL(stem_pool_kernel<1>, dim3(grid), dim3(512), 0, stream, k);
    else hipLaunchKernelGGL(stem_pool_kernel<2>, dim3(grid), dim3(512), 0, stream, k);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

template <int TH>
static int launch_front_th(const FrontK& k, int grid, int act_mode, hipStream_t stream) {
    static std::atomic<bool> attr_done[16];   // zero-initialised; setting the attribute twice is harmless, the flag only must not tear
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)front_kernel<TH, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, front_lds_bytes<TH>()));
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)front_kernel<TH, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, front_lds_bytes<TH>()));
        if (dev < 16) attr_done[dev] = true;
    }
    note_kernel("yolo::front_kernel<%d, %d>", TH, act_mode);
    if (act_mode == 1) hipLaunchKernelGGL((front_kernel<TH, 1>), dim3(grid), dim3(64 * TH), front_lds_bytes<TH>(), stream, k);
    else hipLaunchKernelGGL((front_kernel<TH, 2>), dim3(grid), dim3(64 * TH), front_lds_bytes<TH>(), stream, k);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// images [B,H,W,3] f32 -> out [B,H/2,W/2,64] bf16 = conv2d(conv2d(x, 32, 3), 64, 3, strides=2) of darknet53_body.
// c0 / c1: the two layers' launch descriptions (packed weights, biases, activation); only c1's output geometry is used.
int launch_front(const ConvLaunch& c0, const ConvLaunch& c1, hipStream_t stream) {
    YOLO_REQUIRE(c0.Cin == 3 && c0.Cout == 32 && c0.ksize == 3 && c0.stride == 1 && c0.in_ld == 3 && c0.in_coff == 0 &&
                 c1.Cin == 32 && c1.Cout == 64 && c1.ksize == 3 && c1.stride == 2 && !c1.res && !c1.upsample && !c1.out_f32 &&
                 (c0.leaky == 1 || c0.leaky == 2) && c1.leaky == c0.leaky,
                 YOLO_ERR_UNSUPPORTED, "front: only 3->32 (3x3 s1) followed by 32->64 (3x3 s2) is implemented");
    YOLO_REQUIRE(c0.H % 32 == 0 && c0.W % 32 == 0 && c1.out_ld % 8 == 0 && c1.out_coff % 8 == 0 && c1.kpad_bytes >= 576,
                 YOLO_ERR_SHAPE, "front: H, W must be multiples of 32");
    static const int th_env = [] { const char* e = getenv("YOLO_FRONT_TH"); return e ? atoi(e) : 0; }();
    static const int bpc_env = [] { const char* e = getenv("YOLO_FRONT_BPC"); return e ? atoi(e) : 0; }();
    const int th = th_env == 4 ? 4 : 8;
    FrontK k;
    k.img = (const float*)c0.in; k.w0 = (const u32x4*)c0.w; k.b0 = c0.bias; k.w1 = (const char*)c1.w; k.b1 = c1.bias;
    k.out = (char*)c1.out;
    k.B = c0.B; k.H = c0.H; k.W = c0.W; k.w1_row = c1.kpad_bytes; k.out_ld = c1.out_ld; k.out_coff = c1.out_coff;
    k.act0 = c0.leaky; k.act1 = c1.leaky;
    k.tiles_x = (c0.W / 2) / 16; k.tiles_y = (c0.H / 2) / th;
    const long long nt = (long long)k.B * k.tiles_x * k.tiles_y;
    YOLO_REQUIRE(nt < (1ll << 31), YOLO_ERR_SHAPE, "front: too many tiles");
    k.ntiles = (int)nt;
    // persistent workgroups: as many as fit a CU's LDS (TH = 8: one of 8 waves, TH = 4: two of 4 waves), each walks
    // ntiles / grid tiles
    int grid = 256 * (bpc_env > 0 ? bpc_env : 2);
    if (grid > k.ntiles) grid = k.ntiles;
    return th == 4 ? launch_front_th<4>(k, grid, c0.leaky, stream) : launch_front_th<8>(k, grid, c0.leaky, stream);
}

}  // namespace yolo
