// First residual block of Darknet-53 fused (bf16 path): res_block(net, 32) of darknet53_body (reference
// yolo_v3/utils/layer_utils.py:24-29,33) = conv 64->32 1x1 + BN + leaky, conv 32->64 3x3 + BN + leaky, + shortcut,
// on the 208x208 map.  As two kernels the block moves 354 + 177 (written) + 177 (read again) + 354 (shortcut) +
// 354 MB at batch 64 and both kernels are bound by HBM; fused, the 32-channel intermediate never leaves the CU and the
// shortcut is the centre of the input patch already in LDS: 354 MB in, 354 MB out.
//
// One workgroup of 512 threads walks over output tiles of 8 x 16 pixels (same skeleton as front.hip):
//   1. input patch 10 x 18 pixels x 64 channels (128 B rows, 16-B chunks XOR-swizzled) -> LDS, zeros outside the
//      map; the loads of the NEXT tile's patch are issued before the compute phases and written to LDS after them.
//   2. 1x1 conv on MFMA for the 180 patch pixels (6 groups of 32; weights = A operand, so a lane ends up with 4 x 4
//      consecutive channels of ONE pixel), bias + leaky, zero where the pixel lies outside the map (the 3x3 conv's
//      SAME padding), bf16 -> LDS in planes [channel chunk of 8][pixel][16 B] (conflict-free fragment reads).
//   3. 3x3 conv as an implicit GEMM out of LDS: per wave 32 pixels x 32 channels, 18 k-slices; weights [64][288]
//      bf16 staged once per workgroup.  bias + leaky, + shortcut (read from the input patch and replaced in place by
//      the result) -> 16-byte stores, 128 contiguous bytes per pixel.
// Round 3: no staging tile (in-place result), biases in LDS: 93 -> 75 KB of LDS and 156 -> <= 128 VGPRs, two workgroups
// (32 waves) per CU.
// Same products, accumulation order and roundings as the two separate kernels: the results are bit-identical.
#include <cstdlib>
#include "common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ResK {
    const char* x;         // [B,H,W,x_ld] bf16, 64 channels from x_coff
    const char* w1;        // 1x1 weights [>=32 cout][w1_row bytes] bf16 (64 k)
    const float* b1;       // [32]
    const char* w3;        // 3x3 weights [>=64 cout][w3_row bytes] bf16, k = (ky*3+kx)*32 + ci
    const float* b3;       // [64]
    char* out;             // [B,H,W,out_ld] bf16 (+ out_coff)
    int B, H, W, x_ld, x_coff, w1_row, w3_row, out_ld, out_coff;
    int tiles_x, tiles_y, ntiles;
};

constexpr int kResLds = 192 * 128 + 4 * 192 * 16 + 64 * 592 + 96 * 4 + 32 * 144;

template <int ACT>
__device__ __forceinline__ float res_act(float v) {
    if constexpr (ACT == 1) return leaky_v3(v);                                   // model.py:8-12
    else return __builtin_fmaxf(0.1f * v, v);
}

template <int ACT>
__global__ __launch_bounds__(512, 4) void res_block_kernel(ResK p) {
    constexpr int NT = 512, TH = 8, PW = 18, NP = 180, NPP = 192;
    constexpr int XS_BYTES = NPP * 128, YPLANE = NPP * 16, YS_BYTES = 4 * YPLANE;
    constexpr int W3_ROW = 592, W3_BYTES = 64 * W3_ROW;
    constexpr int NE = (NP * 8 + NT - 1) / NT;         // 16-byte chunks of the input patch per thread

    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* const xs = lds;
    char* const ys = xs + XS_BYTES;
    char* const w3s = ys + YS_BYTES;
    float* const b1s = (float*)(w3s + W3_BYTES);       // [32] + [64] biases (registers are needed elsewhere: <= 128 VGPRs)
    float* const b3s = b1s + 32;
    char* const w1s = (char*)(b3s + 64);               // 1x1 weights [32 cout][64 k] bf16, rows padded to 144 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- once per workgroup ---------------------------------------------------------------------------------
    if (tid < 32) b1s[tid] = p.b1[tid];
    if (tid < 64) b3s[tid] = p.b3[tid];
    for (int i = tid; i < 64 * 36; i += NT) {          // 3x3 weights: 36 x 16 B per row of 288 bf16
        const int row = i / 36, ch = i - row * 36;
        *(u32x4*)(w3s + row * W3_ROW + ch * 16) = *(const u32x4*)(p.w3 + (long long)row * p.w3_row + ch * 16);
    }
    for (int i = tid; i < (NPP - NP) * 8; i += NT)     // pad pixels of the patch: never loaded, keep them finite
        *(u32x4*)(xs + NP * 128 + i * 16) = u32x4{0u, 0u, 0u, 0u};
    if (tid < 32 * 8)                                  // 1x1 weights: A[row = cout fr][k = 16 q + 8 fh + j], read per tile from LDS
        *(u32x4*)(w1s + (tid >> 3) * 144 + (tid & 7) * 16) = *(const u32x4*)(p.w1 + (long long)(tid >> 3) * p.w1_row + (tid & 7) * 16);

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
    // input patch chunk e of this thread: patch pixel (py, px), 16-B chunk ch; tile-invariant LDS address
    u32x4 pre[NE];
    int ppos[NE], ldst[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * NT;
        const bool v = idx < NP * 8;
        const int pix = v ? idx >> 3 : 0, ch = idx & 7;
        const int py = pix / PW, px = pix - py * PW;
        ppos[e] = v ? (py | (px << 8) | (ch << 16)) : -1;
        ldst[e] = pix * 128 + ((ch ^ ((pix >> 1) & 7)) << 4);
    }
    auto load_patch = [&](int b, int oy0, int ox0) {   // global -> registers (out-of-map chunks read a clamped pixel)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int pp = ppos[e] < 0 ? 0 : ppos[e];
            int gy = oy0 - 1 + (pp & 255), gx = ox0 - 1 + ((pp >> 8) & 255);
            gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
            pre[e] = *(const u32x4*)(p.x + ((((long long)b * p.H + gy) * p.W + gx) * p.x_ld + p.x_coff) * 2 + (pp >> 16) * 16);
        }
    };
    auto store_patch = [&](int oy0, int ox0) {         // registers -> LDS, zeros outside the map
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (ppos[e] < 0) continue;
            const int gy = oy0 - 1 + (ppos[e] & 255), gx = ox0 - 1 + ((ppos[e] >> 8) & 255);
            const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            *(u32x4*)(xs + ldst[e]) = inside ? pre[e] : u32x4{0u, 0u, 0u, 0u};
        }
    };

    int tile = blockIdx.x;
    int nb = 0, noy0 = 0, nox0 = 0;
    if (tile < p.ntiles) {
        tile_origin(tile, nb, noy0, nox0);
        load_patch(nb, noy0, nox0);
        store_patch(noy0, nox0);
    }
    __syncthreads();

    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = nb, oy0 = noy0, ox0 = nox0;
        const int next = tile + gridDim.x;
        if (next < p.ntiles) {
            tile_origin(next, nb, noy0, nox0);
            load_patch(nb, noy0, nox0);                // in flight during the two compute phases
        }

        // ---- 1x1 conv: waves 0..5 take one group of 32 patch pixels each ------------------------------------
        if (wave < NPP / 32) {
            const int pix = wave * 32 + fr;
            const char* prow = xs + pix * 128;
            const int sw = (pix >> 1) & 7;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 xf = *(const bf16x8*)(prow + (((2 * q + fh) ^ sw) << 4));   // B[k = 16 q + 8 fh + j][pixel]
                const bf16x8 wf = *(const bf16x8*)(w1s + fr * 144 + q * 32 + fh * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
            }
            const int py = pix / PW, px = pix - py * PW;
            const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
            const bool outside = gy < 0 || gy >= p.H || gx < 0 || gx >= p.W;             // the 3x3 conv's zero padding
            char* dst = ys + pix * 16 + fh * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {              // channels 8q + 4fh .. +3 -> plane q
                const f32x4 bq = *(const f32x4*)(b1s + 8 * q + 4 * fh);
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)(outside ? 0.f : res_act<ACT>(acc[4 * q + i] + bq[i]));
                *(bf16x4*)(dst + q * YPLANE) = o;
            }
        }
        __syncthreads();

        // ---- 3x3 conv out of the intermediate patch: A = weights (32 couts of this wave), B = 32 pixels ---------
        {
            const int ty = wm * 2 + (fr >> 4), tx = fr & 15;
            const char* wrow = w3s + (wn * 32 + fr) * W3_ROW + fh * 16;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            bf16x8 af[2][2], bfr[2][2];
            auto read_tap = [&](int t, int set) {
                const int ky = t / 3, kx = t % 3;
                const char* px = ys + ((ty + ky) * PW + tx + kx) * 16;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {       // channel halves: plane = 2 hh + fh
                    bfr[set][hh] = *(const bf16x8*)(px + (2 * hh + fh) * YPLANE);
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
            // bias + leaky, + shortcut (centre pixel of the input patch); the result replaces the shortcut IN PLACE: the
            // 8 bytes a lane reads are the 8 bytes it writes, no other lane touches them, and the patch row of the centre
            // pixel (64 channels x 2 B) is exactly an output row - no separate staging tile (18 KB), no barrier in between
            const int pc = (ty + 1) * PW + tx + 1;
            char* srow = xs + pc * 128 + fh * 8;
            const int sw = (pc >> 1) & 7;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                char* const sp = srow + (((wn * 4 + q) ^ sw) << 4);                       // channels wn*32 + 8q + 4fh ..
                const bf16x4 sc = *(const bf16x4*)sp;
                const f32x4 bq = *(const f32x4*)(b3s + wn * 32 + 8 * q + 4 * fh);
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)(res_act<ACT>(acc[4 * q + i] + bq[i]) + (float)sc[i]);
                *(bf16x4*)sp = o;
            }
        }
        __syncthreads();                               // the output rows are complete

#pragma unroll
        for (int it = 0; it < (TH * 16 * 8) / NT; ++it) {
            const int idx = tid + it * NT;
            const int pix = idx >> 3, ch = idx & 7;
            const int py = pix >> 4, pxx = pix & 15;
            const int pc = (py + 1) * PW + pxx + 1;
            const u32x4 v = *(const u32x4*)(xs + pc * 128 + ((ch ^ ((pc >> 1) & 7)) << 4));
            *(u32x4*)(p.out + ((((long long)b * p.H + oy0 + py) * p.W + ox0 + pxx) * p.out_ld + p.out_coff) * 2 + ch * 16) = v;
        }
        __syncthreads();                               // every output row has been read: the patch may be replaced
        if (next < p.ntiles) store_patch(noy0, nox0);
        __syncthreads();                               // next input patch visible
    }
}

// ---- conv 32 -> 64 3x3 + BN + leaky + 2x2 max-pool fused (YOLOv2's second conv + pool, yolo_v2.py:36-37; bf16) -----------
// The 3x3 phase of res_block_kernel fed straight from global memory: the 10 x 18 x 32-channel input patch of an 8 x 16
// output tile goes to LDS in the plane layout [channel chunk of 8][pixel][16 B], the weights [64][288] are staged once per
// workgroup, each wave computes 32 pixels x 32 channels (18 MFMAs), and the epilogue writes the maximum of every 2 x 2
// window of the bf16-rounded activations (= the pool of the rounded tensor, rounding is monotonic): the 354 MB
// full-resolution output at batch 64 is never written or read.  67 KB of LDS: two workgroups per CU.
struct C32K {
    const char* x;         // [B,H,W,x_ld] bf16, 32 channels from x_coff
    const char* w3;        // [>=64 cout][w3_row bytes] bf16, k = (ky*3+kx)*32 + ci
    const float* b3;       // [64]
    char* out;             // [B,H/2,W/2,out_ld] bf16 (+ out_coff)
    int B, H, W, x_ld, x_coff, w3_row, out_ld, out_coff;
    int tiles_x, tiles_y, ntiles;
};
constexpr int kC32Lds = 4 * 192 * 16 + 64 * 592 + 128 * 144;

template <int ACT>
__global__ __launch_bounds__(512, 4) void conv3x3_c32_pool_kernel(C32K p) {
    constexpr int NT = 512, TH = 8, PW = 18, NP = 180, NPP = 192;
    constexpr int YPLANE = NPP * 16, YS_BYTES = 4 * YPLANE;
    constexpr int W3_ROW = 592, W3_BYTES = 64 * W3_ROW, ST_ROW = 144;
    constexpr int NE = (NP * 4 + NT - 1) / NT;         // 16-byte chunks of the input patch per thread

    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* const ys = lds;
    char* const w3s = ys + YS_BYTES;
    char* const stage = w3s + W3_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    for (int i = tid; i < 64 * 36; i += NT) {
        const int row = i / 36, ch = i - row * 36;
        *(u32x4*)(w3s + row * W3_ROW + ch * 16) = *(const u32x4*)(p.w3 + (long long)row * p.w3_row + ch * 16);
    }
    for (int i = tid; i < (NPP - NP) * 4; i += NT)     // pad pixels of the planes: never loaded, keep them finite
        *(u32x4*)(ys + (i & 3) * YPLANE + (NP + (i >> 2)) * 16) = u32x4{0u, 0u, 0u, 0u};
    float b3v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b3v[r] = p.b3[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];

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
    u32x4 pre[NE];
    int ppos[NE], ldst[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = tid + e * NT;
        const bool v = idx < NP * 4;
        const int pix = v ? idx >> 2 : 0, q = idx & 3;
        const int py = pix / PW, px = pix - py * PW;
        ppos[e] = v ? (py | (px << 8) | (q << 16)) : -1;
        ldst[e] = q * YPLANE + pix * 16;
    }
    auto load_patch = [&](int b, int oy0, int ox0) {   // global -> registers (out-of-map chunks read a clamped pixel)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int pp = ppos[e] < 0 ? 0 : ppos[e];
            int gy = oy0 - 1 + (pp & 255), gx = ox0 - 1 + ((pp >> 8) & 255);
            gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
            pre[e] = *(const u32x4*)(p.x + ((((long long)b * p.H + gy) * p.W + gx) * p.x_ld + p.x_coff) * 2 + (pp >> 16) * 16);
        }
    };
    auto store_patch = [&](int oy0, int ox0) {         // registers -> LDS, zeros outside the map (SAME padding)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (ppos[e] < 0) continue;
            const int gy = oy0 - 1 + (ppos[e] & 255), gx = ox0 - 1 + ((ppos[e] >> 8) & 255);
            const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            *(u32x4*)(ys + ldst[e]) = inside ? pre[e] : u32x4{0u, 0u, 0u, 0u};
        }
    };

    int tile = blockIdx.x;
    int nb = 0, noy0 = 0, nox0 = 0;
    if (tile < p.ntiles) {
        tile_origin(tile, nb, noy0, nox0);
        load_patch(nb, noy0, nox0);
        store_patch(noy0, nox0);
    }
    __syncthreads();

    const int Ho = p.H >> 1, Wo = p.W >> 1;
    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = nb, oy0 = noy0, ox0 = nox0;
        const int next = tile + gridDim.x;
        if (next < p.ntiles) {
            tile_origin(next, nb, noy0, nox0);
            load_patch(nb, noy0, nox0);                // in flight during the compute phase
        }
        {
            const int ty = wm * 2 + (fr >> 4), tx = fr & 15;
            const char* wrow = w3s + (wn * 32 + fr) * W3_ROW + fh * 16;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            bf16x8 af[2][2], bfr[2][2];
            auto read_tap = [&](int t, int set) {
                const int ky = t / 3, kx = t % 3;
                const char* px = ys + ((ty + ky) * PW + tx + kx) * 16;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {       // channel halves: plane = 2 hh + fh
                    bfr[set][hh] = *(const bf16x8*)(px + (2 * hh + fh) * YPLANE);
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
            char* dst = stage + (wm * 32 + fr) * ST_ROW + (wn * 32 + 4 * fh) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)res_act<ACT>(acc[4 * q + i] + b3v[4 * q + i]);
                *(bf16x4*)(dst + q * 16) = o;
            }
        }
        __syncthreads();                               // staging complete; the patch is free again

        if (next < p.ntiles) store_patch(noy0, nox0);
        if (tid < 32 * 8) {                            // 4 x 8 pooled pixels x 8 chunks of 8 channels
            const int pp = tid >> 3, ch = tid & 7;
            const int py2 = pp >> 3, px2 = pp & 7;
            const char* s0 = stage + ((2 * py2) * 16 + 2 * px2) * ST_ROW + ch * 16;
            bf16x8 r = *(const bf16x8*)s0;
            const bf16x8 v1 = *(const bf16x8*)(s0 + ST_ROW), v2 = *(const bf16x8*)(s0 + 16 * ST_ROW), v3 = *(const bf16x8*)(s0 + 17 * ST_ROW);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                r[e] = (float)v1[e] > (float)r[e] ? v1[e] : r[e];
                r[e] = (float)v2[e] > (float)r[e] ? v2[e] : r[e];
                r[e] = (float)v3[e] > (float)r[e] ? v3[e] : r[e];
            }
            *(bf16x8*)(p.out + ((((long long)b * Ho + (oy0 >> 1) + py2) * Wo + (ox0 >> 1) + px2) * p.out_ld + p.out_coff) * 2 + ch * 16) = r;
        }
        __syncthreads();                               // next input patch visible; staging free
    }
}

// x [B,H,W,32] bf16 -> out [B,H/2,W/2,64] bf16 = pooling_layer(conv_layer(x, 64, 3)) (yolo_v2.py:36-37).  c3: launch description of
// the conv with `out` / `out_ld` / `out_coff` describing the POOLED tensor.
int launch_conv3x3_c32_pool(const ConvLaunch& c3, hipStream_t stream) {
    YOLO_REQUIRE(c3.Cin == 32 && c3.Cout == 64 && c3.ksize == 3 && c3.stride == 1 && !c3.res && !c3.upsample && !c3.out_f32 &&
                 !c3.reorg && (c3.leaky == 1 || c3.leaky == 2), YOLO_ERR_UNSUPPORTED, "conv + pool: only 32->64 3x3 stride 1 is implemented");
    YOLO_REQUIRE(c3.H % 8 == 0 && c3.W % 16 == 0 && c3.in_ld % 8 == 0 && c3.in_coff % 8 == 0 && c3.out_ld % 8 == 0 &&
                 c3.out_coff % 8 == 0 && c3.kpad_bytes >= 576, YOLO_ERR_SHAPE, "conv + pool: bad geometry");
    C32K k;
    k.x = (const char*)c3.in; k.w3 = (const char*)c3.w; k.b3 = c3.bias; k.out = (char*)c3.out;
    k.B = c3.B; k.H = c3.H; k.W = c3.W; k.x_ld = c3.in_ld; k.x_coff = c3.in_coff; k.w3_row = c3.kpad_bytes;
    k.out_ld = c3.out_ld; k.out_coff = c3.out_coff;
    k.tiles_x = c3.W / 16; k.tiles_y = c3.H / 8;
    const long long nt = (long long)k.B * k.tiles_x * k.tiles_y;
    YOLO_REQUIRE(nt < (1ll << 24), YOLO_ERR_SHAPE, "conv + pool: too many tiles");
    k.ntiles = (int)nt;
    int grid = 512;                                    // two workgroups per CU, each walks ntiles / 512 tiles
    if (grid > k.ntiles) grid = k.ntiles;
    static std::atomic<bool> attr_done[16];
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)conv3x3_c32_pool_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kC32Lds));
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)conv3x3_c32_pool_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kC32Lds));
        if (dev < 16) attr_done[dev] = true;
    }
    note_kernel("yolo::conv3x3_c32_pool_kernel<%d>", c3.leaky);
    if (c3.leaky == 1) hipLaunchKernelGGL(conv3x3_c32_pool_kernel<1>, dim3(grid), dim3(512), kC32Lds, stream, k);
    else hipLaunchKernelGGL(conv3x3_c32_pool_kernel<2>, dim3(grid), dim3(512), kC32Lds, stream, k);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// x [B,H,W,64] bf16 -> out [B,H,W,64] bf16 = res_block(x, 32) (layer_utils.py:24-29).  c1 / c3: launch descriptions of
// the 1x1 and the 3x3 conv (c3.res must be the block input c1.in).
int launch_res_block(const ConvLaunch& c1, const ConvLaunch& c3, hipStream_t stream) {
    YOLO_REQUIRE(c1.Cin == 64 && c1.Cout == 32 && c1.ksize == 1 && c1.stride == 1 && c3.Cin == 32 && c3.Cout == 64 &&
                 c3.ksize == 3 && c3.stride == 1 && c3.res == c1.in && c3.res_ld == c1.in_ld && c3.res_coff == c1.in_coff &&
                 !c3.upsample && !c3.out_f32 && !c3.reorg && (c1.leaky == 1 || c1.leaky == 2) && c3.leaky == c1.leaky,
                 YOLO_ERR_UNSUPPORTED, "res_block: only 64->32 (1x1) + 32->64 (3x3) + shortcut is implemented");
    YOLO_REQUIRE(c1.H % 8 == 0 && c1.W % 16 == 0 && c1.in_ld % 8 == 0 && c1.in_coff % 8 == 0 && c3.out_ld % 8 == 0 &&
                 c3.out_coff % 8 == 0 && c1.kpad_bytes >= 128 && c3.kpad_bytes >= 576, YOLO_ERR_SHAPE, "res_block: bad geometry");
    ResK k;
    k.x = (const char*)c1.in; k.w1 = (const char*)c1.w; k.b1 = c1.bias; k.w3 = (const char*)c3.w; k.b3 = c3.bias;
    k.out = (char*)c3.out;
    k.B = c1.B; k.H = c1.H; k.W = c1.W; k.x_ld = c1.in_ld; k.x_coff = c1.in_coff; k.w1_row = c1.kpad_bytes;
    k.w3_row = c3.kpad_bytes; k.out_ld = c3.out_ld; k.out_coff = c3.out_coff;
    k.tiles_x = c1.W / 16; k.tiles_y = c1.H / 8;
    const long long nt = (long long)k.B * k.tiles_x * k.tiles_y;
    YOLO_REQUIRE(nt < (1ll << 24), YOLO_ERR_SHAPE, "res_block: too many tiles");
    k.ntiles = (int)nt;
    int grid = 512;                                    // two workgroups per CU (75 KB of LDS, <= 128 VGPRs), each walks ntiles / 512 tiles
    if (grid > k.ntiles) grid = k.ntiles;
    static std::atomic<bool> attr_done[16];   // zero-initialised; setting the attribute twice is harmless, the flag only must not tear
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)res_block_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kResLds));
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)res_block_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kResLds));
        if (dev < 16) attr_done[dev] = true;
    }
    note_kernel("yolo::res_block_kernel<%d>", c1.leaky);
    if (c1.leaky == 1) hipLaunchKernelGGL(res_block_kernel<1>, dim3(grid), dim3(512), kResLds, stream, k);
    else hipLaunchKernelGGL(res_block_kernel<2>, dim3(grid), dim3(512), kResLds, stream, k);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

}  // namespace yolo
