// Stem conv (reference yolo_v3/utils/layer_utils.py:31: conv2d(inputs, 32, 3, strides=1) on the RGB
// image) and small layout helpers.  Cin = 3 makes K = 27: far below an MFMA K-step and not 16-byte
// addressable, so this layer is a direct f32 VALU convolution.  It is HBM-bound anyway: per pixel
// it reads 12 B and writes 64 B (bf16) / 128 B (f32) against 1728 FLOP.
// One thread = one output pixel x COUT output channels; the 27 x COUT folded weights are read with
// wave-uniform addresses (scalar loads -> SGPR operands of v_fmac).
#include "common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool BF16, int COUT>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, char* __restrict__ out,
                                                   int B, int H, int W, int out_ld, int out_coff, int leaky) {
    const long long npix = (long long)B * H * W;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= npix) return;
    const int x = (int)(m % W);
    const long long t = m / W;
    const int y = (int)(t % H);
    const long long b = t / H;
    float acc[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) acc[n] = bias[n];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        const bool yok = iy >= 0 && iy < H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = x + kx - 1;
            const bool ok = yok && ix >= 0 && ix < W;
            const float* px = in + ((b * H + iy) * W + ix) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = ok ? px[c] : 0.f;
                const float* wr = w + ((ky * 3 + kx) * 3 + c) * COUT;
#pragma unroll
                for (int n = 0; n < COUT; ++n) acc[n] = __builtin_fmaf(v, wr[n], acc[n]);
            }
        }
    }
    if (leaky == 1) {
#pragma unroll
        for (int n = 0; n < COUT; ++n) acc[n] = leaky_v3(acc[n]);
    } else if (leaky == 2) {
#pragma unroll
        for (int n = 0; n < COUT; ++n) acc[n] = __builtin_fmaxf(0.1f * acc[n], acc[n]);
    }
    if constexpr (BF16) {
        char* op = out + (m * out_ld + out_coff) * 2;
#pragma unroll
        for (int n = 0; n < COUT; n += 8) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[n + e];
            *(bf16x8*)(op + n * 2) = o;
        }
    } else {
        char* op = out + (m * out_ld + out_coff) * 4;
#pragma unroll
        for (int n = 0; n < COUT; n += 4) {
            f32x4 o = {acc[n], acc[n + 1], acc[n + 2], acc[n + 3]};
            *(f32x4*)(op + n * 4) = o;
        }
    }
}

// bf16 path: the same layer on MFMA.  K = 27 is padded to 32: two v_mfma_f32_32x32x16_bf16 per 32
// pixels x 32 channels.  A lane (pixel = lane&31, half = lane>>5) gathers its 16 im2col values
// k = 16m + 8*half + j straight from the f32 image (k = (ky*3+kx)*3 + c; the three taps of one input
// row are 9 contiguous floats), rounds them to bf16 and feeds the MFMA; weights [32 cout][32 k] bf16
// stay in registers.  The 32x32 f32 result is transposed through a wave-private LDS tile so that
// every lane stores 32 contiguous bytes (16 channels of one pixel): 2 KiB contiguous per wave.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ in, const u32x4* __restrict__ w,
                                                        const float* __restrict__ bias, char* __restrict__ out,
                                                        int B, int H, int W, int out_ld, int out_coff, int leaky) {
    __shared__ float tile[4][32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    float* tl = tile[wave];
    // weight fragments: B[slot (m, fh, j)][cout = fr]  <-  w[cout][tap] (common.h stem_slot_tap)
    bf16x8 wf0, wf1;
    {
        const unsigned short* wh = (const unsigned short*)w;
        typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
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
    // per-lane im2col tables (invariant over pixels)
    int koff[2][8];
    unsigned need[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = stem_slot_tap(m, fh, j);
            const int kq = tap < 0 ? 0 : tap;
            const int ky = kq / 9, rem = kq - ky * 9, kx = rem / 3;
            koff[m][j] = (ky - 1) * W * 3 + rem - 3;
            need[m][j] = (ky == 0 ? 1u : 0u) | (ky == 2 ? 2u : 0u) | (kx == 0 ? 4u : 0u) | (kx == 2 ? 8u : 0u) | (tap < 0 ? 16u : 0u);
        }
    const float bv = bias[fr];
    const long long npix = (long long)B * H * W;
    const long long ngroups = (npix + 31) >> 5;
    for (long long grp = (long long)blockIdx.x * 4 + wave; grp < ngroups; grp += (long long)gridDim.x * 4) {
        const long long m = grp * 32 + fr;
        const bool pv = m < npix;
        const int x = (int)(m % W);
        const int y = (int)((m / W) % H);
        const unsigned lack = (y > 0 ? 0u : 1u) | (y < H - 1 ? 0u : 2u) | (x > 0 ? 0u : 4u) | (x < W - 1 ? 0u : 8u) | 16u;
        const float* base = in + m * 3;
        bf16x8 a0, a1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok0 = pv && (need[0][j] & lack) == 0;
            const bool ok1 = pv && (need[1][j] & lack) == 0;
            a0[j] = (__bf16)(ok0 ? base[koff[0][j]] : 0.f);
            a1[j] = (__bf16)(ok1 ? base[koff[1][j]] : 0.f);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf1, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            float v = acc[r] + bv;
            if (leaky == 1) v = leaky_v3(v);
            else if (leaky == 2) v = __builtin_fmaxf(0.1f * v, v);
            tl[row * 33 + fr] = v;
        }
        const int pr = lane >> 1, c0 = (lane & 1) * 16;
        const long long mo = grp * 32 + pr;
        bf16x8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o0[e] = (__bf16)tl[pr * 33 + c0 + e];
            o1[e] = (__bf16)tl[pr * 33 + c0 + 8 + e];
        }
        if (mo < npix) {
            char* op = out + (mo * out_ld + out_coff + c0) * 2;
            *(bf16x8*)op = o0;
            *(bf16x8*)(op + 16) = o1;
        }
    }
}

int launch_stem(int dtype, const ConvLaunch& p, hipStream_t stream) {
    YOLO_REQUIRE(p.Cin == 3 && p.ksize == 3 && p.stride == 1 && p.Cout == 32 && p.in_ld == 3 && p.in_coff == 0,
                 YOLO_ERR_UNSUPPORTED, "stem: only 3x3 s1 3->32 on a dense RGB image is implemented");
    YOLO_REQUIRE(p.out_ld % 8 == 0 && p.out_coff % 8 == 0, YOLO_ERR_SHAPE, "stem: output stride");
    const long long npix = (long long)p.B * p.H * p.W;
    const unsigned grid = (unsigned)((npix + 255) / 256);
    if (dtype == YOLO_DTYPE_BF16) {
        const long long ngroups = (npix + 31) / 32;
        long long g = (ngroups + 3) / 4;
        if (g > 256 * 16) g = 256 * 16;
        note_kernel("yolo::stem_mfma_kernel");
        hipLaunchKernelGGL(stem_mfma_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float*)p.in,
                           (const u32x4*)p.w, p.bias, (char*)p.out, p.B, p.H, p.W, p.out_ld, p.out_coff, p.leaky);
    } else {
        note_kernel("yolo::stem_kernel<false, 32>");
        hipLaunchKernelGGL((stem_kernel<false, 32>), dim3(grid), dim3(256), 0, stream, (const float*)p.in,
                           (const float*)p.w, p.bias, (char*)p.out, p.B, p.H, p.W, p.out_ld, p.out_coff, p.leaky);
    }
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

static unsigned cast_grid_fwd(size_t n) { size_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }
// ---- pooling_layer (reference yolo_v2/yolo/yolo_v2.py:91-94): max_pool 2x2 stride 2 SAME on even sizes.
// HBM-bound: 16 bytes per lane, channel-contiguous.
template <bool BF16>
__global__ void maxpool2x2_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int B, int H, int W, int C16) {
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t total = (size_t)B * Ho * Wo * C16;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < total; i += stride) {
        const int c = (int)(i % C16);
        size_t t = i / C16;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        const size_t b = t / Ho;
        const size_t base = ((b * H + 2 * y) * W + 2 * x) * C16 + c;
        const u32x4 q[4] = {in[base], in[base + C16], in[base + (size_t)W * C16], in[base + (size_t)W * C16 + C16]};
        u32x4 o;
        if constexpr (BF16) {
            bf16x8 r = __builtin_bit_cast(bf16x8, q[0]);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const bf16x8 v = __builtin_bit_cast(bf16x8, q[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (float)v[e] > (float)r[e] ? v[e] : r[e];
            }
            o = __builtin_bit_cast(u32x4, r);
        } else {
            f32x4 r = __builtin_bit_cast(f32x4, q[0]);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const f32x4 v = __builtin_bit_cast(f32x4, q[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = v[e] > r[e] ? v[e] : r[e];
            }
            o = __builtin_bit_cast(u32x4, r);
        }
        out[i] = o;
    }
}

int launch_maxpool2x2(int dtype, const void* in, int B, int H, int W, int C, void* out, hipStream_t stream) {
    const int per16 = dtype == YOLO_DTYPE_BF16 ? 8 : 4;
    YOLO_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % per16 == 0, YOLO_ERR_SHAPE,
                 "maxpool2x2: need even H, W and C %% %d == 0", per16);
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / per16);
    if (dtype == YOLO_DTYPE_BF16)
        hipLaunchKernelGGL(maxpool2x2_kernel<true>, dim3(cast_grid_fwd(total)), dim3(256), 0, stream, (const u32x4*)in, (u32x4*)out, B, H, W, C / per16);
    else
        hipLaunchKernelGGL(maxpool2x2_kernel<false>, dim3(cast_grid_fwd(total)), dim3(256), 0, stream, (const u32x4*)in, (u32x4*)out, B, H, W, C / per16);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// ---- f32 <-> bf16 casts used by the single-layer entry point (yolo_conv_forward) ---------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        if (i + 4 <= n) {
            f32x4 v = *(const f32x4*)(in + i);
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *(bf16x4*)(out + i) = o;
        } else {
            for (size_t j = i; j < n; ++j) out[j] = (__bf16)in[j];
        }
    }
}

__global__ void cast_bf16_f32_kernel(const __bf16* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = (float)in[i];
}

static unsigned cast_grid(size_t n, int per_thread) {
    size_t g = (n + 256ull * per_thread - 1) / (256ull * per_thread);
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int launch_cast_f32_to_bf16(const float* in, void* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cast_grid(n, 4)), dim3(256), 0, stream, in, (__bf16*)out, n);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

int launch_cast_bf16_to_f32(const void* in, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(cast_grid(n, 1)), dim3(256), 0, stream, (const __bf16*)in, out, n);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// ---- upsample_layer (layer_utils.py:67-72), exact-2x case: src = dst // 2 ----------------------
__global__ void upsample2x_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int B, int H, int W, int C4) {
    const size_t total = (size_t)B * 2 * H * 2 * W * C4;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < total; i += stride) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int x = (int)(t % (2 * W));
        t /= 2 * W;
        const int y = (int)(t % (2 * H));
        const size_t b = t / (2 * H);
        out[i] = in[((b * H + (y >> 1)) * W + (x >> 1)) * C4 + c];
    }
}

// ---- upsample_layer for any output size: tf.image.resize_nearest_neighbor(align_corners=True) as TF 1.x computes it:
// scale = (in - 1) / (out - 1) (0 when out == 1), src = min(round(dst * scale), in - 1), in float32 with separately rounded
// multiply and add.
template <typename T>
__global__ void upsample_nearest_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int CV, int Ho, int Wo,
                                        float sy, float sx) {
    const size_t total = (size_t)B * Ho * Wo * CV;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < total; i += stride) {
        const int c = (int)(i % CV);
        size_t t = i / CV;
        const int x = (int)(t % Wo);
        t /= Wo;
        const int y = (int)(t % Ho);
        const size_t b = t / Ho;
        int iy = (int)floorf(__fadd_rn(__fmul_rn((float)y, sy), 0.5f));
        int ix = (int)floorf(__fadd_rn(__fmul_rn((float)x, sx), 0.5f));
        iy = iy < H - 1 ? iy : H - 1;
        ix = ix < W - 1 ? ix : W - 1;
        out[i] = in[((b * H + iy) * W + ix) * CV + c];
    }
}

}  // namespace yolo

extern "C" int yolo_upsample_nearest(const float* in, int B, int H, int W, int C, int Ho, int Wo, float* out, void* stream) {
    YOLO_REQUIRE(in && out, YOLO_ERR_ARG, "upsample_nearest: null pointer");
    YOLO_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, YOLO_ERR_SHAPE, "upsample_nearest: non-positive size");
    const float sy = Ho > 1 ? (float)((double)(H - 1) / (double)(Ho - 1)) : 0.f;
    const float sx = Wo > 1 ? (float)((double)(W - 1) / (double)(Wo - 1)) : 0.f;
    const bool vec = C % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0;
    const size_t total = (size_t)B * Ho * Wo * (vec ? C / 4 : C);
    if (vec)
        hipLaunchKernelGGL(yolo::upsample_nearest_kernel<yolo::f32x4>, dim3(yolo::cast_grid(total, 1)), dim3(256), 0, (hipStream_t)stream,
                           (const yolo::f32x4*)in, (yolo::f32x4*)out, B, H, W, C / 4, Ho, Wo, sy, sx);
    else
        hipLaunchKernelGGL(yolo::upsample_nearest_kernel<float>, dim3(yolo::cast_grid(total, 1)), dim3(256), 0, (hipStream_t)stream,
                           in, out, B, H, W, C, Ho, Wo, sy, sx);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_upsample2x(const float* in, int B, int H, int W, int C, float* out, void* stream) {
    YOLO_REQUIRE(in && out, YOLO_ERR_ARG, "upsample2x: null pointer");
    YOLO_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, YOLO_ERR_SHAPE, "upsample2x: bad shape (C %% 4 == 0 required)");
    const size_t total = (size_t)B * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(yolo::upsample2x_kernel, dim3(yolo::cast_grid(total, 1)), dim3(256), 0, (hipStream_t)stream,
                       (const yolo::f32x4*)in, (yolo::f32x4*)out, B, H, W, C / 4);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// ---- reorg (reference yolo_v2/yolo/yolo_v2.py:97-113): space-to-depth 2x2, channel block (dy*2+dx)*C.
// The fused network never runs this (26_conv scatters in its epilogue); layer-level drop-in only.
__global__ void reorg2x_kernel(const yolo::f32x4* __restrict__ in, yolo::f32x4* __restrict__ out, int B, int H, int W, int C4) {
    const size_t total = (size_t)B * H * W * C4;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const int Ho = H >> 1, Wo = W >> 1;
    for (; i < total; i += stride) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const size_t b = t / H;
        out[(((b * Ho + (y >> 1)) * Wo + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1))) * C4 + c] = in[i];
    }
}

extern "C" int yolo_reorg2x(const float* in, int B, int H, int W, int C, float* out, void* stream) {
    YOLO_REQUIRE(in && out, YOLO_ERR_ARG, "reorg2x: null pointer");
    YOLO_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0, YOLO_ERR_SHAPE, "reorg2x: need even H, W and C %% 4 == 0");
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(reorg2x_kernel, dim3(yolo::cast_grid_fwd(total)), dim3(256), 0, (hipStream_t)stream,
                       (const yolo::f32x4*)in, (yolo::f32x4*)out, B, H, W, C / 4);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

extern "C" int yolo_maxpool2x2(const float* in, int B, int H, int W, int C, float* out, void* stream) {
    YOLO_REQUIRE(in && out, YOLO_ERR_ARG, "maxpool2x2: null pointer");
    return yolo::launch_maxpool2x2(YOLO_DTYPE_F32, in, B, H, W, C, out, (hipStream_t)stream);
}
