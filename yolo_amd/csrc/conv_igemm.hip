// Fused implicit-GEMM convolution for gfx950 (MI355X): conv (1x1 / 3x3, stride 1 / 2) + folded BN
// bias + leaky(0.1) + optional residual add + optional 2x nearest-upsample scatter + channel-offset
// (route concat) store.  Replaces slim.conv2d+batch_norm+leaky_relu (reference
// yolo_v3/utils/layer_utils.py:9-21, yolo_v3/model.py:8-12,32-36), `net + shortcut`
// (layer_utils.py:28), upsample_layer (layer_utils.py:67-72) and tf.concat (model.py:49,59).
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = (ky,kx,ci).  NHWC activations make every
// K-chunk of one tap a contiguous run of channels, so an A row of one K-step is 128 contiguous bytes
// (64 bf16 / 32 f32) of one input pixel (or, when Cin*sizeof < 128, of consecutive taps).
//   * staging: global_load_lds_dwordx4 (LDS-DMA, 16 B per lane, no VGPR round trip); zero padding by
//     pointing out-of-image lanes at a zero page.  LDS image is lane-linear; the bank-conflict
//     swizzle (16-B chunk index ^ ((row>>1)&7)) is applied on the per-lane SOURCE chunk and on the
//     ds_read_b128 fragment address.
//   * math: v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact f32), f32 accumulate.
//   * double-buffered LDS, one barrier per 128-byte K-step.
//   * epilogue through LDS so that stores are 16 B per lane, channel-contiguous.
#include "common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __attribute__((aligned(256))) static const unsigned char g_zero_page[256] = {0};

struct ConvK {
    const char* in;
    const char* w;
    const float* bias;
    char* out;
    const char* res;
    int B, H, W, Cin, Ho, Wo, Cout;
    int ksize, stride, pad;
    int in_ld, in_coff, out_ld, out_coff, res_ld, res_coff;
    int kpad_bytes, nsteps;
    int leaky, upsample, out_f32;
    int M, ntm, ntn;
};

#define GLDS16(gptr, ldsptr)                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <bool BF16, int BM, int BN, int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvK p) {
    constexpr int SZ = BF16 ? 2 : 4;
    constexpr int TM = BM / WARPS_M / 32;
    constexpr int TN = BN / WARPS_N / 32;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int AI = BM / 32;   // LDS-DMA instructions per wave per K-step for A
    constexpr int BI = BN / 32;
    constexpr int LDC = BN + 4;   // epilogue tile row stride in floats
    static_assert(WARPS_M * WARPS_N == 4, "4 waves");

    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order: blocks b and b+8 share an XCD (L2); give each XCD a contiguous run of
    // tiles so that the N-tiles of one M-tile (same input rows) and neighbouring M-tiles (3x3 halo)
    // hit the same L2.  Bijective for any grid size.
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = tile / p.ntn;
    const int nt = tile - mt * p.ntn;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    const int CinB = p.Cin * SZ;
    const int ntaps = p.ksize * p.ksize;
    const int HoWo = p.Ho * p.Wo;

    // ---- per-thread loader state ---------------------------------------------------------
    // This lane fills LDS slot (row = 8g + (lane>>3), chunk' = lane&7) of row-group g = wave + 4i;
    // it therefore fetches logical chunk c = chunk' ^ ((row>>1)&7), identical for all its rows.
    const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    long long a_off[AI];
    unsigned a_mask[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = 8 * (wave + 4 * i) + (lane >> 3);
        const int m = m0 + row;
        unsigned mask = 0;
        long long off = 0;
        if (m < p.M) {
            const int b = m / HoWo;
            const int r = m - b * HoWo;
            const int oy = r / p.Wo;
            const int ox = r - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad;
            const int ix0 = ox * p.stride - p.pad;
            off = (((long long)b * p.H + iy0) * p.W + ix0) * (long long)p.in_ld * SZ;
            for (int t = 0; t < ntaps; ++t) {
                const int ky = (t * 11) >> 5, kx = t - ky * 3;   // ksize is 1 or 3
                const int iy = iy0 + ky, ix = ix0 + kx;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << t;
            }
        }
        a_off[i] = off;
        a_mask[i] = mask;
    }
    int tap = 0;
    int cib = chunk * 16;
    while (cib >= CinB) { cib -= CinB; ++tap; }
    const char* wsrc = p.w + (long long)(n0 + 8 * wave + (lane >> 3)) * p.kpad_bytes + chunk * 16;
    const long long w_step = (long long)32 * p.kpad_bytes;
    const int in_cb = p.in_coff * SZ;
    const int pix_b = p.in_ld * SZ;

    auto stage = [&](int buf) {
        char* la = lds + buf * BUF_BYTES + wave * 1024;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        const long long toff = (long long)(ky * p.W + kx) * pix_b + cib + in_cb;
        const unsigned tbit = (tap < ntaps) ? (1u << tap) : 0u;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const char* src = (a_mask[i] & tbit) ? (p.in + a_off[i] + toff) : (const char*)g_zero_page;
            GLDS16(src, la + i * 4096);
        }
        char* lb = lds + buf * BUF_BYTES + A_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16(wsrc + i * w_step, lb + i * 4096);
        // advance to the next K-step
        wsrc += 128;
        cib += 128;
        while (cib >= CinB) { cib -= CinB; ++tap; }
    };

    // ---- fragment addressing ---------------------------------------------------------------
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
    const int wm0 = wm * (TM * 32), wn0 = wn * (TN * 32);
    const int fr = lane & 31, fh = lane >> 5;
    int a_row_off[TM], a_sw[TM], b_row_off[TN], b_sw[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = wm0 + t * 32 + fr;
        a_row_off[t] = row * 128;
        a_sw[t] = (row >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int row = wn0 + t * 32 + fr;
        b_row_off[t] = A_BYTES + row * 128;
        b_sw[t] = (row >> 1) & 7;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop ---------------------------------------------------------------------------
    stage(0);
    __syncthreads();
    for (int s = 0; s < p.nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < p.nsteps) stage(cur ^ 1);
        const char* base = lds + cur * BUF_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u32x4 af[TM], bf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                af[t] = *(const u32x4*)(base + a_row_off[t] + (((2 * kk + fh) ^ a_sw[t]) << 4));
#pragma unroll
            for (int t = 0; t < TN; ++t)
                bf[t] = *(const u32x4*)(base + b_row_off[t] + (((2 * kk + fh) ^ b_sw[t]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (BF16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 av = __builtin_bit_cast(f32x4, af[i]);
                        const f32x4 bv = __builtin_bit_cast(f32x4, bf[j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }

    // ---- epilogue: bias + activation in registers, transpose through LDS, coalesced stores ------
    float* ct = (float*)lds;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn0 + j * 32 + fr;
        const float bv = p.bias[n0 + col];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = acc[i][j][r] + bv;
                if (p.leaky) v = 0.55f * v + 0.45f * __builtin_fabsf(v);
                ct[row * LDC + col] = v;
            }
    }
    __syncthreads();

    if (p.out_f32) {
        // detection heads: Cout = 3*(5+C) is not a multiple of 8 -> scalar f32 stores, channel-coalesced
        float* outf = (float*)p.out;
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int row = idx / BN, col = idx - row * BN;
            const int m = m0 + row, n = n0 + col;
            if (m < p.M && n < p.Cout) outf[(long long)m * p.out_ld + p.out_coff + n] = ct[row * LDC + col];
        }
        return;
    }

    constexpr int CH = BN / 8;
    for (int idx = tid; idx < BM * CH; idx += 256) {
        const int row = idx / CH, ch = idx - row * CH;
        const int m = m0 + row, n = n0 + ch * 8;
        if (m >= p.M || n >= p.Cout) continue;
        f32x4 v0 = *(const f32x4*)(ct + row * LDC + ch * 8);
        f32x4 v1 = *(const f32x4*)(ct + row * LDC + ch * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.res) {
            const char* rp = p.res + ((long long)m * p.res_ld + p.res_coff + n) * SZ;
            if constexpr (BF16) {
                bf16x8 rv = *(const bf16x8*)rp;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            } else {
                f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
            }
        }
        long long pix[4];
        int npix = 1;
        if (p.upsample) {
            const int b = m / HoWo;
            const int r = m - b * HoWo;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            const long long W2 = 2 * p.Wo;
            const long long base = ((long long)b * 2 * p.Ho + 2 * oy) * W2 + 2 * ox;
            pix[0] = base; pix[1] = base + 1; pix[2] = base + W2; pix[3] = base + W2 + 1;
            npix = 4;
        } else {
            pix[0] = m;
        }
        if constexpr (BF16) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            for (int q = 0; q < npix; ++q)
                *(bf16x8*)(p.out + (pix[q] * p.out_ld + p.out_coff + n) * SZ) = o;
        } else {
            f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            for (int q = 0; q < npix; ++q) {
                char* op = p.out + (pix[q] * p.out_ld + p.out_coff + n) * SZ;
                *(f32x4*)op = o0;
                *(f32x4*)(op + 16) = o1;
            }
        }
    }
}

template <bool BF16, int BM, int BN, int WARPS_M, int WARPS_N>
static int launch_cfg(const ConvK& k, hipStream_t stream) {
    constexpr int loop_bytes = 2 * (BM + BN) * 128;
    constexpr int epi_bytes = BM * (BN + 4) * 4;
    constexpr int lds_bytes = loop_bytes > epi_bytes ? loop_bytes : epi_bytes;
    auto fn = conv_igemm_kernel<BF16, BM, BN, WARPS_M, WARPS_N>;
    static bool attr_done[16] = {false};
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 16 && !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_done[dev] = true;
    } else if (dev >= 16) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    }
    ConvK kk = k;
    kk.ntm = (k.M + BM - 1) / BM;
    kk.ntn = (k.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(fn, dim3(kk.ntm * kk.ntn), dim3(256), lds_bytes, stream, kk);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

template <bool BF16>
static int launch_dtype(const ConvK& k, hipStream_t stream) {
    // Tile choice: widest N tile the layer fills; tall M tile when N is narrow (early layers).
    if (k.Cout <= 32) return launch_cfg<BF16, 256, 32, 4, 1>(k, stream);
    if (k.Cout <= 64) return launch_cfg<BF16, 256, 64, 4, 1>(k, stream);
    return launch_cfg<BF16, 128, 128, 2, 2>(k, stream);
}

int launch_conv_igemm(int dtype, const ConvLaunch& p, hipStream_t stream) {
    const int sz = dtype == YOLO_DTYPE_BF16 ? 2 : 4;
    YOLO_REQUIRE(p.ksize == 1 || p.ksize == 3, YOLO_ERR_UNSUPPORTED, "conv: ksize %d not in {1,3}", p.ksize);
    YOLO_REQUIRE((p.Cin * sz) % 16 == 0 && (128 % (p.Cin * sz) == 0 || (p.Cin * sz) % 128 == 0),
                 YOLO_ERR_SHAPE, "conv: Cin=%d not supported by the implicit-GEMM loader", p.Cin);
    YOLO_REQUIRE(p.kpad_bytes % 128 == 0 && p.kpad_bytes >= p.ksize * p.ksize * p.Cin * sz, YOLO_ERR_ARG,
                 "conv: bad packed row length %d", p.kpad_bytes);
    YOLO_REQUIRE(p.in_ld % 8 == 0 && p.in_coff % 8 == 0, YOLO_ERR_SHAPE, "conv: input stride/offset must be multiples of 8");
    if (!p.out_f32)
        YOLO_REQUIRE(p.out_ld % 8 == 0 && p.out_coff % 8 == 0 && p.Cout % 8 == 0, YOLO_ERR_SHAPE,
                     "conv: output stride/offset/Cout must be multiples of 8");
    if (p.res) YOLO_REQUIRE(p.res_ld % 8 == 0 && p.res_coff % 8 == 0, YOLO_ERR_SHAPE, "conv: residual stride");
    ConvK k;
    k.in = (const char*)p.in; k.w = (const char*)p.w; k.bias = p.bias; k.out = (char*)p.out; k.res = (const char*)p.res;
    k.B = p.B; k.H = p.H; k.W = p.W; k.Cin = p.Cin; k.Ho = p.Ho; k.Wo = p.Wo; k.Cout = p.Cout;
    k.ksize = p.ksize; k.stride = p.stride; k.pad = p.pad;
    k.in_ld = p.in_ld; k.in_coff = p.in_coff; k.out_ld = p.out_ld; k.out_coff = p.out_coff;
    k.res_ld = p.res_ld; k.res_coff = p.res_coff;
    k.kpad_bytes = p.kpad_bytes; k.nsteps = p.kpad_bytes / 128;
    k.leaky = p.leaky; k.upsample = p.upsample; k.out_f32 = p.out_f32;
    const long long M = (long long)p.B * p.Ho * p.Wo;
    YOLO_REQUIRE(M > 0 && M < (1ll << 31) - 512, YOLO_ERR_SHAPE, "conv: M=%lld out of range", M);
    k.M = (int)M; k.ntm = k.ntn = 0;
    return dtype == YOLO_DTYPE_BF16 ? launch_dtype<true>(k, stream) : launch_dtype<false>(k, stream);
}

}  // namespace yolo
