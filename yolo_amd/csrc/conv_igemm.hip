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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include "common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __attribute__((aligned(256))) static const unsigned char g_zero_page[256] = {0};

#ifdef YOLO_CONV_STAMPS
// Diagnostic build only (make STAMPS=1): per-phase s_memtime sums of wave 0 of every 64th block.
// Never compiled into the shipped library; stamp values go to a buffer nothing else reads.
__device__ unsigned long long g_stamps[64 * 8];
__device__ unsigned g_census[4096 * 4];      // per workgroup of the halo kernel: HW_ID, XCC_ID, start / end in 100 MHz ticks
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[i] += t_ - tl_; tl_ = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

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
    int leaky, upsample, out_f32, reorg;
    int M, ntm, ntn;
    int splitk;            // > 1: blockIdx.y = K slice; raw f32 accumulators go to partial[slice][M][Cout], no epilogue
    float* partial;
    long long partial_cap; // floats the scratch area holds (host side only)
};

#define GLDS16(gptr, ldsptr)                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

// ---- shared epilogue: bias + activation in registers, transpose through LDS, then per thread one
// (pixel, 8-channel) chunk at a time: residual add, optional 2x2 upsample scatter, 16 B/lane stores.
// All residual loads of a thread are issued before the first one is consumed.
// (A register-direct epilogue with the MFMA operands swapped - 4 consecutive channels per lane, 8-byte
// stores, no LDS - was measured 15-25 % SLOWER per layer: 32 partial lines per store instruction.)
// MS: MFMA tile edge (32: v_mfma_f32_32x32x16_bf16 / 32x32x2_f32, 16 registers per lane; 16: v_mfma_f32_16x16x32_bf16,
// 4 registers per lane).  fr = lane % MS (column), fh = lane / MS; register r holds row (r&3) + 8*(r>>2) + 4*fh.
// PASSES > 1: the staging buffer holds BM / PASSES rows at a time (the waves that own those rows write, everybody stores),
// so that a 256-row tile's epilogue fits the LDS share of two workgroups per CU.
template <bool BF16, int BM, int BN, int TM, int TN, int NT, int MS = 32, typename AccT = f32x16, int PASSES = 1>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, AccT (&acc)[TM][TN], char* lds, int m0, int n0,
                                              int wm0, int wn0, int tid, int fr, int fh
#ifdef YOLO_CONV_STAMPS
                                              , unsigned long long (&st_)[8], unsigned long long& tl_
#endif
                                              ) {
    constexpr int SZ = BF16 ? 2 : 4;
    constexpr int LDC = BN + 4;
    constexpr int BMP = BM / PASSES;            // rows staged per pass
    static_assert(BM % PASSES == 0 && BMP % (TM * MS) == 0, "epilogue passes must be whole wave row blocks");
    float* ct = (float*)lds;
    __syncthreads();                      // every wave is done reading the staging buffers
    STAMP(5);
    const int HoWo = p.Ho * p.Wo;
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
    const int r0 = pass * BMP;                  // first tile row of this pass
    if (PASSES > 1 && pass > 0) __syncthreads();   // the previous pass's staging has been read
    auto to_lds = [&](auto mode) {              // activation mode as a compile-time constant: no branch per element
        constexpr int MODE = decltype(mode)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn0 + j * MS + fr;
            const float bv = p.bias[n0 + col];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < MS * MS / 64; ++r) {
                    const int row = wm0 - r0 + i * MS + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    float v = acc[i][j][r] + bv;
                    if constexpr (MODE == 1) v = leaky_v3(v);
                    else if constexpr (MODE == 2) v = __builtin_fmaxf(0.1f * v, v);
                    ct[row * LDC + col] = v;
                }
        }
    };
    if (PASSES == 1 || (wm0 >= r0 && wm0 < r0 + BMP)) {
        if (p.leaky == 1) to_lds(std::integral_constant<int, 1>{});
        else if (p.leaky == 2) to_lds(std::integral_constant<int, 2>{});
        else to_lds(std::integral_constant<int, 0>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(6);
    __syncthreads();
    STAMP(7);

    if (p.out_f32) {
        // detection heads: Cout = 3*(5+C) is not a multiple of 8 -> scalar f32 stores, channel-coalesced
        float* outf = (float*)p.out;
        for (int idx = tid; idx < BMP * BN; idx += NT) {
            const int row = idx / BN, col = idx - row * BN;
            const int m = m0 + r0 + row, n = n0 + col;
            if (m < p.M && n < p.Cout) outf[(long long)m * p.out_ld + p.out_coff + n] = ct[row * LDC + col];
        }
        continue;
    }

    constexpr int CH = BN / 8;
    constexpr int ITERS = BMP * CH / NT;
    static_assert(BMP * CH % NT == 0, "epilogue geometry");
    u32x4 rres[ITERS][BF16 ? 1 : 2];
    if (p.res) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / CH, ch = idx - row * CH;
            const int m = m0 + r0 + row, n = n0 + ch * 8;
            const bool ok = m < p.M && n < p.Cout;
            const char* rp = p.res + ((long long)(ok ? m : 0) * p.res_ld + p.res_coff + (ok ? n : 0)) * SZ;
            rres[it][0] = *(const u32x4*)rp;
            if constexpr (!BF16) rres[it][1] = *(const u32x4*)(rp + 16);
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NT;
        const int row = idx / CH, ch = idx - row * CH;
        const int m = m0 + r0 + row;
        int n = n0 + ch * 8;
        if (m >= p.M || n >= p.Cout) continue;
        f32x4 v0 = *(const f32x4*)(ct + row * LDC + ch * 8);
        f32x4 v1 = *(const f32x4*)(ct + row * LDC + ch * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.res) {
            if constexpr (BF16) {
                const bf16x8 rv = __builtin_bit_cast(bf16x8, rres[it][0]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            } else {
                const f32x4 r0v = __builtin_bit_cast(f32x4, rres[it][0]), r1v = __builtin_bit_cast(f32x4, rres[it][1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0v[e]; v[4 + e] += r1v[e]; }
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
        } else if (p.reorg) {
            const int b = m / HoWo;
            const int r = m - b * HoWo;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            pix[0] = ((long long)b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
            n += ((oy & 1) * 2 + (ox & 1)) * p.Cout;
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
    }   // pass
}

// counted wait: at most N vector-memory loads (register or LDS-DMA) still pending; loads complete in issue order
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// ---- split-K (small-batch plan): raw f32 accumulators -> partial[slice][M][Cout]; splitk_reduce_kernel finishes the layer ----
// D[pixel][channel] accumulators (fr = lane % MS channel column, fh = lane / MS; register r = pixel row (r&3) + 8*(r>>2) + 4*fh)
template <int TM, int TN, int MS, typename AccT>
__device__ __forceinline__ void store_partial_rows(const ConvK& p, AccT (&acc)[TM][TN], int slice, int m0, int n0, int wm0, int wn0, int fr, int fh) {
    float* const base = p.partial + (long long)slice * p.M * p.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < MS * MS / 64; ++r) {
            const int m = m0 + wm0 + i * MS + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn0 + j * MS + fr;
                if (n < p.Cout) base[(long long)m * p.Cout + n] = acc[i][j][r];
            }
        }
}
// D[channel][pixel] accumulators of the 16x16x32 MFMAs with swapped operands: a lane holds channels 16 j + 4 fq + 0..3 of pixel 16 i + fr
template <int TM, int TN>
__device__ __forceinline__ void store_partial_cols(const ConvK& p, f32x4 (&acc)[TM][TN], int slice, int m0, int n0, int wm0, int wn0, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    float* const base = p.partial + (long long)slice * p.M * p.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + 16 * i + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 16 * j + 4 * fq;
            if (n < p.Cout) *(f32x4*)(base + (long long)m * p.Cout + n) = acc[i][j];      // Cout % 8 == 0 on this path
        }
    }
}
// out = bf16(act(sum over slices in slice order + bias) + residual): one thread per (pixel, 8 channels), the arithmetic of the
// epilogues above on the summed accumulator
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvK p) {
    const int CH = p.Cout >> 3;
    const long long total = (long long)p.M * CH;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int m = (int)(idx / CH), n = (int)(idx - (long long)m * CH) * 8;
    const long long slice = (long long)p.M * p.Cout;
    const float* src = p.partial + (long long)m * p.Cout + n;
    f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
    // four slices' loads in flight at a time (a slice-by-slice loop is a chain of S memory latencies); the adds stay in slice order
    int s = 1;
    for (; s + 3 < p.splitk; s += 4) {
        f32x4 b[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) { b[u][0] = *(const f32x4*)(src + (s + u) * slice); b[u][1] = *(const f32x4*)(src + (s + u) * slice + 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] += b[u][0][e]; a1[e] += b[u][1][e]; }
    }
    for (; s < p.splitk; ++s) {
        const f32x4 b0 = *(const f32x4*)(src + s * slice), b1 = *(const f32x4*)(src + s * slice + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] += b0[e]; a1[e] += b1[e]; }
    }
    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] += p.bias[n + e];
        if (p.leaky == 1) v[e] = leaky_v3(v[e]);
        else if (p.leaky == 2) v[e] = __builtin_fmaxf(0.1f * v[e], v[e]);
    }
    if (p.res) {
        const bf16x8 rv = *(const bf16x8*)(p.res + ((long long)m * p.res_ld + p.res_coff + n) * 2);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    *(bf16x8*)(p.out + ((long long)m * p.out_ld + p.out_coff + n) * 2) = o;
}
static int launch_splitk_reduce(const ConvK& k, hipStream_t stream) {
    const long long total = (long long)k.M * (k.Cout >> 3);
    note_kernel("yolo::splitk_reduce_kernel");
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, k);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// ---- register epilogue (bf16, 16x16x32 MFMAs with the operands swapped: D[channel][pixel]) --------------------------------
// A lane then holds, per 16x16 tile, 4 CONSECUTIVE channels (4 * (lane >> 4) + r) of ONE pixel (lane & 15): bias, activation
// and the residual add happen in registers, v_permlane16_swap pairs two channel tiles so that every lane owns 8 consecutive
// channels (16 bytes) of one pixel, and a store instruction writes 64 contiguous bytes per pixel.  No LDS, no barrier:
// waves finish independently.  Same arithmetic per element as conv_epilogue: (acc + bias) -> activation -> + residual -> bf16.
template <int TM, int TN, int ACT>
__device__ __forceinline__ void conv_epilogue_regs_act(const ConvK& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lane) {
    static_assert(TN % 2 == 0, "channel tiles are stored in pairs");
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = *(const f32x4*)(p.bias + n0 + wn0 + 16 * j + 4 * fq);
    // this lane's 16-byte piece of tile pair q: channel tile 2q + (fq & 1), channels 8 * (fq >> 1) .. + 7 of it
    const int cpiece = wn0 + 16 * (fq & 1) + 8 * (fq >> 1);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + 16 * i + fr;
        const bool mok = m < p.M;
        const long long mrow = mok ? m : 0;
        u32x4 rres[TN / 2];
        if (p.res) {
#pragma unroll
            for (int q = 0; q < TN / 2; ++q) {
                const int n = n0 + cpiece + 32 * q;
                const bool ok = mok && n < p.Cout;
                rres[q] = *(const u32x4*)(p.res + ((ok ? mrow : 0) * p.res_ld + p.res_coff + (ok ? n : 0)) * 2);
            }
        }
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) {
            float va[4], vb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = acc[i][2 * q][r] + bv[2 * q][r];
                float b = acc[i][2 * q + 1][r] + bv[2 * q + 1][r];
                if constexpr (ACT == 1) { a = leaky_v3(a); b = leaky_v3(b); }
                else if constexpr (ACT == 2) { a = __builtin_fmaxf(0.1f * a, a); b = __builtin_fmaxf(0.1f * b, b); }
                va[r] = a; vb[r] = b;
            }
            if (p.res) {
                // the loaded piece is in STORE order; the same swap (an involution) brings it back to accumulator order
                const auto s0 = __builtin_amdgcn_permlane16_swap(rres[q][0], rres[q][2], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(rres[q][1], rres[q][3], false, false);
                const unsigned ra[2] = {s0[0], s1[0]}, rb[2] = {s0[1], s1[1]};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    va[2 * h] += __uint_as_float(ra[h] << 16);
                    va[2 * h + 1] += __uint_as_float(ra[h] & 0xffff0000u);
                    vb[2 * h] += __uint_as_float(rb[h] << 16);
                    vb[2 * h + 1] += __uint_as_float(rb[h] & 0xffff0000u);
                }
            }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            unsigned pa[2], pb[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x2 ta = {(__bf16)va[2 * h], (__bf16)va[2 * h + 1]}, tb = {(__bf16)vb[2 * h], (__bf16)vb[2 * h + 1]};
                pa[h] = __builtin_bit_cast(unsigned, ta);
                pb[h] = __builtin_bit_cast(unsigned, tb);
            }
            const auto o0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
            const auto o1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
            const int n = n0 + cpiece + 32 * q;
            if (mok && n < p.Cout)
                *(u32x4*)(p.out + (mrow * p.out_ld + p.out_coff + n) * 2) = u32x4{o0[0], o1[0], o0[1], o1[1]};
        }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_regs(const ConvK& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lane) {
    if (p.leaky == 1) conv_epilogue_regs_act<TM, TN, 1>(p, acc, m0, n0, wm0, wn0, lane);
    else if (p.leaky == 2) conv_epilogue_regs_act<TM, TN, 2>(p, acc, m0, n0, wm0, wn0, lane);
    else conv_epilogue_regs_act<TM, TN, 0>(p, acc, m0, n0, wm0, wn0, lane);
}

// ---- wave-private epilogue (plain or residual stores only): every wave transposes its own accumulator tile through a
// private LDS region, 32 pixel rows at a time, and stores it itself - one workgroup barrier (the staging buffers are
// free) instead of two per pass, accumulator registers die half a tile at a time, and a wave's stores overlap the
// other waves' transposes.  Same arithmetic as conv_epilogue: (acc + bias) -> activation -> + residual -> bf16.
template <bool BF16, int TM, int TN, int MS, typename AccT>
__device__ __forceinline__ void conv_epilogue_wave(const ConvK& p, AccT (&acc)[TM][TN], char* lds, int m0, int n0,
                                                   int wm0, int wn0, int wave, int lane, int fr, int fh
#ifdef YOLO_CONV_STAMPS
                                                   , unsigned long long (&st_)[8], unsigned long long& tl_
#endif
                                                   ) {
    constexpr int SZ = BF16 ? 2 : 4;
    constexpr int WN = TN * MS;                 // channels of the wave tile
    constexpr int LDW = WN + 4;
    constexpr int SUB = (TM * MS) % 32 == 0 ? 32 : 16;   // rows per sub-pass
    constexpr int NSUB = TM * MS / SUB;
    constexpr int IPS = SUB / MS;               // accumulator row tiles per sub-pass
    constexpr int CH = WN / 8;
    constexpr int ITERS = SUB * CH / 64;
    static_assert(TM * MS % SUB == 0 && SUB % MS == 0 && (SUB * CH) % 64 == 0, "wave epilogue geometry");
    float* ct = (float*)lds + wave * (SUB * LDW);
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = p.bias[n0 + wn0 + j * MS + fr];
    __syncthreads();                            // every wave is done reading the staging buffers
    STAMP(5);
    const int leaky = p.leaky;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const int mrow0 = m0 + wm0 + s * SUB;
        u32x4 rres[ITERS][BF16 ? 1 : 2];
        if (p.res) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = lane + it * 64;
                const int row = idx / CH, ch = idx - row * CH;
                const int m = mrow0 + row, n = n0 + wn0 + ch * 8;
                const bool ok = m < p.M && n < p.Cout;
                const char* rp = p.res + ((long long)(ok ? m : 0) * p.res_ld + p.res_coff + (ok ? n : 0)) * SZ;
                rres[it][0] = *(const u32x4*)rp;
                if constexpr (!BF16) rres[it][1] = *(const u32x4*)(rp + 16);
            }
        }
        auto to_lds = [&](auto mode) {              // activation mode as a compile-time constant: no branch per element
            constexpr int MODE = decltype(mode)::value;
#pragma unroll
            for (int ii = 0; ii < IPS; ++ii) {
                const int i = s * IPS + ii;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < MS * MS / 64; ++r) {
                        const int row = ii * MS + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        float v = acc[i][j][r] + bv[j];
                        if constexpr (MODE == 1) v = leaky_v3(v);
                        else if constexpr (MODE == 2) v = __builtin_fmaxf(0.1f * v, v);
                        ct[row * LDW + j * MS + fr] = v;
                    }
            }
        };
        if (leaky == 1) to_lds(std::integral_constant<int, 1>{});
        else if (leaky == 2) to_lds(std::integral_constant<int, 2>{});
        else to_lds(std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        STAMP(6);
#ifdef YOLO_CONV_STAMPS
        if (p.res) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(1); }     // diagnostic: residual latency still exposed here
#endif
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = lane + it * 64;
            const int row = idx / CH, ch = idx - row * CH;
            const int m = mrow0 + row, n = n0 + wn0 + ch * 8;
            const f32x4 v0 = *(const f32x4*)(ct + row * LDW + ch * 8);
            const f32x4 v1 = *(const f32x4*)(ct + row * LDW + ch * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.res) {
                if constexpr (BF16) {
                    const bf16x8 rv = __builtin_bit_cast(bf16x8, rres[it][0]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                } else {
                    const f32x4 r0v = __builtin_bit_cast(f32x4, rres[it][0]), r1v = __builtin_bit_cast(f32x4, rres[it][1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0v[e]; v[4 + e] += r1v[e]; }
                }
            }
            if (m < p.M && n < p.Cout) {
                char* op = p.out + ((long long)m * p.out_ld + p.out_coff + n) * SZ;
                if constexpr (BF16) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                    *(bf16x8*)op = o;
                } else {
                    *(f32x4*)op = f32x4{v[0], v[1], v[2], v[3]};
                    *(f32x4*)(op + 16) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging rows have been read before the next sub-pass rewrites them
        __builtin_amdgcn_wave_barrier();
        STAMP(7);
    }
}

// MS: MFMA tile edge (32, or 16 = v_mfma_f32_16x16x32_bf16, bf16 only); EPASS = 0: wave-private epilogue (plain / residual
// stores only - the host never picks it for upsample / reorg / f32-head layers); MINW: waves per SIMD to allocate registers for.
template <bool BF16, int BM, int BN, int WARPS_M, int WARPS_N, int STAGES, int MS = 32, int EPASS = 1, int MINW = 1>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, MINW) void conv_igemm_kernel(ConvK p) {
    static_assert(MS == 32 || (MS == 16 && BF16), "MFMA shape");
    constexpr int SZ = BF16 ? 2 : 4;
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int NT = 64 * NW;
    constexpr int TM = BM / WARPS_M / MS;
    constexpr int TN = BN / WARPS_N / MS;
    constexpr int KK = MS / 8;
    constexpr int KSH = MS == 32 ? 5 : 6;
    typedef typename std::conditional<MS == 32, f32x16, f32x4>::type AccT;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int AI = BM / 8 / NW;   // LDS-DMA instructions per wave per K-step for A (8 rows each)
    constexpr int BI = BN / 8 / NW;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 2 == 0, "loader geometry");
    static_assert(STAGES == 2 || STAGES == 3, "ring depth");

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
    // This lane fills LDS slot (row = 8g + (lane>>3), chunk' = lane&7) of row-group g = wave + NW*i;
    // it therefore fetches logical chunk c = chunk' ^ ((row>>1)&7), identical for all its rows.
    const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    long long a_off[AI];
    unsigned a_mask[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = 8 * (wave + NW * i) + (lane >> 3);
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
    // K slice of this workgroup (split-K: blockIdx.y; otherwise the whole K)
    const int nsl = p.splitk > 1 ? (p.nsteps + p.splitk - 1) / p.splitk : p.nsteps;
    const int s_begin = p.splitk > 1 ? (int)blockIdx.y * nsl : 0;
    const int s_end = s_begin + nsl < p.nsteps ? s_begin + nsl : p.nsteps;
    int tap = 0;
    int cib = chunk * 16 + s_begin * 128;
    while (cib >= CinB) { cib -= CinB; ++tap; }
    const char* wsrc = p.w + (long long)(n0 + 8 * wave + (lane >> 3)) * p.kpad_bytes + chunk * 16 + s_begin * 128;
    const long long w_step = (long long)(8 * NW) * p.kpad_bytes;
    const int in_cb = p.in_coff * SZ;
    const int pix_b = p.in_ld * SZ;

    auto stage = [&](int buf) {
        char* la = lds + buf * BUF_BYTES + wave * 1024;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        const long long toff = (long long)(ky * p.W + kx) * pix_b + cib + in_cb;
        const unsigned tbit = (tap < ntaps) ? (1u << tap) : 0u;
#if !defined(YOLO_EXP) || YOLO_EXP != 4         // EXP 4: im2col kernel without activation loads in the loop (first stage only)
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const char* src = (a_mask[i] & tbit) ? (p.in + a_off[i] + toff) : (const char*)g_zero_page;
            GLDS16(src, la + i * (NW * 1024));
        }
#else
        if (buf == 0 && tap == 0 && cib == chunk * 16)
            for (int i = 0; i < AI; ++i) GLDS16((a_mask[i] & tbit) ? (p.in + a_off[i] + toff) : (const char*)g_zero_page, la + i * (NW * 1024));
#endif
        char* lb = lds + buf * BUF_BYTES + A_BYTES + wave * 1024;
#if !defined(YOLO_EXP) || YOLO_EXP != 5         // EXP 5: ... without weight loads
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16(wsrc + i * w_step, lb + i * (NW * 1024));
#endif
        // advance to the next K-step
        wsrc += 128;
        cib += 128;
        while (cib >= CinB) { cib -= CinB; ++tap; }
    };

    // ---- fragment addressing ---------------------------------------------------------------
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
    const int wm0 = wm * (TM * MS), wn0 = wn * (TN * MS);
    const int fr = lane & (MS - 1), fh = lane / MS;
    int a_row_off[TM], a_sw[TM], b_row_off[TN], b_sw[TN];    // *_sw: byte offset of this lane's first k-chunk, swizzled
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int row = wm0 + t * MS + fr;
        a_row_off[t] = row * 128;
        a_sw[t] = (((row >> 1) & 7) ^ fh) << 4;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int row = wn0 + t * MS + fr;
        b_row_off[t] = A_BYTES + row * 128;
        b_sw[t] = (((row >> 1) & 7) ^ fh) << 4;
    }

    AccT acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MS * MS / 64; ++r) acc[i][j][r] = 0.f;

    // ---- main loop: STAGES-deep LDS ring, LDS-DMA prefetch distance STAGES-1, one barrier per step ----
    // Iteration s: (1) wait until this thread's loads of stage s have landed, leaving the younger
    // stages in flight (counted vmcnt); (2) workgroup barrier: every wave's share of stage s is in
    // LDS and every wave has finished reading stage s-1; (3) refill the buffer of stage s-1 with
    // stage s+STAGES-1; (4) fragments + MFMAs of stage s.
    constexpr int LPS = AI + BI;      // LDS-DMA instructions per thread per stage
#ifdef YOLO_CONV_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tbeg_ = __builtin_amdgcn_s_memtime();
    unsigned long long tl_ = tbeg_;
#endif
    const int nsteps = s_end - s_begin;
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nsteps) stage(st);
    int cur = 0;
    STAMP(0);                          // prologue: address set-up + first stage issue
    for (int s = 0; s < nsteps; ++s) {
        if (STAGES == 3 && s + 1 < nsteps) {
            if constexpr (LPS == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (LPS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if constexpr (LPS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (LPS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (LPS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (LPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (LPS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        STAMP(1);                      // waited for own loads
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);                      // barrier
        if (s + STAGES - 1 < nsteps) {
            int nb = cur + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            stage(nb);
        }
        STAMP(3);                      // stage issue
        const char* base = lds + cur * BUF_BYTES;
        // all fragment reads of the K-step are issued up front: the MFMAs start as soon as their
        // operands land (counted lgkmcnt) instead of exposing one LDS latency per 16-deep k slice
        u32x4 af[KK][TM], bf[KK][TN];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
                af[kk][t] = *(const u32x4*)(base + a_row_off[t] + ((kk << KSH) ^ a_sw[t]));
#pragma unroll
            for (int t = 0; t < TN; ++t)
                bf[kk][t] = *(const u32x4*)(base + b_row_off[t] + ((kk << KSH) ^ b_sw[t]));
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (MS == 16 && EPASS == 3) {       // D[channel][pixel] for the register epilogue
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, bf[kk][j]), __builtin_bit_cast(bf16x8, af[kk][i]), acc[i][j], 0, 0, 0);
                    } else if constexpr (MS == 16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, af[kk][i]), __builtin_bit_cast(bf16x8, bf[kk][j]), acc[i][j], 0, 0, 0);
                    } else if constexpr (BF16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[kk][i]), __builtin_bit_cast(bf16x8, bf[kk][j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 av = __builtin_bit_cast(f32x4, af[kk][i]);
                        const f32x4 bv = __builtin_bit_cast(f32x4, bf[kk][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
        cur = cur + 1 == STAGES ? 0 : cur + 1;
#ifdef YOLO_CONV_STAMPS
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]));
        STAMP(4);                      // fragment reads + MFMA issue
#endif
    }
    if (p.splitk > 1) {                // uniform: raw partial tile, the reduce launch finishes the layer
        if constexpr (EPASS == 3 && MS == 16) store_partial_cols<TM, TN>(p, acc, blockIdx.y, m0, n0, wm0, wn0, lane);
        else store_partial_rows<TM, TN, MS, AccT>(p, acc, blockIdx.y, m0, n0, wm0, wn0, fr, fh);
        return;
    }
#if defined(YOLO_EXP) && YOLO_EXP == 6          // EXP 6: im2col kernel without epilogue
    {
        float sacc = 0.f;
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < MS * MS / 64; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) ((float*)p.out)[tid] = sacc;
    }
#else
    if constexpr (EPASS == 3) {
        static_assert(EPASS != 3 || (MS == 16 && BF16), "register epilogue: bf16 on 16x16x32 MFMAs");
        if constexpr (MS == 16) conv_epilogue_regs<TM, TN>(p, acc, m0, n0, wm0, wn0, lane);
        STAMP(5);
    } else if constexpr (EPASS == 0) {
#ifdef YOLO_CONV_STAMPS
        conv_epilogue_wave<BF16, TM, TN, MS, AccT>(p, acc, lds, m0, n0, wm0, wn0, wave, lane, fr, fh, st_, tl_);
#else
        conv_epilogue_wave<BF16, TM, TN, MS, AccT>(p, acc, lds, m0, n0, wm0, wn0, wave, lane, fr, fh);
#endif
    } else {
#ifdef YOLO_CONV_STAMPS
    conv_epilogue<BF16, BM, BN, TM, TN, NT, MS, AccT, EPASS>(p, acc, lds, m0, n0, wm0, wn0, tid, fr, fh, st_, tl_);
#else
    conv_epilogue<BF16, BM, BN, TM, TN, NT, MS, AccT, EPASS>(p, acc, lds, m0, n0, wm0, wn0, tid, fr, fh);
#endif
    }
#endif
#ifdef YOLO_CONV_STAMPS
    {
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();
        st_[0] = t_ - tl_;             // residual + stores (the prologue slot is reused)
        tl_ = t_;
    }
    if ((blockIdx.x & 63) == 0 && tid == 0 && (blockIdx.x >> 6) < 64) {
        unsigned long long* o = g_stamps + (blockIdx.x >> 6) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_[i];
    }
#endif
}

#ifdef YOLO_CONV_STAMPS
extern "C" int yolo_debug_read_stamps(unsigned long long* host, int n) {
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_stamps)) != hipSuccess) return YOLO_ERR_HIP;
    if (hipDeviceSynchronize() != hipSuccess) return YOLO_ERR_HIP;
    if (hipMemcpy(host, dptr, sizeof(unsigned long long) * (size_t)(n < 512 ? n : 512), hipMemcpyDeviceToHost) != hipSuccess) return YOLO_ERR_HIP;
    return YOLO_OK;
}
extern "C" int yolo_debug_read_census(unsigned* host, int n) {
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_census)) != hipSuccess) return YOLO_ERR_HIP;
    if (hipDeviceSynchronize() != hipSuccess) return YOLO_ERR_HIP;
    if (hipMemcpy(host, dptr, sizeof(unsigned) * (size_t)(n < 16384 ? n : 16384), hipMemcpyDeviceToHost) != hipSuccess) return YOLO_ERR_HIP;
    return YOLO_OK;
}
#endif

// LDS swizzle of the halo kernel: 16-byte chunk c of patch / weight row r lives at chunk c ^ lds_swz<MS>(r) of its 128-byte row.
// ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, ... : MI355X_MICROARCH.md, LDS), 16 bytes = 4 of the
// 64 banks per lane, two rows per bank wrap.  32-row fragments (one chunk column per group): (r >> 1) & 7 separates any 16
// rows.  16-row fragments of v_mfma_f32_16x16x32_bf16 put chunk columns c and c ^ 1 into one group (lanes 0-15 / 16-31):
// with (r >> 1) & 7 two rows whose terms differ by exactly 1 collide whenever the fragment's first row is not a multiple of
// 4 - every shifted tap of the patch (SQ_LDS_BANK_CONFLICT was 27 % of the LDS cycles).  An even term, 2 * ((r >> 1) & 3),
// never maps c and c ^ 1 of two rows onto each other and still separates the four same-parity rows of a column: conflict-free
// for every first row.
template <int MS>
__device__ __forceinline__ int lds_swz(int row) { return MS == 16 ? ((row >> 1) & 3) << 1 : (row >> 1) & 7; }

// ---- 3x3 stride-1 convolution with an LDS-resident input patch ("halo" kernel) -------------------
// The im2col kernel above re-loads every input pixel once per tap (9x) and is bound by the global->LDS
// fill rate (measured ~900 cycles per 32 KB stage, the MFMAs of that stage need 512).  Here a tile of BM
// consecutive output pixels (flattened b,y,x) stages, per 128-byte channel chunk, the BM + 2W + 2
// consecutive input pixel rows it touches ONCE; the nine taps are shifted fragment reads of that patch
// (patch row = output row + ky*W + kx).  Taps that fall outside the image (where the flattened
// neighbour is a real pixel of the previous/next row or image) are zeroed in registers with a per-row
// 9-bit mask.  Only the [BN x 128 B] weight tile is loaded per tap.  Global->LDS traffic per FLOP drops
// ~2.4x (BM=256) against im2col at the same tile.
// K order of the packed weights stays (ky,kx,ci); the loop order here is (ci-chunk, tap).
// SPB (single patch buffer): the patch of the next channel chunk is loaded only after the last tap of the current
// one (one exposed load latency per chunk) - half the patch LDS, so that two workgroups fit a CU at larger W.
// MS = 16 (bf16 only): the same wave tile out of v_mfma_f32_16x16x32_bf16 - same LDS reads and MFMA cycles per FLOP;
// the chip holds a higher clock on that shape (MI355X_MICROARCH.md, DVFS give-back item 7).
// EPASS: epilogue staging passes (conv_epilogue); MINW: waves per SIMD the register allocation must allow.
// WB: weight-tile buffers.  2 = one tile of lookahead behind a full wait (the throughput plan: LDS for two workgroups per
// CU).  > 2 (small-batch plan, one short K slice per workgroup, LDS to spare) = WB - 1 tiles in flight behind a COUNTED
// wait (XP build only: measured at batch 1, a 9-step slice takes 15 us with WB = 2, 4 or 6).
template <bool BF16, int BM, int BN, int WARPS_M, int WARPS_N, bool SPB, int MS, int EPASS = 1, int MINW = 1, int WB = 2>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, MINW) void conv3x3_halo_kernel(ConvK p, int prows) {
    static_assert(MS == 32 || (MS == 16 && BF16), "MFMA shape");
    static_assert(WB == 2 || (SPB && WB <= 8), "deep weight ring: single patch buffer only (no patch loads between the counted weight loads)");
    static_assert(BM % (WARPS_M * MS) == 0 && BN % (WARPS_N * MS) == 0, "wave tiles must be whole MFMA tiles");
    constexpr int SZ = BF16 ? 2 : 4;
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int NT = 64 * NW;
    constexpr int TM = BM / WARPS_M / MS;
    constexpr int TN = BN / WARPS_N / MS;
    constexpr int KK = MS / 8;                    // MFMA k-slices per 128-byte chunk: 4 x 32 B or 2 x 64 B
    constexpr int KSH = MS == 32 ? 5 : 6;
    typedef typename std::conditional<MS == 32, f32x16, f32x4>::type AccT;
    constexpr int B_BYTES = BN * 128;
    constexpr int BI = BN / 8 / NW;
    static_assert(BN % (8 * NW) == 0 && NW % 2 == 0 && BI >= 1, "loader geometry");

    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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

    const int W = p.W;
    const int CinB = p.Cin * SZ;
    const int nchunks = CinB >> 7;
    const int patch_bytes = prows * 128;          // prows: BM + 2W + 2 rounded up to 8
    char* const patch0 = lds;
    char* const bbuf0 = lds + ((nchunks > 1 && !SPB) ? 2 : 1) * patch_bytes;   // one patch buffer suffices for one chunk
    char* const zrow = bbuf0 + WB * B_BYTES;      // 128 zero bytes: source of out-of-image taps
    if (tid < 32) ((unsigned*)zrow)[tid] = 0u;    // ordered before the first fragment read by the loop's barrier

    // ---- loader state ----------------------------------------------------------------------------
    const int chunk = (lane & 7) ^ lds_swz<MS>(8 * (wave & 1) + (lane >> 3));   // rows 8 (wave + NW i) + (lane >> 3), NW even
    const int ngroups = prows >> 3;               // 8-row LDS-DMA pieces of the patch
    // Addresses are a wave-uniform 64-bit base (SALU) + a 32-bit per-lane offset, so a piece costs almost
    // no VALU.  Patch rows outside the tensor are only ever referenced by masked taps: they are clamped
    // into range instead of zero-filled (the mask redirects those fragment reads to `zrow`).
    const int qlane = m0 - W - 1 + (lane >> 3);   // input pixel (flattened) of this lane's row in piece 0
    const unsigned pix_b = (unsigned)(p.in_ld * SZ);
    const char* const in_base = p.in + (long long)p.in_coff * SZ;
    const char* const w_base = p.w + (long long)n0 * p.kpad_bytes;
    unsigned w_off[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) w_off[i] = (unsigned)(8 * (wave + NW * i) + (lane >> 3)) * (unsigned)p.kpad_bytes + chunk * 16;

    // patch piece `slot` of this wave for channel chunk c -> patch buffer c&1
    auto load_patch_piece = [&](int c, int slot) {
        const int g = wave + NW * slot;
        if (g < ngroups) {
            int q = qlane + 8 * g;
            q = q < 0 ? 0 : (q >= p.M ? p.M - 1 : q);
            const unsigned off = (unsigned)q * pix_b + chunk * 16;
            GLDS16(in_base + c * 128 + off, patch0 + (SPB ? 0 : (c & 1) * patch_bytes) + g * 1024);
        }
    };
    // weight tile of (chunk c, tap t) -> weight buffer `buf`
    auto load_b = [&](int c, int t, int buf) {
        const char* src = w_base + ((long long)t * CinB + c * 128);
        char* lb = bbuf0 + buf * B_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16(src + w_off[i], lb + i * (NW * 1024));
    };
    const int slots = (ngroups + NW - 1) / NW;    // patch pieces per wave per chunk

    // ---- fragment addressing -----------------------------------------------------------------------
    const int wm = wave / WARPS_N, wn = wave - wm * WARPS_N;
    const int wm0 = wm * (TM * MS), wn0 = wn * (TN * MS);
    const int fr = lane & (MS - 1), fh = lane / MS;
    unsigned a_mask[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + wm0 + t * MS + fr;
        unsigned mask = 0;
        if (m < p.M) {
            const int r = m % (p.H * W);
            const int y = r / W, x = r - y * W;
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) {
                const int iy = y + tt / 3 - 1, ix = x + tt % 3 - 1;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < W) mask |= 1u << tt;
            }
        }
        a_mask[t] = mask;
    }
    int b_row_off[TN], b_sx[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int row = wn0 + t * MS + fr;
        b_row_off[t] = row * 128;
        b_sx[t] = (lds_swz<MS>(row) ^ fh) << 4;
    }

    AccT acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MS * MS / 64; ++r) acc[i][j][r] = 0.f;

#ifdef YOLO_CONV_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tbeg_ = __builtin_amdgcn_s_memtime();
    const unsigned long long rbeg_ = __builtin_amdgcn_s_memrealtime();   // 100 MHz: in-kernel clock = cycles / ticks x 100 MHz
    unsigned long long tl_ = tbeg_;
#endif
    // channel chunks of this workgroup (split-K: blockIdx.y takes nchunks / splitk of them; otherwise all)
    const int cper = p.splitk > 1 ? nchunks / p.splitk : nchunks;
    const int c_begin = p.splitk > 1 ? (int)blockIdx.y * cper : 0;
    const int c_end = c_begin + cper;
    // ---- prologue: whole patch of the first chunk + its first weight tile ----------------------------
    for (int sl = 0; sl < slots; ++sl) load_patch_piece(c_begin, sl);
    load_b(c_begin, 0, 0);
    // deep weight ring: the tiles of steps 1 .. WB-2 as well; (lc, lt) = the next tile to request, `ahead` = requested tiles
    // beyond the current step's
    int lc = c_begin, lt = 1, ahead = 0;
    if constexpr (WB > 2) {
        const int total = (c_end - c_begin) * 9;
        for (int u = 1; u < WB - 1 && u < total; ++u) {
            load_b(lc, lt, u);
            if (++lt == 9) { lt = 0; ++lc; }
            ++ahead;
        }
    }

    // ---- main loop over (chunk c, tap t); one barrier per step ----------------------------------------
    // Loads issued during step s (next weight tile, 1/9 of the next chunk's patch) must have landed by
    // the barrier that opens step s+1; the buffers they overwrite were last read in step s-1.
    int bcur = 0;
    STAMP(0);
    for (int c = c_begin; c < c_end; ++c) {
        const char* patch = patch0 + (SPB ? 0 : (c & 1) * patch_bytes);
        if (SPB && c > c_begin) {              // every wave is done with chunk c-1: refill the single patch buffer
            __builtin_amdgcn_s_barrier();
#if !defined(YOLO_EXP) || YOLO_EXP != 1         // EXP 1: no refill (stale patch): what the exposed refills cost
            for (int sl = 0; sl < slots; ++sl) load_patch_piece(c, sl);
#endif
        }
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            if constexpr (WB > 2) {
                // loads complete in order: with WB-2 younger tiles requested the current one has landed once at most
                // (WB-2) * BI loads are pending.  A patch refill (first step of a later chunk) is younger than all of them.
                if (ahead == WB - 2 && !(t == 0 && c > c_begin)) wait_vmcnt<(WB - 2) * BI>();
                else wait_vmcnt<0>();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            STAMP(1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            STAMP(2);
            // prefetch
#if !defined(YOLO_EXP) || YOLO_EXP != 2         // EXP 2: no weight loads in the loop (stale weights)
            if constexpr (WB > 2) {
                // the buffer read in the previous step (free since this step's barrier) takes the tile WB-1 steps ahead
                if (lc < c_end) {
                    load_b(lc, lt, bcur == 0 ? WB - 1 : bcur - 1);
                    if (++lt == 9) { lt = 0; ++lc; }
                } else {
                    --ahead;                   // nothing left to request: the ring drains (full waits from here on)
                }
            } else {
                if (t < 8) load_b(c, t + 1, bcur ^ 1);
                else if (c + 1 < c_end) load_b(c + 1, 0, bcur ^ 1);
            }
#endif
            if (!SPB && c + 1 < c_end)
                for (int sl = t; sl < slots; sl += 9) load_patch_piece(c + 1, sl);
            STAMP(3);
            // fragments of this tap
            const int ky = (t * 11) >> 5, kx = t - ky * 3;
            const int shift = ky * W + kx;
            const char* bbase = bbuf0 + bcur * B_BYTES;
            u32x4 af[KK][TM], bf[KK][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int prow = wm0 + i * MS + fr + shift;
                const bool ok = (a_mask[i] >> t) & 1;
                const char* rowp = ok ? patch + prow * 128 : zrow;     // masked tap: read zeros
                const int sx = (lds_swz<MS>(prow) ^ fh) << 4;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) af[kk][i] = *(const u32x4*)(rowp + ((kk << KSH) ^ sx));
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[kk][j] = *(const u32x4*)(bbase + b_row_off[j] + ((kk << KSH) ^ b_sx[j]));
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (MS == 16 && EPASS == 3) {       // D[channel][pixel] for the register epilogue
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, bf[kk][j]), __builtin_bit_cast(bf16x8, af[kk][i]), acc[i][j], 0, 0, 0);
                        } else if constexpr (MS == 16) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, af[kk][i]), __builtin_bit_cast(bf16x8, bf[kk][j]), acc[i][j], 0, 0, 0);
                        } else if constexpr (BF16) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, af[kk][i]), __builtin_bit_cast(bf16x8, bf[kk][j]), acc[i][j], 0, 0, 0);
                        } else {
                            const f32x4 av = __builtin_bit_cast(f32x4, af[kk][i]);
                            const f32x4 bv = __builtin_bit_cast(f32x4, bf[kk][j]);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[i][j], 0, 0, 0);
                        }
                    }
            }
            if constexpr (WB > 2) bcur = bcur + 1 == WB ? 0 : bcur + 1;
            else bcur ^= 1;
#ifdef YOLO_CONV_STAMPS
            asm volatile("s_nop 0" :: "v"(acc[0][0][0]));
            STAMP(4);
#endif
        }
    }
    if (p.splitk > 1) {                // uniform: raw partial tile, the reduce launch finishes the layer
        if constexpr (EPASS == 3 && MS == 16) store_partial_cols<TM, TN>(p, acc, blockIdx.y, m0, n0, wm0, wn0, lane);
        else store_partial_rows<TM, TN, MS, AccT>(p, acc, blockIdx.y, m0, n0, wm0, wn0, fr, fh);
        return;
    }
#if defined(YOLO_EXP) && YOLO_EXP == 3          // EXP 3: no epilogue (accumulators kept live, one store per lane)
    {
        float sacc = 0.f;
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < MS * MS / 64; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) ((float*)p.out)[tid] = sacc;
    }
#else
    if constexpr (EPASS == 3) {
        static_assert(EPASS != 3 || (MS == 16 && BF16), "register epilogue: bf16 on 16x16x32 MFMAs");
        if constexpr (MS == 16) conv_epilogue_regs<TM, TN>(p, acc, m0, n0, wm0, wn0, lane);
        STAMP(5);
    } else if constexpr (EPASS == 0) {
#ifdef YOLO_CONV_STAMPS
        conv_epilogue_wave<BF16, TM, TN, MS, AccT>(p, acc, lds, m0, n0, wm0, wn0, wave, lane, fr, fh, st_, tl_);
#else
        conv_epilogue_wave<BF16, TM, TN, MS, AccT>(p, acc, lds, m0, n0, wm0, wn0, wave, lane, fr, fh);
#endif
    } else {
#ifdef YOLO_CONV_STAMPS
    conv_epilogue<BF16, BM, BN, TM, TN, NT, MS, AccT, EPASS>(p, acc, lds, m0, n0, wm0, wn0, tid, fr, fh, st_, tl_);
#else
    conv_epilogue<BF16, BM, BN, TM, TN, NT, MS, AccT, EPASS>(p, acc, lds, m0, n0, wm0, wn0, tid, fr, fh);
#endif
    }
#endif
#ifdef YOLO_CONV_STAMPS
    {
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();
        st_[0] = ((__builtin_amdgcn_s_memrealtime() - rbeg_) << 32) | ((t_ - tbeg_) & 0xffffffffull);   // whole tile: 100 MHz ticks | shader cycles
        tl_ = t_;
    }
    if ((blockIdx.x & 63) == 0 && tid == 0 && (blockIdx.x >> 6) < 64) {
        unsigned long long* o = g_stamps + (blockIdx.x >> 6) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_[i];
    }
    if (tid == 0 && blockIdx.x < 4096) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_census[blockIdx.x * 4 + 0] = hw;
        g_census[blockIdx.x * 4 + 1] = xcc;
        g_census[blockIdx.x * 4 + 2] = (unsigned)rbeg_;
        g_census[blockIdx.x * 4 + 3] = (unsigned)__builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef YOLO_EXPERIMENTAL_TILINGS
// ---- 3x3 stride-1 convolution, 256x256 tile, ping-pong wave rows ("p8", bf16 only) -----------------------------------
// The LDS-resident patch of conv3x3_halo_kernel under the GEMM schedule measured in scripts/probe/gemm8p.hip (1.0-1.2
// PFLOP/s on random bf16 data against ~0.83 for the 128x128 / two-workgroup structure above):
//   * one workgroup of 8 waves per CU, 2 (pixel rows) x 4 (channels); a wave owns 128 pixels x 64 channels as 8 x 4 tiles of
//     v_mfma_f32_16x16x32_bf16 (128 accumulator VGPRs): 24 ds_read_b128 per 64 MFMAs instead of 12 per 8
//   * a K-step is one (128-byte channel chunk, tap) pair in four phases of 16 MFMAs (one 64 x 32 quadrant); fragment
//     reads 12 / 4 / 8 / 0 per phase; the two wave rows run half a phase apart (two barriers per phase), so one row's MFMAs
//     cover the other row's LDS reads and LDS-DMA issue
//   * the next step's weight tile [256 x 128 B] is requested in phases 0-2 (1 / 2 / 1 LDS-DMAs per wave), the next chunk's
//     patch one piece per wave and step behind them, so that the counted vmcnt of phase 3 waits for the weights only and
//     a patch piece has two steps to arrive
// Patch geometry, tap masks, zero row, swizzle, weight packing: exactly as in conv3x3_halo_kernel.
template <int MT>
__global__ __launch_bounds__(512) void conv3x3_p8_kernel(ConvK p, int prows) {
    constexpr int QH = 16 * MT;                 // pixel rows of a quadrant; a wave owns 2 QH rows, the tile 4 QH
    constexpr int RH = 2 * QH;
    constexpr int BM = 2 * RH, BN = 256;
    constexpr int B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
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

    const int W = p.W;
    const int CinB = p.Cin * 2;
    const int nchunks = CinB >> 7;
    const int patch_bytes = prows * 128;
    char* const patch0 = lds;
    char* const bbuf0 = lds + (nchunks > 1 ? 2 : 1) * patch_bytes;
    char* const zrow = bbuf0 + 2 * B_BYTES;
    if (tid < 32) ((unsigned*)zrow)[tid] = 0u;

    // ---- loader state (see conv3x3_halo_kernel)
    const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const int ngroups = prows >> 3;
    const int qlane = m0 - W - 1 + (lane >> 3);
    const unsigned pix_b = (unsigned)(p.in_ld * 2);
    const char* const in_base = p.in + (long long)p.in_coff * 2;
    const char* const w_base = p.w + (long long)n0 * p.kpad_bytes;
    unsigned w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w_off[i] = (unsigned)(8 * (wave + 8 * i) + (lane >> 3)) * (unsigned)p.kpad_bytes + chunk * 16;
    auto load_patch_piece = [&](int c, int g) {
        int q = qlane + 8 * g;
        q = q < 0 ? 0 : (q >= p.M ? p.M - 1 : q);
        const unsigned off = (unsigned)q * pix_b + chunk * 16;
        GLDS16(in_base + c * 128 + off, patch0 + (c & 1) * patch_bytes + g * 1024);
    };
    auto load_b = [&](int c, int t, int buf, int i) {            // piece wave + 8 i of the weight tile of (chunk c, tap t)
        GLDS16(w_base + ((long long)t * CinB + c * 128) + w_off[i], bbuf0 + buf * B_BYTES + (wave + 8 * i) * 1024);
    };

    // ---- fragment addressing (16x16x32: lane row = l & 15, 16-byte k-chunk = 4 kk + (l >> 4))
    const int fr = lane & 15, fq = lane >> 4;
    unsigned a_mask[2][MT];
#pragma unroll
    for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rt = wr * RH + mq * QH + i * 16 + fr;
            const int m = m0 + rt;
            unsigned mask = 0;
            if (m < p.M) {
                const int r = m % (p.H * W);
                const int y = r / W, x = r - y * W;
#pragma unroll
                for (int tt = 0; tt < 9; ++tt) {
                    const int iy = y + tt / 3 - 1, ix = x + tt % 3 - 1;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < W) mask |= 1u << tt;
                }
            }
            a_mask[mq][i] = mask;
        }
    unsigned b_off[2][2];
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wc * 64 + nq * 32 + j * 16 + fr;
            b_off[nq][j] = row * 128 + ((fq ^ ((row >> 1) & 7)) << 4);
        }

    f32x4 acc[2][MT][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 af[MT][2], bf0[2][2], bf1[2][2];

#define P8_BAR()                            \
    do {                                    \
        __builtin_amdgcn_sched_barrier(0);  \
        __builtin_amdgcn_s_barrier();       \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)
#define P8_LGKM0()                                           \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)

    // rows of one lane's eight A fragments are 16 apart, so they share the swizzle term ((row >> 1) & 7): one base
    // offset per tap, the fragments at constant distances from it
    const int zrow_off = (int)(zrow - lds) + (fq << 4);
    const int rt0 = wr * RH + fr;
    int a_base = 0;
    auto set_tap = [&](const char* patch, int shift) {
        const int prow0 = rt0 + shift;
        a_base = (int)(patch - lds) + prow0 * 128 + ((fq ^ ((prow0 >> 1) & 7)) << 4);
    };
    auto read_a = [&](int mq, int t) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bool ok = (a_mask[mq][i] >> t) & 1;
            const int off = ok ? a_base + (mq * QH + i * 16) * 128 : zrow_off;   // masked tap: zeros
            af[i][0] = *(const u32x4*)(lds + off);
            af[i][1] = *(const u32x4*)(lds + (off ^ 64));
        }
    };
    auto read_b = [&](const char* bbase, int nq, u32x4 (&bf)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf[j][0] = *(const u32x4*)(bbase + b_off[nq][j]);
            bf[j][1] = *(const u32x4*)(bbase + (b_off[nq][j] ^ 64));
        }
    };
    auto mfma_quadrant = [&](int mq, int nq, u32x4 (&bf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mq][i][nq][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, af[i][kk]), __builtin_bit_cast(bf16x8, bf[j][kk]), acc[mq][i][nq][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: patch of chunk 0 + weight tile (0,0); the second wave row then falls one barrier behind
    for (int g = wave; g < ngroups; g += 8) load_patch_piece(0, g);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_b(0, 0, 0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_BAR();
    if (wr == 1) P8_BAR();

    int bcur = 0;
    for (int c = 0; c < nchunks; ++c) {
        const char* patch = patch0 + (c & 1) * patch_bytes;
        const bool next_chunk = c + 1 < nchunks;
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int ky = (t * 11) >> 5, kx = t - ky * 3;
            const int shift = ky * W + kx;
            const char* bbase = bbuf0 + bcur * B_BYTES;
            const bool more = t < 8 || next_chunk;
            const int nc = t < 8 ? c : c + 1, ntap = t < 8 ? t + 1 : 0, nb = bcur ^ 1;
            const int piece = t * 8 + wave;                       // patch piece of the next chunk this wave requests in this step
            const bool has_piece = next_chunk && piece < ngroups;
            // phase 0: quadrant (0,0)
            set_tap(patch, shift);
            read_b(bbase, 0, bf0);
            read_a(0, t);
            if (more) load_b(nc, ntap, nb, 0);
            P8_BAR();
            P8_LGKM0();
            mfma_quadrant(0, 0, bf0);
            P8_BAR();
            // phase 1: quadrant (0,1)
            read_b(bbase, 1, bf1);
            if (more) { load_b(nc, ntap, nb, 1); load_b(nc, ntap, nb, 2); }
            P8_BAR();
            P8_LGKM0();
            mfma_quadrant(0, 1, bf1);
            P8_BAR();
            // phase 2: quadrant (1,1)
            read_a(1, t);
            if (more) load_b(nc, ntap, nb, 3);
            if (has_piece) load_patch_piece(c + 1, piece);        // youngest load of the step
            P8_BAR();
            P8_LGKM0();
            mfma_quadrant(1, 1, bf1);
            P8_BAR();
            // phase 3: quadrant (1,0) out of registers.  Before the barrier that lets the other wave row read the next
            // weight tile, this wave's share of it has landed; a patch piece may stay in flight except before a new chunk.
            if (wr == 1) {
                if (has_piece && t < 8) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            P8_BAR();
            mfma_quadrant(1, 0, bf0);
            if (wr == 0) {
                if (has_piece && t < 8) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            P8_BAR();
            bcur = nb;
        }
    }
    if (wr == 0) P8_BAR();                      // both wave rows have passed the same number of barriers

    // ---- epilogue in two passes (pixel rows mq*64 .. +63 of both wave rows): bias + activation -> f32 staging in LDS
    // [2 QH][260] -> per thread (pixel, 8 channels): residual add, bf16, 16-byte store
    constexpr int LDC = BN + 4;
    float* const ct = (float*)lds;
    for (int mq = 0; mq < 2; ++mq) {
        if (mq == 1) P8_BAR();                  // staging of pass 0 fully read
        auto to_lds = [&](auto mode) {
            constexpr int MODE = decltype(mode)::value;
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = wc * 64 + nq * 32 + j * 16 + fr;
                    const float bv = p.bias[n0 + col];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = wr * QH + i * 16 + 4 * fq + r;
                            float v = (mq == 0 ? acc[0][i][nq][j][r] : acc[1][i][nq][j][r]) + bv;
                            if constexpr (MODE == 1) v = leaky_v3(v);
                            else if constexpr (MODE == 2) v = __builtin_fmaxf(0.1f * v, v);
                            ct[row * LDC + col] = v;
                        }
                }
        };
        if (p.leaky == 1) to_lds(std::integral_constant<int, 1>{});
        else if (p.leaky == 2) to_lds(std::integral_constant<int, 2>{});
        else to_lds(std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P8_BAR();
        constexpr int ITERS = RH * (BN / 8) / 512;      // 2 MT
        u32x4 rres[ITERS];
        if (p.res) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = tid + it * 512;
                const int row = idx >> 5, ch = idx & 31;
                const int m = m0 + (row / QH) * RH + mq * QH + (row % QH);
                const bool ok = m < p.M;
                rres[it] = *(const u32x4*)(p.res + ((long long)(ok ? m : 0) * p.res_ld + p.res_coff + n0 + ch * 8) * 2);
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = tid + it * 512;
            const int row = idx >> 5, ch = idx & 31;
            const int m = m0 + (row / QH) * RH + mq * QH + (row % QH);
            if (m >= p.M) continue;
            const f32x4 v0 = *(const f32x4*)(ct + row * LDC + ch * 8);
            const f32x4 v1 = *(const f32x4*)(ct + row * LDC + ch * 8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.res) {
                const bf16x8 rv = __builtin_bit_cast(bf16x8, rres[it]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *(bf16x8*)(p.out + ((long long)m * p.out_ld + p.out_coff + n0 + ch * 8) * 2) = o;
        }
    }
#undef P8_BAR
#undef P8_LGKM0
}

template <int MT>
static int launch_p8(const ConvK& k, hipStream_t stream) {
    constexpr int BM = 64 * MT;
    const int prows = (int)align_up((size_t)(BM + 2 * k.W + 2), 8);
    const int nchunks = k.Cin * 2 / 128;
    const int loop_bytes = (nchunks > 1 ? 2 : 1) * prows * 128 + 2 * 256 * 128 + 128;
    constexpr int epi_bytes = 32 * MT * (256 + 4) * 4;
    const int lds_bytes = loop_bytes > epi_bytes ? loop_bytes : epi_bytes;
    YOLO_REQUIRE(lds_bytes <= 160 * 1024, YOLO_ERR_UNSUPPORTED, "conv3x3 p8: W=%d needs %d bytes of LDS", k.W, lds_bytes);
    YOLO_REQUIRE(prows <= 512, YOLO_ERR_UNSUPPORTED, "conv3x3 p8: W=%d too wide", k.W);
    YOLO_REQUIRE(k.Cout % 256 == 0, YOLO_ERR_UNSUPPORTED, "conv3x3 p8: Cout=%d not a multiple of 256", k.Cout);
    YOLO_REQUIRE((long long)k.M * k.in_ld * 2 < (1ll << 32), YOLO_ERR_SHAPE, "conv3x3 p8: input larger than 4 GiB");
    auto fn = conv3x3_p8_kernel<MT>;
    static std::atomic<bool> attr_done[16];   // zero-initialised; setting the attribute twice is harmless, the flag only must not tear
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev < 16) attr_done[dev] = true;
    }
    ConvK kk = k;
    kk.ntm = (k.M + BM - 1) / BM;
    kk.ntn = k.Cout / 256;
    note_kernel("yolo::conv3x3_p8_kernel<%d>", MT);
    hipLaunchKernelGGL(fn, dim3(kk.ntm * kk.ntn), dim3(512), lds_bytes, stream, kk, prows);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

#endif  // YOLO_EXPERIMENTAL_TILINGS

// ---- 1x1 stride-1 convolution with the activations loaded straight into MFMA operand registers ("direct-A", bf16) ----
// The im2col kernel stages A and B through LDS by LDS-DMA; on the 1x1 layers (4 - 16 K-steps per tile) it is bound by the
// issue cost and the landing latency of those pieces, not by HBM or the matrix pipe (round 3: dropping either operand's
// pieces from the loop made the 1x1 layers 1.3 - 1.5x faster).  A 1x1 conv needs no im2col: a pixel's K-chunk is 128
// contiguous bytes of its row, and a wave that owns whole pixels x ALL BN output channels is the only reader of its rows.
//   * each wave owns 32 pixels x BN channels (2 x TN tiles of v_mfma_f32_16x16x32_bf16, operands swapped: D[channel][pixel],
//     so the register epilogue applies); lane (pixel fr, quarter fq) loads 32 contiguous bytes of its pixel's 128-byte
//     K-chunk with two global_load_dwordx4 - the four lanes of a pixel cover one whole cache line.  MFMA kk of the step
//     takes the lane's kk-th 16 bytes, i.e. the k-chunk 2 fq + kk (any k permutation is fine as long as B agrees).
//   * only the [BN x 128 B] weight tile goes through LDS (LDS-DMA, double-buffered, shared by the workgroup's waves).  Its
//     rows are stored with the k-chunks permuted (logical chunk 2 q + k at slot 4 k + q) so that the fragment reads use the
//     addressing - and the conflict-free XOR swizzle - of the im2col kernel.
//   * activation loads run two K-steps ahead in two register sets; they are inline asm so that hipcc neither counts nor
//     drains them (cdna_hip_programming.md 5.7 form (ii)): the wait before a set's first use names its registers.
//     vmcnt order per step: [weights s+1 (BI pieces)] ... [activations s+2 (4 loads)], so vmcnt(4) at the top of step s+1
//     leaves exactly the youngest activation set in flight.
#define GLOAD16(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr))
// D: K-steps the loads run ahead (D activation register sets, D + 1 weight buffers); NW waves of 32 pixels x BN = 16 TN channels.
// PERM: a lane's two loads are the adjacent 16-byte pieces 2 fq, 2 fq + 1 of the K-chunk (described above); !PERM: pieces fq and
// 4 + fq (the plain fragment order: every load instruction reads 64 contiguous bytes per pixel).
// TAPS: the same kernel for a 3x3 convolution of any stride (the stride-2 layers, whose patch is too large for the LDS-patch
// kernel): K-step s is (tap s / chunks, chunk s % chunks); the lane's pixel pointer moves to the tap's input pixel, out-of-image
// taps read a zero page - the im2col gather done by address arithmetic instead of through LDS.
template <int NW, int TN, int D, int MINW, bool PERM = true, bool TAPS = false>
__global__ __launch_bounds__(64 * NW, MINW) void conv1x1_direct_kernel(ConvK p) {
    static_assert(!(TAPS && PERM), "tap addressing uses the plain piece order");
    constexpr int TM = 2;
    constexpr int BM = NW * 32, BN = TN * 16;
    constexpr int B_BYTES = BN * 128;
    constexpr int BI = BN / 8 / NW;
    constexpr int V = BI + 2 * TM;                    // vector-memory operations a wave issues per K-step
    static_assert(BN % (8 * NW) == 0 && NW % 2 == 0 && BI >= 1, "weight loader geometry");
    static_assert(D >= 2 && D <= 4 && V * (D - 1) < 64, "prefetch depth");
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int fr = lane & 15, fq = lane >> 4;
    // K slice of this workgroup (split-K: blockIdx.y; otherwise the whole K)
    const int nsl = p.splitk > 1 ? (p.nsteps + p.splitk - 1) / p.splitk : p.nsteps;
    const int s_begin = p.splitk > 1 ? (int)blockIdx.y * nsl : 0;
    const int nsteps = (s_begin + nsl < p.nsteps ? s_begin + nsl : p.nsteps) - s_begin;

    // ---- weight loader: this lane fills slot (lane & 7) of row 8 g + (lane >> 3), g = wave + NW i
    const int phys = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);      // slot ^ swizzle(row) = 4 k + q of the stored chunk
    const int chunk = PERM ? 2 * (phys & 3) + (phys >> 2) : phys;             // ... which is logical k-chunk 2 q + k
    const char* wsrc = p.w + (long long)(n0 + 8 * wave + (lane >> 3)) * p.kpad_bytes + chunk * 16 + s_begin * 128;
    const long long w_step = (long long)(8 * NW) * p.kpad_bytes;
    auto load_b = [&](int buf) {
        char* lb = lds + buf * B_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16(wsrc + i * w_step, lb + i * (NW * 1024));
        wsrc += 128;
    };
    // ---- activation rows of this lane (rows past M are clamped; the epilogue drops them)
    const char* aptr[TM];
    unsigned amask[TM];                               // TAPS: bit t = tap t of this pixel lies inside the image
    const int CinB = p.Cin * 2;
    const int pix_b = p.in_ld * 2;
    int ltap = 0, lcib = s_begin * 128;               // TAPS: tap and channel byte offset of the next activation load
    if constexpr (TAPS) while (lcib >= CinB) { lcib -= CinB; ++ltap; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = m0 + wave * 32 + i * 16 + fr;
        m = m < p.M ? m : p.M - 1;
        if constexpr (TAPS) {
            const int HoWo = p.Ho * p.Wo;
            const int b = m / HoWo, r = m - b * HoWo;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            aptr[i] = p.in + (((long long)b * p.H + iy0) * p.W + ix0) * pix_b + p.in_coff * 2 + fq * 16;
            unsigned mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << t;
            }
            amask[i] = mask;
        } else {
            aptr[i] = p.in + ((long long)m * p.in_ld + p.in_coff) * 2 + fq * (PERM ? 32 : 16) + s_begin * 128;
            amask[i] = 0;
        }
    }
    int b_off[TN], b_sw[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = j * 16 + fr;
        b_off[j] = row * 128;
        b_sw[j] = (((row >> 1) & 7) ^ fq) << 4;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 a[D][TM][2];                                // register set d: K-steps s with s % D == d
    auto load_a = [&](u32x4 (&x)[TM][2]) {
        if constexpr (TAPS) {
            const int ky = (ltap * 11) >> 5, kx = ltap - ky * 3;
            const long long toff = (long long)(ky * p.W + kx) * pix_b + lcib;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const char* src = ((amask[i] >> ltap) & 1) ? aptr[i] + toff : (const char*)g_zero_page + fq * 16;
                GLOAD16(x[i][0], src, 0);
                GLOAD16(x[i][1], src, 64);
            }
            lcib += 128;
            if (lcib >= CinB) { lcib = 0; ++ltap; }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                GLOAD16(x[i][0], aptr[i], 0);
                if constexpr (PERM) GLOAD16(x[i][1], aptr[i], 16);
                else GLOAD16(x[i][1], aptr[i], 64);
                aptr[i] += 128;
            }
        }
    };
    auto compute = [&](const char* base, u32x4 (&x)[TM][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 bf[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const u32x4*)(base + b_off[j] + ((kk << 6) ^ b_sw[j]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, x[i][kk]), acc[i][j], 0, 0, 0);
        }
    };
    // One K-step.  Issue order of a wave's vector-memory operations is [weights u][activations u] for u = 0, 1, 2, ...; at the
    // top of step s everything up to u = s must have landed, the (at most D - 1) younger groups may stay in flight.  This
    // wave's share: counted vmcnt; the other waves' weight pieces: the barrier.  Then the weights of s + D go into the buffer
    // last read in step s - 1, step s is computed, and the register set just consumed is refilled for step s + D.
    int bcur = 0;
    auto step = [&](int s, u32x4 (&x)[TM][2]) {
        const int younger = nsteps - 1 - s;           // groups issued after step s's
        if (younger >= D - 1) wait_vmcnt<V * (D - 1)>();
        else if (D > 3 && younger == 2) wait_vmcnt<V * 2>();
        else if (D > 2 && younger == 1) wait_vmcnt<V>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]));   // no consumer of the set is scheduled above this point
#if !defined(YOLO_EXP) || YOLO_EXP != 8         // EXP 8: direct kernel without weight loads in the loop
        if (s + D < nsteps) load_b(bcur == 0 ? D : bcur - 1);
#endif
        compute(lds + bcur * B_BYTES, x);
        __builtin_amdgcn_sched_barrier(0);
#if !defined(YOLO_EXP) || YOLO_EXP != 7         // EXP 7: ... without activation loads in the loop
        if (s + D < nsteps) load_a(x);
#endif
        bcur = bcur == D ? 0 : bcur + 1;
    };
    static_assert(TM == 2, "the fence names the four registers of a set");

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nsteps) { load_b(d); load_a(a[d]); }
    for (int s = 0; s < nsteps; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (s + d < nsteps) step(s + d, a[d]);
    }
    if (p.splitk > 1) { store_partial_cols<TM, TN>(p, acc, blockIdx.y, m0, n0, wave * 32, 0, lane); return; }
#if defined(YOLO_EXP) && YOLO_EXP == 9          // EXP 9: ... without epilogue
    {
        float sacc = 0.f;
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 4; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) ((float*)p.out)[tid] = sacc;
    }
#else
    conv_epilogue_regs<TM, TN>(p, acc, m0, n0, wave * 32, 0, lane);
#endif
}
#undef GLOAD16

template <int NW, int TN, int D, int MINW, bool PERM = true, bool TAPS = false>
static int launch_direct1x1(const ConvK& k, hipStream_t stream) {
    constexpr int BM = NW * 32, BN = TN * 16;
    constexpr int lds_bytes = (D + 1) * BN * 128;
    YOLO_REQUIRE((TAPS ? k.ksize == 3 : (k.ksize == 1 && k.stride == 1)) && (k.Cin * 2) % 128 == 0 && !k.upsample && !k.reorg && !k.out_f32, YOLO_ERR_UNSUPPORTED,
                 "conv direct: plain bf16 1x1 stride-1 (3x3 for the tap variant) layers with Cin a multiple of 64 only");
    auto fn = conv1x1_direct_kernel<NW, TN, D, MINW, PERM, TAPS>;
    static std::atomic<bool> attr_done[16];
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        if (dev < 16) attr_done[dev] = true;
    }
    ConvK kk = k;
    kk.ntm = (k.M + BM - 1) / BM;
    kk.ntn = (k.Cout + BN - 1) / BN;
    note_kernel("yolo::conv1x1_direct_kernel<%d, %d, %d, %d, %s, %s>", NW, TN, D, MINW, PERM ? "true" : "false", TAPS ? "true" : "false");
    hipLaunchKernelGGL(fn, dim3(kk.ntm * kk.ntn, kk.splitk > 1 ? kk.splitk : 1), dim3(64 * NW), lds_bytes, stream, kk);
    YOLO_HIP_CHECK(hipGetLastError());
    return kk.splitk > 1 ? launch_splitk_reduce(kk, stream) : YOLO_OK;
}

template <bool BF16, int BM, int BN, int WARPS_M, int WARPS_N, bool SPB = false, int MS = 32, int EPASS = 1, int MINW = 1, int WB = 2>
static int launch_halo(const ConvK& k, hipStream_t stream) {
    if constexpr (MS == 16 && !BF16) return launch_halo<BF16, BM, BN, WARPS_M, WARPS_N, SPB, 32, EPASS == 3 ? 0 : EPASS, MINW, WB>(k, stream);
    else {
    const int prows = (int)align_up((size_t)(BM + 2 * k.W + 2), 8);
    const int nchunks = k.Cin * (BF16 ? 2 : 4) / 128;
    const int loop_bytes = ((nchunks > 1 && !SPB) ? 2 : 1) * prows * 128 + WB * BN * 128 + 128;
    constexpr int epi_bytes = EPASS == 3 ? 0 : EPASS == 0 ? WARPS_M * WARPS_N * 32 * (BN / WARPS_N + 4) * 4 : BM / (EPASS ? EPASS : 1) * (BN + 4) * 4;
    const int lds_bytes = loop_bytes > epi_bytes ? loop_bytes : epi_bytes;
    YOLO_REQUIRE(lds_bytes <= 160 * 1024, YOLO_ERR_UNSUPPORTED, "conv3x3 halo: W=%d needs %d bytes of LDS", k.W, lds_bytes);
    YOLO_REQUIRE((long long)k.M * k.in_ld * (BF16 ? 2 : 4) < (1ll << 32), YOLO_ERR_SHAPE, "conv3x3 halo: input larger than 4 GiB");
    auto fn = conv3x3_halo_kernel<BF16, BM, BN, WARPS_M, WARPS_N, SPB, MS, EPASS, MINW, WB>;
    static std::atomic<bool> attr_done[16];   // zero-initialised; setting the attribute twice is harmless, the flag only must not tear
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev < 16) attr_done[dev] = true;
    }
    ConvK kk = k;
    kk.ntm = (k.M + BM - 1) / BM;
    kk.ntn = (k.Cout + BN - 1) / BN;
    note_kernel("yolo::conv3x3_halo_kernel<%s, %d, %d, %d, %d, %s, %d, %d, %d, %d>", BF16 ? "true" : "false", BM, BN, WARPS_M, WARPS_N, SPB ? "true" : "false", MS, EPASS, MINW, WB);
    hipLaunchKernelGGL(fn, dim3(kk.ntm * kk.ntn, kk.splitk > 1 ? kk.splitk : 1), dim3(64 * WARPS_M * WARPS_N), lds_bytes, stream, kk, prows);
    YOLO_HIP_CHECK(hipGetLastError());
    return kk.splitk > 1 ? launch_splitk_reduce(kk, stream) : YOLO_OK;
    }
}

template <bool BF16, int BM, int BN, int WARPS_M, int WARPS_N, int STAGES, int MS = 32, int EPASS = 1, int MINW = 1>
static int launch_cfg(const ConvK& k, hipStream_t stream) {
    if constexpr (MS == 16 && !BF16) return launch_cfg<BF16, BM, BN, WARPS_M, WARPS_N, STAGES, 32, EPASS == 3 ? 0 : EPASS, MINW>(k, stream);
    else {
    constexpr int loop_bytes = STAGES * (BM + BN) * 128;
    constexpr int epi_bytes = EPASS == 3 ? 0 : EPASS == 0 ? WARPS_M * WARPS_N * 32 * (BN / WARPS_N + 4) * 4 : BM / (EPASS ? EPASS : 1) * (BN + 4) * 4;
    constexpr int lds_bytes = loop_bytes > epi_bytes ? loop_bytes : epi_bytes;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    if (EPASS == 0 || EPASS == 3) YOLO_REQUIRE(!k.upsample && !k.reorg && !k.out_f32, YOLO_ERR_UNSUPPORTED, "conv: wave epilogue on a scatter / f32-head layer");
    auto fn = conv_igemm_kernel<BF16, BM, BN, WARPS_M, WARPS_N, STAGES, MS, EPASS, MINW>;
    static std::atomic<bool> attr_done[16];   // zero-initialised; setting the attribute twice is harmless, the flag only must not tear
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        if (dev < 16) attr_done[dev] = true;
    }
    ConvK kk = k;
    kk.ntm = (k.M + BM - 1) / BM;
    kk.ntn = (k.Cout + BN - 1) / BN;
    if (MS == 32 && EPASS == 1 && MINW == 1)
        note_kernel("yolo::conv_igemm_kernel<%s, %d, %d, %d, %d, %d>", BF16 ? "true" : "false", BM, BN, WARPS_M, WARPS_N, STAGES);
    else
        note_kernel("yolo::conv_igemm_kernel<%s, %d, %d, %d, %d, %d, %d, %d, %d>", BF16 ? "true" : "false", BM, BN, WARPS_M, WARPS_N, STAGES, MS, EPASS, MINW);
    hipLaunchKernelGGL(fn, dim3(kk.ntm * kk.ntn, kk.splitk > 1 ? kk.splitk : 1), dim3(64 * WARPS_M * WARPS_N), lds_bytes, stream, kk);
    YOLO_HIP_CHECK(hipGetLastError());
    return kk.splitk > 1 ? launch_splitk_reduce(kk, stream) : YOLO_OK;
    }
}

// ---- tile configurations -------------------------------------------------------------------------------------------------
// The shipped library holds the configurations the default choice below can pick plus a few alternates the parity tests use as
// independent cross-checks (5: im2col + LDS epilogue, 81 / 88: wave-private / register epilogue, 96: 8-wave direct-A, 75 = 72 / 73 /
// 74: the halo tiles with the register epilogue).  yolo_debug_set_conv_cfg(id) forces one (tests, scripts); the environment
// variable YOLO_CONV_CFG is read ONCE, when the library first launches a conv, as the initial value.  Everything that was measured
// and rejected (ping-pong tile, 192- / 320-row tiles, deeper rings, ...) is compiled only with -DYOLO_EXPERIMENTAL_TILINGS
// (make XP=1 -> libyolo_hip_xp.so), which also reads YOLO_HALO_CFG / YOLO_IGEMM_CFG ("<W>=<cfg>,..." / "<k>x<Wo>=<cfg>,...").
static std::atomic<int> g_forced_cfg{-2};              // -2: not initialised, -1: default choice
static int forced_cfg() {
    int v = g_forced_cfg.load(std::memory_order_relaxed);
    if (v == -2) {
        const char* e = getenv("YOLO_CONV_CFG");
        v = e ? atoi(e) : -1;
        if (v < -1) v = -1;
        int expect = -2;
        g_forced_cfg.compare_exchange_strong(expect, v);
        v = g_forced_cfg.load(std::memory_order_relaxed);
    }
    return v;
}
extern "C" int yolo_debug_set_conv_cfg(int cfg) {
    g_forced_cfg.store(cfg < 0 ? -1 : cfg, std::memory_order_relaxed);
    return YOLO_OK;
}

#ifdef YOLO_EXPERIMENTAL_TILINGS
static int env_map_cfg(const char* var, int ksize, int w) {       // "<W>=<cfg>,..." (ksize 0) or "<k>x<Wo>=<cfg>,..."
    static std::mutex mu;
    static std::map<std::string, std::string> cache;
    std::string text;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(var);
        if (it == cache.end()) { const char* e = getenv(var); it = cache.emplace(var, e ? e : "").first; }
        text = it->second;
    }
    for (const char* q = text.c_str(); q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr) {
        int ks = 0, ww = 0, c = 0;
        if (ksize == 0 ? (sscanf(q, "%d=%d", &ww, &c) == 2 && ww == w) : (sscanf(q, "%dx%d=%d", &ks, &ww, &c) == 3 && ks == ksize && ww == w)) return c;
    }
    return -1;
}
#endif

// Largest divisor S of `units` with tiles * S <= 512 workgroups and at least `min_units` K units per slice (1 = no split).
static int pick_splitk(int tiles, int units, int min_units) {
    int best = 1;
    for (int s = 2; s <= units; ++s)
        if (units % s == 0 && units / s >= min_units && (long long)tiles * s <= 512) best = s;
    return best;
}

template <bool BF16>
static int launch_dtype(const ConvK& k_in, hipStream_t stream) {
    ConvK k = k_in;
    int cfg = forced_cfg();
    const int sz = BF16 ? 2 : 4;
    const bool plain = !k.upsample && !k.reorg && !k.out_f32;
    const bool halo_ok = k.ksize == 3 && k.stride == 1 && (k.Cin * sz) % 128 == 0 && k.Cout > 64 && plain;
    const bool direct_ok = BF16 && k.ksize == 1 && k.stride == 1 && plain && (k.Cin * 2) % 128 == 0;
    const bool direct3_ok = BF16 && k.ksize == 3 && plain && (k.Cin * 2) % 128 == 0 && k.Cout > 64;
    // ---- small-batch plan (opt-in per handle): a layer whose 128 x 128 tiles would occupy fewer than half the CUs runs its
    // K loop in S slices on S times as many workgroups (the latency of such a launch is its K-step chain, 0.5 - 1 us a step),
    // f32 partial tiles + one reduce launch.  The sums are added in slice order, so the result is deterministic, but it is
    // NOT bit-identical to the unsplit plan (within the bf16 tolerance: the f32 accumulation order differs).
    if constexpr (BF16) {
        if (k.partial && cfg < 0 && plain && k.Cout % 8 == 0) {
            const int tiles = ((k.M + 127) / 128) * ((k.Cout + 127) / 128);
            int S = 1, pick = -1;
            if (tiles < 128) {
                if (halo_ok) {
                    if ((int)align_up((size_t)(128 + 2 * k.W + 2), 8) * 128 + 2 * 128 * 128 + 128 <= 160 * 1024) {
                        // eight waves of 32 x 64 on 32x32x16 MFMAs: a short slice at one workgroup per CU is a chain of LDS-read and
                        // load latencies (stamps at batch 1: 470 wait + 220 issue + 920 fragment/MFMA cycles per step for 512 MFMA
                        // cycles), which two waves per SIMD overlap: batch-1 forward 0.965 ms with the four-wave tile 68, 0.893
                        // with 69, 0.877 with 24, 0.867 with 28; 16 waves of 32 x 32 0.88 - 0.91
                        S = pick_splitk(tiles, k.Cin * 2 / 128, 1); pick = 28;
                    }
                }
                // 1x1: a launch of up to 8 K steps already sits at the ~6 us floor of launch + first-load latency (measured at
                // batch 1: 52x52 6.4 us split or not, 26x26 8.8 vs 6.0 + 4.0 for the reduce launch): only K >= 1024 gains
                else if (direct_ok && k.Cout > 64 && k.nsteps >= 16) { S = pick_splitk(tiles, k.nsteps, 2); pick = 97; }
                else if (direct3_ok && k.stride == 2) { S = pick_splitk(tiles, k.nsteps, 4); pick = 98; }
            }
#ifdef YOLO_EXPERIMENTAL_TILINGS
            if (pick == 28) { static const int e = [] { const char* v = getenv("YOLO_SPLIT_HALO_CFG"); return v ? atoi(v) : 0; }(); if (e > 0) pick = e; }
#endif
            if (S > 1 && (long long)S * k.M * k.Cout <= k.partial_cap) { k.splitk = S; cfg = pick; }
        }
    }
    if (k.splitk <= 1) { k.splitk = 1; k.partial = nullptr; }
#ifdef YOLO_EXPERIMENTAL_TILINGS
    if (halo_ok && cfg < 0) cfg = env_map_cfg("YOLO_HALO_CFG", 0, k.W);
    if (cfg >= 43 && cfg <= 46) {                               // ping-pong tile on the small (43, 44) / large (45, 46) maps only
        if constexpr (BF16) {
            if (halo_ok && k.Cout % 256 == 0 && (cfg <= 44 ? k.W <= 32 : k.W > 32) && 256 + 2 * k.W + 2 <= 504)
                return (cfg & 1) ? launch_p8<4>(k, stream) : launch_p8<3>(k, stream);
        }
        cfg = -1;
    }
    if (cfg >= 53 && cfg <= 56) {                               // the 256 x 128 two-workgroup tile on the small maps only
        if (halo_ok && k.W <= 32) cfg = cfg == 53 ? 50 : cfg == 54 ? 51 : cfg == 55 ? 60 : 61;
        else cfg = -1;
    }
    if (cfg == 67) cfg = !halo_ok ? -1 : k.W <= 32 ? 62 : 63;    // the round-1 tiling with the wave-private epilogue
    if (cfg >= 65 && cfg <= 66) {                               // ... and up to 52 x 52
        if (halo_ok && k.W <= 52) cfg = cfg == 65 ? 60 : 61;
        else cfg = -1;
    }
    if (!halo_ok && cfg < 0 && k.Cout > 64) cfg = env_map_cfg("YOLO_IGEMM_CFG", k.ksize, k.Wo);
#endif
    if (cfg == 75) cfg = !halo_ok ? -1 : k.W <= 16 ? 73 : k.W <= 52 ? 72 : 74;   // the shipped tile choice with the register epilogue
    if (cfg >= 80 && cfg < 90 && (halo_ok || k.Cout <= 64)) cfg = -1;
    if (cfg >= 90 && cfg < 98 && !direct_ok) cfg = -1;
    if (cfg >= 98 && cfg < 102 && (!direct3_ok || k.stride == 1)) cfg = -1;   // forced: the stride-2 layers only
    if (cfg >= 102 && cfg < 104 && (!direct3_ok || k.stride != 1 || k.W <= 64)) cfg = -1;   // XP: the 104 x 104 stride-1 layers
    if (((cfg >= 20 && cfg < 80) || (cfg >= 110 && cfg < 120)) && !halo_ok) cfg = -1;
    if (cfg >= 0 && cfg < 10 && k.Cout <= 64) cfg = -1;
    if (cfg >= 10 && cfg < 20 && (k.Cout > 64)) cfg = -1;
    if (cfg < 0) {
        // measured per layer shape at batch 64 (scripts/bench_layer.py, profiles/): the LDS-resident patch
        // kernel wins on every 3x3 stride-1 layer it fits; small grids (many tiles of few pixels) prefer
        // 128x128 tiles at 2 blocks/CU, large grids 256x128 at 8 waves.
        const int prows256 = 256 + 2 * k.W + 2 + 7;
        const int patch_bufs = k.Cin * sz > 128 ? 2 : 1;
        if (k.Cout <= 32) cfg = 10;
        else if (k.Cout <= 64) cfg = direct_ok ? 97 : k.ksize == 1 ? 14 : 12;
        // 256 x 128 tile, 8 waves of 64 x 64 on 16x16x32 MFMAs, wave-private epilogue, two workgroups per CU (round 2:
        // +9 % end to end over 24 / 28 on the 13x13 .. 52x52 maps)
        else if (halo_ok && BF16 && k.W <= 16) cfg = 68;     // 13x13: 128 x 128 tiles of 4 such waves fill the CUs more evenly
        // 52 x 52: the same tile as four waves of 64 x 128 (0.4 - 0.7 % per step over 61 in eleven interleaved A/B rounds; at 26 x 26 no
        // resolvable difference, at 13 x 13 worse)
        else if (halo_ok && BF16 && k.W > 32 && k.W <= 52) cfg = 115;
        else if (halo_ok && BF16 && k.W <= 52) cfg = 61;
        else if (halo_ok && k.W <= 32) cfg = 24;
        // larger maps: a single patch buffer keeps two 128x128 workgroups per CU (measured 10 % faster than the
        // 16-wave 256x128 workgroup at 52x52, 7 % at 104x104)
        else if (halo_ok && (int)align_up((size_t)(128 + 2 * k.W + 2), 8) * 128 + 2 * 128 * 128 + 128 <= 80 * 1024) cfg = BF16 ? 69 : 28;
        else if (halo_ok && patch_bufs * prows256 * 128 + 2 * 128 * 128 + 128 <= 160 * 1024) cfg = 26;
        // plain bf16 1x1 layers: activations straight into MFMA operand registers (round 3: 104x104 65 -> 54 us, 52x52 37 -> 35,
        // 26x26 30 -> 27 per launch on cold inputs; bit-identical to 88).  At 13x13 (K = 1024, four channel tiles re-read every
        // pixel row) a launch on cold inputs is slower (28 vs 25 us) but the step is 0.5 - 1 % faster (three interleaved A/B runs
        // of 100 steps): in the network the rows come from L2 / MALL and the small LDS footprint lets other lanes' kernels share the CU.
        else if (direct_ok) cfg = 97;
        // stride-2 3x3 layers: the same kernel with tap addressing (their patch does not fit the LDS-patch kernel at two
        // workgroups per CU); 0.6 - 0.8 % per step over the im2col kernel in two interleaved A/B sessions of 3 x 60 steps
        else if (direct3_ok && k.stride == 2) cfg = 98;
        else cfg = (BF16 && plain) ? 88 : 5;   // 88: 5 on 16x16x32 MFMAs with the register epilogue (+8 % on the 1x1 layers)
    }
    switch (cfg) {
        case 5: return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);     // 8 waves, 2 blocks/CU
        case 24: return launch_halo<BF16, 128, 128, 4, 2>(k, stream);      // 8 waves, 2 blocks/CU
        case 26: return launch_halo<BF16, 256, 128, 8, 2>(k, stream);      // 16 waves, 1 block/CU
        case 28: return launch_halo<BF16, 128, 128, 4, 2, true>(k, stream);   // single patch buffer: 2 blocks/CU up to W ~ 60
        // 256 x 128 tile, 8 waves of 64 x 64 (half the LDS fragment reads and LDS-DMA pieces per MFMA of 24 / 28), single
        // patch buffer + wave-private epilogue: <= 80 KB of LDS up to W = 52 and <= 128 VGPRs, so two workgroups per CU
        case 61: return launch_halo<BF16, 256, 128, 4, 2, true, 16, 0, 4>(k, stream);
        case 68: return launch_halo<BF16, 128, 128, 2, 2, true, 16, 0, 2>(k, stream);   // 4 waves of 64 x 64, up to three workgroups per CU
        // 256 x 128 tile as FOUR waves of 64 x 128: a wave covers all 128 channels of its rows, 0.375 instead of 0.5 LDS fragment reads
        // per MFMA and half the waves at every barrier; 243 VGPRs = two waves per SIMD, two workgroups per CU (78 KB at 52 x 52)
        case 115:
            if constexpr (BF16) return launch_halo<true, 256, 128, 4, 1, true, 16, 0, 2>(k, stream);
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 69: return launch_halo<BF16, 128, 128, 4, 2, true, 16, 0, 1>(k, stream);   // 28 on 16x16x32 with the wave-private epilogue
        case 72: return launch_halo<BF16, 256, 128, 4, 2, true, 16, 3, 4>(k, stream);   // 61 / 68 / 69 with the register epilogue
        case 73: return launch_halo<BF16, 128, 128, 2, 2, true, 16, 3, 2>(k, stream);
        case 74: return launch_halo<BF16, 128, 128, 4, 2, true, 16, 3, 1>(k, stream);
        // im2col kernel on 16x16x32 MFMAs with the wave-private (81) / register (88) epilogue (plain and residual layers only)
        case 81: if (plain) return launch_cfg<BF16, 128, 128, 4, 2, 2, 16, 0, 1>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
        case 88: if (plain) return launch_cfg<BF16, 128, 128, 4, 2, 2, 16, 3, 1>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
        case 89: if (plain) return launch_cfg<BF16, 192, 128, 4, 2, 2, 16, 3, 2>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 88 with 48 x 64 waves: 80 KB, two workgroups per CU
        case 96: case 97:                                                   // direct-A 1x1 kernel, 8 / 4 waves per workgroup
            if constexpr (BF16) {
                if (direct_ok) {
                    if (k.Cout <= 64) return cfg & 1 ? launch_direct1x1<4, 4, 2, 2, false>(k, stream) : launch_direct1x1<8, 4, 2, 2, false>(k, stream);
                    return cfg & 1 ? launch_direct1x1<4, 8, 2, 2, false>(k, stream) : launch_direct1x1<8, 8, 2, 2, false>(k, stream);
                }
            }
            return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
        case 98: case 99:                                                   // direct-A kernel on a 3x3 conv (tap addressing), 8 / 4 waves
            if constexpr (BF16) {
                if (direct3_ok) return cfg & 1 ? launch_direct1x1<4, 8, 2, 2, false, true>(k, stream) : launch_direct1x1<8, 8, 2, 2, false, true>(k, stream);
            }
            return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
        case 12: return launch_cfg<BF16, 256, 64, 4, 2, 2>(k, stream);     // 8 waves
        case 14: return launch_cfg<BF16, 128, 64, 4, 2, 2>(k, stream);     // 8 waves, small tile
        case 10: return launch_cfg<BF16, 256, 32, 4, 1, 2>(k, stream);
#ifdef YOLO_EXPERIMENTAL_TILINGS
        case 102: case 103:                                                 // 98 / 99 on the stride-1 layers of the widest map
            if constexpr (BF16) {
                if (direct3_ok) return cfg & 1 ? launch_direct1x1<4, 8, 2, 2, false, true>(k, stream) : launch_direct1x1<8, 8, 2, 2, false, true>(k, stream);
            }
            return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
        case 110: case 111:                                                 // 70 / 61 with a three-deep weight ring behind counted waits
            if constexpr (BF16) {
                if (cfg == 110) return launch_halo<true, 192, 128, 4, 2, true, 16, 0, 4, 3>(k, stream);
                return launch_halo<true, 256, 128, 4, 2, true, 16, 0, 4, 3>(k, stream);
            }
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 112: return launch_halo<BF16, 128, 128, 2, 2, true, 16, 0, 2, 3>(k, stream);   // 68 with a three-deep ring
        case 117: case 118:                                                 // 256 x 256 tile, one workgroup per CU: four waves of 128 x 128 / eight of 64 x 128
            if constexpr (BF16) {
                if (k.Cout % 256 == 0) {
                    if (cfg == 117) return launch_halo<true, 256, 256, 2, 2, true, 16, 0, 1>(k, stream);
                    return launch_halo<true, 256, 256, 4, 2, true, 16, 0, 2>(k, stream);
                }
            }
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 120: case 121:
            if constexpr (BF16) {
                if (cfg == 120) return launch_halo<true, 512, 128, 8, 1, true, 16, 0, 2>(k, stream);   // 115 as one workgroup of eight waves per CU
                return launch_halo<true, 256, 128, 4, 1, true, 16, 3, 2>(k, stream);                  // 115 with the register epilogue
            }
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 119: return launch_halo<BF16, 256, 128, 2, 2, true, 16, 0, 2>(k, stream);   // four waves of 128 x 64
        case 116:                                                           // 115 as 128 x 128 workgroups of two waves (13 x 13: +6 % per step)
            if constexpr (BF16) return launch_halo<true, 128, 128, 2, 1, true, 16, 0, 2>(k, stream);
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 113: case 114:                                                 // 512 x 128 tile, 16 waves of 64 x 64, one workgroup per CU: half the weight stream per MFMA of 61
            if constexpr (BF16) {
                if (cfg == 113) return launch_halo<true, 512, 128, 8, 2, true, 16, 0, 4, 2>(k, stream);
                return launch_halo<true, 512, 128, 8, 2, true, 16, 0, 4, 3>(k, stream);
            }
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 78: return launch_halo<BF16, 128, 128, 4, 4, true, 16, 0, 1>(k, stream);    // 128 x 128 as 16 waves of 32 x 32 (small-batch slices)
        case 79: return launch_halo<BF16, 128, 128, 4, 4, false, 32, 1, 1>(k, stream);
        case 100: case 101:                                                 // 98 / 99 three K-steps ahead
            if constexpr (BF16) {
                if (direct3_ok) return cfg & 1 ? launch_direct1x1<4, 8, 3, 2, false, true>(k, stream) : launch_direct1x1<8, 8, 3, 2, false, true>(k, stream);
            }
            return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
        // 68 with a six- / four-deep weight ring behind counted waits: at batch 1 a 9-step K slice took 15.0 us either way (the
        // chain is launch + first-load latency + one wave per SIMD reading LDS and issuing MFMAs back to back, not the weight tiles)
        case 76: return launch_halo<BF16, 128, 128, 2, 2, true, 16, 0, 2, 6>(k, stream);
        case 77: return launch_halo<BF16, 128, 128, 2, 2, true, 16, 0, 2, 4>(k, stream);
        case 0: return launch_cfg<BF16, 128, 128, 2, 2, 2>(k, stream);     // 2 blocks/CU
        case 1: return launch_cfg<BF16, 256, 128, 4, 2, 3>(k, stream);     // 8 waves, 3-stage ring, 1 block/CU
        case 2: return launch_cfg<BF16, 128, 128, 2, 2, 3>(k, stream);     // 4 waves, 3-stage ring, 1 block/CU
        case 3:                                                            // BN = 256: packed weights / bias are padded to 128 rows only
            if (k.Cout % 256 == 0) return launch_cfg<BF16, 128, 256, 2, 4, 3>(k, stream);
            return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
        case 4: return launch_cfg<BF16, 256, 128, 4, 2, 2>(k, stream);
        case 6: return launch_cfg<BF16, 128, 128, 2, 4, 2>(k, stream);
        case 11: return launch_cfg<BF16, 256, 64, 4, 1, 2>(k, stream);
        case 20: return launch_halo<BF16, 256, 128, 4, 2>(k, stream);
        case 21: return launch_halo<BF16, 128, 128, 2, 2>(k, stream);
        case 25: return launch_halo<BF16, 128, 128, 2, 4>(k, stream);
        case 27: return launch_halo<BF16, 256, 128, 4, 4>(k, stream);
        case 50: return launch_halo<BF16, 256, 128, 4, 2, true, 32, 2, 4>(k, stream);
        case 51: return launch_halo<BF16, 256, 128, 4, 2, true, 16, 2, 4>(k, stream);   // the same out of 16x16x32 MFMAs
        case 52: return launch_halo<BF16, 256, 128, 4, 2, false, 32, 2, 2>(k, stream);  // two patch buffers, one workgroup per CU
        case 60: return launch_halo<BF16, 256, 128, 4, 2, true, 32, 0, 4>(k, stream);   // 50 / 51 with the wave-private epilogue
        case 62: return launch_halo<BF16, 128, 128, 4, 2, false, 32, 0, 1>(k, stream);  // 24 / 28 with the wave-private epilogue
        case 63: return launch_halo<BF16, 128, 128, 4, 2, true, 32, 0, 1>(k, stream);
        case 64: return launch_halo<BF16, 128, 128, 2, 2, true, 32, 0, 2>(k, stream);   // 4 waves of 64 x 64, up to four workgroups per CU
        case 70: case 71:                                                   // bf16 only (48 / 80 rows per wave need 16-row MFMA tiles)
            if constexpr (BF16) {
                if (cfg == 70) return launch_halo<true, 192, 128, 4, 2, true, 16, 0, 4>(k, stream);   // 48 x 64 per wave: 904 instead of 676 tiles at 26x26 / batch 64
                return launch_halo<true, 320, 128, 4, 2, true, 16, 0, 2>(k, stream);                  // 80 x 64 per wave (one workgroup per CU)
            }
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 40: case 41: case 42:                                          // 256 / 192 / 128 x 256 ping-pong tile, bf16 only
            if constexpr (BF16) {
                if (k.Cout % 256 == 0) return cfg == 40 ? launch_p8<4>(k, stream) : cfg == 41 ? launch_p8<3>(k, stream) : launch_p8<2>(k, stream);
            }
            return launch_halo<BF16, 128, 128, 4, 2>(k, stream);
        case 34: return launch_halo<BF16, 128, 128, 4, 2, false, 16>(k, stream);   // 24 / 28 out of 16x16x32 MFMAs
        case 38: return launch_halo<BF16, 128, 128, 4, 2, true, 16>(k, stream);
        case 80: if (plain) return launch_cfg<BF16, 128, 128, 4, 2, 2, 32, 0, 1>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 5 + wave epilogue
        case 82: if (plain) return launch_cfg<BF16, 256, 128, 4, 2, 2, 16, 0, 2>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 96 KB: 1 workgroup / CU
        case 83: if (plain) return launch_cfg<BF16, 256, 128, 4, 2, 3, 16, 0, 2>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 3-stage ring, 144 KB
        case 84: if (plain) return launch_cfg<BF16, 128, 128, 2, 2, 2, 16, 0, 2>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 4 waves of 64 x 64, 2 workgroups / CU
        case 85: if (plain) return launch_cfg<BF16, 128, 128, 2, 2, 3, 16, 0, 2>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 3-stage ring, 96 KB
        case 86: if (plain) return launch_cfg<BF16, 128, 128, 4, 2, 3, 16, 0, 1>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 8 waves, 3-stage ring
        case 87: if (plain && k.Cout % 256 == 0) return launch_cfg<BF16, 256, 256, 4, 4, 2, 16, 0, 1>(k, stream); return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);   // 16 waves of 64 x 64, 128 KB
        case 90: case 91: case 92: case 93: case 94: case 95:              // direct-A 1x1 kernel with the k-chunks permuted (adjacent pieces per lane), deeper prefetch
            if constexpr (BF16) {
                if (direct_ok) {
                    if (k.Cout <= 64) return cfg & 1 ? launch_direct1x1<4, 4, 3, 2>(k, stream) : launch_direct1x1<8, 4, 3, 2>(k, stream);
                    if (cfg == 90) return launch_direct1x1<8, 8, 2, 2>(k, stream);
                    if (cfg == 91) return launch_direct1x1<4, 8, 2, 2>(k, stream);
                    if (cfg == 92) return launch_direct1x1<8, 8, 3, 2>(k, stream);
                    if (cfg == 93) return launch_direct1x1<4, 8, 3, 2>(k, stream);
                    if (cfg == 94) return launch_direct1x1<8, 8, 4, 2>(k, stream);
                    return launch_direct1x1<4, 8, 4, 2>(k, stream);
                }
            }
            return launch_cfg<BF16, 128, 128, 4, 2, 2>(k, stream);
#endif
        default: break;
    }
    set_error("conv: unknown tile configuration %d (experimental tilings need the XP build)", cfg);
    return YOLO_ERR_ARG;
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
    k.leaky = p.leaky; k.upsample = p.upsample; k.out_f32 = p.out_f32; k.reorg = p.reorg;
    if (p.reorg) YOLO_REQUIRE(p.Ho % 2 == 0 && p.Wo % 2 == 0 && !p.upsample && !p.out_f32 && !p.res, YOLO_ERR_SHAPE, "conv: reorg needs even output size");
    const long long M = (long long)p.B * p.Ho * p.Wo;
    YOLO_REQUIRE(M > 0 && M < (1ll << 31) - 512, YOLO_ERR_SHAPE, "conv: M=%lld out of range", M);
    k.M = (int)M; k.ntm = k.ntn = 0;
    k.splitk = 1;
    k.partial = (p.small_plan && p.scratch && ((uintptr_t)p.scratch & 15) == 0) ? (float*)p.scratch : nullptr;
    k.partial_cap = (long long)(p.scratch_bytes / 4);
    return dtype == YOLO_DTYPE_BF16 ? launch_dtype<true>(k, stream) : launch_dtype<false>(k, stream);
}

}  // namespace yolo
