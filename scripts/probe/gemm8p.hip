// Probe (not part of the library): how fast does a 256x256-tile, 8-wave, ping-pong bf16 GEMM core run on this box?
// C[M,N] f32 = A[M,K] * B[N,K]^T, bf16 operands, K contiguous for both (the layout of the conv kernels: pixel rows and
// packed weight rows).  Structure after cdna_hip_programming.md section 5 ("8-phase" template), with a schedule of its own:
//   * 8 waves = 2 (M) x 4 (N), 128x64 outputs per wave as 8x4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator VGPRs)
//   * K-step 64 (128 B per row); LDS = 2 buffers x (A 256 rows + B 256 rows) x 128 B = 128 KB, one workgroup per CU
//   * four phases per K-step, 16 MFMAs each (one 64x32 quadrant x K=64); fragment reads 12 / 4 / 8 / 0 per phase
//   * the two wave rows run half a phase apart (two barriers per phase): one row's MFMAs cover the other row's
//     ds_reads and LDS-DMA issue
//   * the next K-step is staged by LDS-DMA during phases 0-2 (2 / 4 / 2 loads per wave) and waited for with vmcnt(0) at
//     the end of phase 3, one phase after its last load was issued
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm8p gemm8p.hip ; run: ./gemm8p [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define GLDS16(gptr, ldsptr)                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

constexpr int BM = 256, BN = 256;
constexpr int HALF_BYTES = 128 * 128;          // 128 rows x 128 B
constexpr int BUF_BYTES = 4 * HALF_BYTES;      // A0 A1 B0 B1

__global__ __launch_bounds__(512) void gemm8p_kernel(const char* __restrict__ A, const char* __restrict__ B, float* __restrict__ C,
                                                     int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ntn = N / BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int KB = K * 2;                       // bytes per row
    const int nk = KB / 128;

    // ---- loader: half-tile h (0,1: A rows 0-127 / 128-255; 2,3: B) = 16 pieces of 8 rows; this wave loads pieces wave, wave+8
    const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const char* gsrc[4];
    {
        const int row = 8 * wave + (lane >> 3);
        gsrc[0] = A + (long long)(m0 + row) * KB + chunk * 16;
        gsrc[1] = A + (long long)(m0 + 128 + row) * KB + chunk * 16;
        gsrc[2] = B + (long long)(n0 + row) * KB + chunk * 16;
        gsrc[3] = B + (long long)(n0 + 128 + row) * KB + chunk * 16;
    }
    const long long piece8 = 64ll * KB;         // 8 pieces = 64 rows further
    auto stage_half = [&](int h, int kt, int buf) {
        const char* src = gsrc[h] + (long long)kt * 128;
        char* dst = lds + buf * BUF_BYTES + h * HALF_BYTES + wave * 1024;
        GLDS16(src, dst);
        GLDS16(src + piece8, dst + 8 * 1024);
    };

    // ---- fragment addressing (16x16x32: lane row = l & 15, k-chunk = 4*kk + (l >> 4))
    const int fr = lane & 15, fq = lane >> 4;
    unsigned a_off[2][4], b_off[2][2];           // byte offset inside a buffer, kk = 0; kk = 1 is ^ 64
#pragma unroll
    for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wr * 128 + mq * 64 + i * 16 + fr;
            a_off[mq][i] = row * 128 + ((fq ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wc * 64 + nq * 32 + j * 16 + fr;
            b_off[nq][j] = 2 * HALF_BYTES + row * 128 + ((fq ^ ((row >> 1) & 7)) << 4);
        }

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 af[4][2], bf0[2][2], bf1[2][2];       // A sub-tile (64 rows), B sub-tiles nq = 0 / 1 (32 columns each), [.][kk]

    auto read_a = [&](const char* base, int mq) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const u32x4*)(base + a_off[mq][i]);
            af[i][1] = *(const u32x4*)(base + (a_off[mq][i] ^ 64));
        }
    };
    auto read_b = [&](const char* base, int nq, u32x4 (&bf)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf[j][0] = *(const u32x4*)(base + b_off[nq][j]);
            bf[j][1] = *(const u32x4*)(base + (b_off[nq][j] ^ 64));
        }
    };
    auto mfma_quadrant = [&](int mq, int nq, u32x4 (&bf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mq][i][nq][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, af[i][kk]), __builtin_bit_cast(bf16x8, bf[j][kk]), acc[mq][i][nq][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define BAR()                               \
    do {                                    \
        __builtin_amdgcn_sched_barrier(0);  \
        __builtin_amdgcn_s_barrier();       \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)
#define LGKM0()                                              \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
#define VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    // ---- prologue: K-step 0 into buffer 0
#pragma unroll
    for (int h = 0; h < 4; ++h) stage_half(h, 0, 0);
    VM0();
    BAR();
    if (wr == 1) BAR();                          // the second wave row runs one barrier behind

    for (int kt = 0; kt < nk; ++kt) {
        const char* base = lds + (kt & 1) * BUF_BYTES;
        const int nb = (kt & 1) ^ 1;
        const bool more = kt + 1 < nk;
        // phase 0: quadrant (0,0)
        read_b(base, 0, bf0);
        read_a(base, 0);
        if (more) stage_half(0, kt + 1, nb);
        BAR();
        LGKM0();
        mfma_quadrant(0, 0, bf0);
        BAR();
        // phase 1: quadrant (0,1)
        read_b(base, 1, bf1);
        if (more) { stage_half(1, kt + 1, nb); stage_half(2, kt + 1, nb); }
        BAR();
        LGKM0();
        mfma_quadrant(0, 1, bf1);
        BAR();
        // phase 2: quadrant (1,1)
        read_a(base, 1);
        if (more) stage_half(3, kt + 1, nb);
        BAR();
        LGKM0();
        mfma_quadrant(1, 1, bf1);
        BAR();
        // phase 3: quadrant (1,0), operands already in registers; the next K-step must have landed before the barrier
        // that lets the other wave row read it
        if (wr == 1) VM0();
        BAR();
        mfma_quadrant(1, 0, bf0);
        if (wr == 0) VM0();
        BAR();
    }
    if (wr == 0) BAR();                          // even out the barrier count

    // ---- epilogue: C/D layout of 16x16: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
    for (int mq = 0; mq < 2; ++mq)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wc * 64 + nq * 32 + j * 16 + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + wr * 128 + mq * 64 + i * 16 + 4 * fq + r;
                        C[(long long)row * N + col] = acc[mq][i][nq][j][r];
                    }
                }
}

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    int M = 4096, N = 4096, K = 4096;
    if (argc > 3) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    if (M % 256 || N % 256 || K % 64) { printf("M, N multiples of 256, K of 64\n"); return 1; }
    std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    char *dA, *dB;
    float* dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    const int lds_bytes = 2 * BUF_BYTES;
    CK(hipFuncSetAttribute((const void*)gemm8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const int grid = (M / BM) * (N / BN);
    hipLaunchKernelGGL(gemm8p_kernel, dim3(grid), dim3(512), lds_bytes, 0, dA, dB, dC, M, N, K);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double max_err = 0;
    unsigned t = 777;
    for (int it = 0; it < 4000; ++it) {
        t = t * 1664525u + 1013904223u;
        const int r = (t >> 8) % M;
        t = t * 1664525u + 1013904223u;
        const int c = (t >> 8) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
        const double err = fabs(ref - hC[(size_t)r * N + c]) / (1.0 + fabs(ref));
        if (err > max_err) max_err = err;
    }
    printf("check: max rel err over 4000 samples %.3g  %s\n", max_err, max_err < 2e-3 ? "OK" : "WRONG");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 20;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm8p_kernel, dim3(grid), dim3(512), lds_bytes, 0, dA, dB, dC, M, N, K);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("%d x %d x %d: %.4f ms  %.1f TFLOP/s (grid %d)\n", M, N, K, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, grid);
    }
    return 0;
}
