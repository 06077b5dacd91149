// Probe: per-CU rate of LDS-DMA (global_load_lds_dwordx4) vs register staging, L2-resident source.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dma_rate scripts/probe/dma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)

template <int MODE>   // 0: DMA, 1: global_load + ds_write_b128, 2: global_load only (accumulate)
__global__ void k(const char* src, int iters, unsigned long long* cyc, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)(blockIdx.x & 63) * 65536 + wave * 4096;   // 64 KB window per block, L2 resident
    char* l = lds + wave * 4096;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* g = base + ((it * 4 + j) & 3) * 1024 + lane * 16 + ((it & 3) * 16384);
            if (MODE == 0) GLDS16(g, l + j * 1024);
            else {
                u32x4 v = *(const u32x4*)g;
                if (MODE == 1) *(u32x4*)(l + j * 1024 + lane * 16) = v;
                else acc ^= v;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE != 0 && acc[0] == 0x12345 && lds[threadIdx.x] == 1) sink[0] = acc[1];
}

template <int MODE>
void run(const char* name, const char* src, int nw, unsigned long long* dcyc, unsigned* sink) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<MODE><<<grid, nw * 64, 160 * 1024>>>(src, 100, dcyc, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<grid, nw * 64, 160 * 1024>>>(src, iters, dcyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    const double pieces = (double)iters * 4 * nw;
    printf("%-28s waves/CU %2d : %8.1f cycles/piece/CU  (%6.1f cycles per piece per wave)  %7.1f B/clk/CU  wall %.3f ms -> %.2f TB/s chip\n",
           name, nw, avg / pieces, avg / (iters * 4.0), 1024.0 * pieces / avg, ms, 1024.0 * pieces * grid / (ms * 1e-3) / 1e12);
}

// MODE 3: waves 0..nd-1 issue LDS-DMA; the other waves run a ds_read_b128 loop (16 reads, wait, repeat)
__global__ void kmix(const char* src, int iters, int nd, unsigned long long* cyc, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    if (wave < nd) {
        const char* base = src + (size_t)(blockIdx.x & 63) * 65536 + wave * 4096;
        char* l = lds + wave * 4096;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) GLDS16(base + ((it * 4 + j) & 3) * 1024 + lane * 16 + ((it & 3) * 16384), l + j * 1024);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
    } else {
        u32x4 acc = {0, 0, 0, 0};
        const char* l = lds + 65536 + (wave & 7) * 8192 + lane * 16;
        for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= *(volatile const u32x4*)(l + j * 1024);
        }
        if (acc[0] == 0x12345) sink[0] = acc[1];
    }
}

void runmix(const char* src, int nd, int nr, unsigned long long* dcyc, unsigned* sink) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)kmix, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    kmix<<<grid, (nd + nr) * 64, 160 * 1024>>>(src, 100, nd, dcyc, sink);
    kmix<<<grid, (nd + nr) * 64, 160 * 1024>>>(src, iters, nd, dcyc, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("DMA waves %d + ds_read waves %2d : %6.1f cycles per piece per DMA wave, %6.1f cycles/piece/CU\n", nd, nr, avg / (iters * 4.0), avg / (iters * 4.0 * nd));
}

int main() {
    char* src; hipMalloc(&src, 64 * 65536 + 65536); hipMemset(src, 1, 64 * 65536 + 65536);
    unsigned long long* dcyc; hipMalloc(&dcyc, 256 * 8);
    unsigned* sink; hipMalloc(&sink, 4);
    for (int nw : {1, 2, 4, 8, 16}) run<0>("LDS-DMA dwordx4", src, nw, dcyc, sink);
    for (int nw : {1, 2, 4, 8, 16}) run<1>("global_load + ds_write_b128", src, nw, dcyc, sink);
    for (int nw : {1, 2, 4, 8, 16}) run<2>("global_load only", src, nw, dcyc, sink);
    for (int nr : {0, 4, 8}) runmix(src, 4, nr, dcyc, sink);
    for (int nr : {0, 4, 8}) runmix(src, 2, nr, dcyc, sink);
    return 0;
}
