// Where does the dispatcher put the workgroups of one launch?  512 workgroups of 512 threads with 72 KB of LDS each (two fit
// a CU), every workgroup records (XCC, SE, CU) and its start time; the host prints which workgroup ids share a CU.
// hipcc --offload-arch=gfx950 -O2 scripts/probe/wg_placement.hip -o scripts/probe/wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)t0;
    }
    volatile char* l = lds;
    for (int i = 0; i < spin; ++i) l[threadIdx.x] = (char)i;      // keep the workgroup resident for a while
    if (threadIdx.x == 0) out[blockIdx.x * 4 + 3] = (unsigned)__builtin_amdgcn_s_memrealtime();
}
int main() {
    const int G = 700;
    unsigned* d;
    hipMalloc(&d, G * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(G), dim3(512), 72 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 4);
    hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    unsigned tmin = ~0u;
    for (int b = 0; b < G; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 4], x = h[b * 4 + 1] & 15;
        const unsigned key = (x << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15);
        cu[key].push_back(b);
    }
    printf("%zu distinct (xcc, se, cu); workgroup ids (start tick) per CU:\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ % 8 == 0 || kv.second.size() != 2) {
            printf("xcc %u se %u cu %2u :", kv.first >> 16, (kv.first >> 8) & 7, kv.first & 15);
            for (int b : kv.second) printf("  %d (t=%u..%u)", b, h[b * 4 + 2] - tmin, h[b * 4 + 3] - tmin);
            printf("\n");
        }
    }
    return 0;
}
