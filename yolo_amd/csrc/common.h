// Shared host-side helpers for libyolo_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/yolo_hip.h"

namespace yolo {

void set_error(const char* fmt, ...);

#define YOLO_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            yolo::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return YOLO_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define YOLO_REQUIRE(cond, code, ...)       \
    do {                                    \
        if (!(cond)) {                      \
            yolo::set_error(__VA_ARGS__);   \
            return (code);                  \
        }                                   \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// leaky_relu of the reference (yolo_v3/model.py:8-12): f1*x + f2*|x| with f1 = 0.55, f2 = 0.45.  Written as ONE
// explicit fma so that every kernel (and the fused / unfused variants of a layer) rounds it the same way,
// whatever the compiler would contract.
__device__ __forceinline__ float leaky_v3(float v) { return __builtin_fmaf(0.45f, __builtin_fabsf(v), 0.55f * v); }

// ---- conv layer launch description (device pointers) ------------------------------------
struct ConvLaunch {
    const void* in;        // activations, element type = net dtype (stem: always f32)
    const void* w;         // packed weights [Npad][Kpad] (k = (ky,kx,ci)), BN folded
    const float* bias;     // [Npad] folded bias
    void* out;
    const void* res;       // residual or nullptr (same dtype/geometry as out)
    const void* zero;      // >= 16 zero bytes (source of padded taps)
    int B, H, W, Cin;      // input geometry
    int Ho, Wo, Cout;      // conv output geometry (before the optional 2x upsample scatter)
    int ksize, stride, pad;
    int in_ld, in_coff;    // elements per input pixel row / first channel
    int out_ld, out_coff;
    int res_ld, res_coff;
    int kpad_bytes;        // bytes per packed weight row (multiple of 128)
    int leaky;             // 0: linear; 1: 0.55x + 0.45|x| (v3, model.py:8-12); 2: max(0.1x, x) (v2, yolo_v2.py:86)
    int upsample;          // 1: write every output pixel to the 2x2 block of a [B,2Ho,2Wo] tensor
    int out_f32;           // 1: out is float32 regardless of dtype (detection heads), scalar stores
    int reorg;             // 1: space-to-depth scatter (yolo_v2.py:97-113): pixel (y,x) -> (y/2,x/2), channel block (y&1)*2+(x&1)
    // small-batch plan (yolo_net_set_small_batch_plan): the K loop of a layer whose grid would leave most CUs idle is cut into
    // `S` slices run by S times as many workgroups; f32 partial tiles go to `scratch` and a second launch adds them in slice order
    int small_plan;        // 1: the launcher may split K
    void* scratch;         // >= S * M * Cout * 4 bytes when small_plan
    size_t scratch_bytes;
};

// Measurement aid: when `record_kernel_names` is set (only yolo_net_forward_timed sets it), every launcher
// writes the name of the kernel instantiation it launches (as rocprofv3 prints it, without return type and
// arguments) with note_kernel(); last_kernel_name() returns it.
extern thread_local bool record_kernel_names;
void note_kernel(const char* fmt, ...);
const char* last_kernel_name();

int launch_conv_igemm(int dtype, const ConvLaunch& p, hipStream_t stream);
int launch_stem(int dtype, const ConvLaunch& p, hipStream_t stream);   // Cin = 3, 3x3 s1, f32 in
// conv 3->32 (3x3 s1) + conv 32->64 (3x3 s2) fused, bf16 only (front.hip)
int launch_front(const ConvLaunch& stem, const ConvLaunch& down, hipStream_t stream);
// res_block(x, 32): conv 64->32 (1x1) + conv 32->64 (3x3) + shortcut fused, bf16 only (resblock.hip)
// Stem conv (3 -> 32, K = 27 padded to 32) on two v_mfma_f32_32x32x16_bf16: which im2col tap (ky * 9 + kx * 3 + c, -1 = zero
// pad) a lane's operand element j of MFMA m holds, for the lane half fh = lane >> 5.  Chosen so that three of the four
// (m, fh) groups are 8 CONTIGUOUS values of one image row (one 16-byte read in front.hip): rows 0, 1, 2 of the window give
// their first eight values to (0,0), (1,0), (0,1); the ninth value of each row goes to (1,1).  stem.hip and front.hip use the
// same assignment, so the fused and the separate kernels add the products in the same order (bit-identical results).
__host__ __device__ inline int stem_slot_tap(int m, int fh, int j) {
    return fh == 0 ? (m == 0 ? j : 9 + j) : (m == 0 ? 18 + j : (j < 3 ? 9 * j + 8 : -1));
}
int launch_res_block(const ConvLaunch& c1, const ConvLaunch& c3, hipStream_t stream);
int launch_stem_pool(const ConvLaunch& c0, hipStream_t stream);          // front.hip: stem conv + 2x2 max-pool out of an LDS image patch, bf16
int launch_conv3x3_c32_pool(const ConvLaunch& c3, hipStream_t stream);   // resblock.hip: conv 32->64 3x3 + 2x2 max-pool, bf16
int launch_maxpool2x2(int dtype, const void* in, int B, int H, int W, int C, void* out, hipStream_t stream);
int launch_cast_f32_to_bf16(const float* in, void* out, size_t n, hipStream_t stream);
int launch_cast_bf16_to_f32(const void* in, float* out, size_t n, hipStream_t stream);

}  // namespace yolo
