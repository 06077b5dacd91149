// Host side of libyolo_hip.so: weight ingestion (Darknet order, BN folding, K-major packing),
// the static execution plan of yolov3.forward (reference yolo_v3/model.py:21-67 +
// yolo_v3/utils/layer_utils.py:23-72) and the C-ABI entry points for the conv stack.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

#include "common.h"

namespace yolo {

static thread_local char g_err[512] = "";

thread_local bool record_kernel_names = false;
static thread_local char g_kernel_name[128] = "";
void note_kernel(const char* fmt, ...) {
    if (!record_kernel_names) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel_name, sizeof(g_kernel_name), fmt, ap);
    va_end(ap);
}
const char* last_kernel_name() { return g_kernel_name; }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// ---- one conv layer's device-resident parameters -------------------------------------------------
struct ConvParams {
    int cin = 0, cout = 0, k = 0, stride = 1, bn = 1;
    int act = 1;                // ConvLaunch::leaky mode applied when bn != 0
    int dtype = YOLO_DTYPE_F32;
    bool stem = false;          // Cin == 3 direct kernel, weights packed [27][Cout] f32
    int kpad_bytes = 0, npad = 0;
    void* w = nullptr;
    float* bias = nullptr;
    ~ConvParams() {
        if (w) (void)hipFree(w);
        if (bias) (void)hipFree(bias);
    }
};

// w_at(n, ci, ky, kx) abstracts the host layout (Darknet OIHW stream or TF HWIO variable).
// Fold inference BN into per-channel (scale, bias).  v3 (model.py:24-30, eps 1e-5):
// y = (x - mean) * gamma / sqrt(var + eps) + beta.
static void fold_bn(int co, int bn, const float* p0, const float* p1, const float* p2, const float* p3,
                    std::vector<float>& scale, std::vector<float>& bias) {
    scale.resize(co);
    bias.resize(co);
    for (int n = 0; n < co; ++n) {
        if (bn) {
            const float beta = p0[n], gamma = p1[n], mean = p2[n], var = p3[n];
            const float s = gamma / std::sqrt(var + 1e-5f);
            scale[n] = s;
            bias[n] = beta - mean * s;
        } else {
            scale[n] = 1.0f;
            bias[n] = p0[n];
        }
    }
}

// w_at(n, ci, ky, kx) abstracts the host layout (Darknet OIHW stream or TF HWIO variable).
template <class WAt>
static int pack_conv_sb(ConvParams& L, WAt w_at, const std::vector<float>& scale, const std::vector<float>& bias) {
    const int co = L.cout, ci = L.cin, k = L.k;
    L.stem = (ci == 3);
    const int sz = L.dtype == YOLO_DTYPE_BF16 ? 2 : 4;
    size_t wbytes;
    std::vector<char> host;
    if (L.stem && sz == 2) {
        // MFMA stem: [Cout][K pad 32] bf16, k = (ky*3+kx)*3 + c
        L.npad = co;
        L.kpad_bytes = 64;
        wbytes = (size_t)co * 64;
        host.assign(wbytes, 0);
        for (int n = 0; n < co; ++n)
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx)
                    for (int c = 0; c < ci; ++c)
                        ((uint16_t*)host.data())[n * 32 + (ky * k + kx) * ci + c] = f32_to_bf16_rne(w_at(n, c, ky, kx) * scale[n]);
    } else if (L.stem) {
        L.npad = co;
        L.kpad_bytes = 0;
        wbytes = (size_t)k * k * ci * co * 4;
        host.assign(wbytes, 0);
        float* hw = (float*)host.data();
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
                for (int c = 0; c < ci; ++c)
                    for (int n = 0; n < co; ++n) hw[((ky * k + kx) * ci + c) * co + n] = w_at(n, c, ky, kx) * scale[n];
    } else {
        L.npad = (int)align_up((size_t)co, 128);
        L.kpad_bytes = (int)align_up((size_t)k * k * ci * sz, 128);
        wbytes = (size_t)L.npad * L.kpad_bytes;
        host.assign(wbytes, 0);
        for (int n = 0; n < co; ++n) {
            char* rowp = host.data() + (size_t)n * L.kpad_bytes;
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx)
                    for (int c = 0; c < ci; ++c) {
                        const float v = w_at(n, c, ky, kx) * scale[n];
                        const size_t kidx = (size_t)(ky * k + kx) * ci + c;
                        if (sz == 2) ((uint16_t*)rowp)[kidx] = f32_to_bf16_rne(v);
                        else ((float*)rowp)[kidx] = v;
                    }
        }
    }
    std::vector<float> bias_pad(L.npad, 0.f);
    memcpy(bias_pad.data(), bias.data(), (size_t)co * 4);
    YOLO_HIP_CHECK(hipMalloc(&L.w, wbytes));
    YOLO_HIP_CHECK(hipMalloc((void**)&L.bias, (size_t)L.npad * 4));
    YOLO_HIP_CHECK(hipMemcpy(L.w, host.data(), wbytes, hipMemcpyHostToDevice));
    YOLO_HIP_CHECK(hipMemcpy(L.bias, bias_pad.data(), (size_t)L.npad * 4, hipMemcpyHostToDevice));
    return YOLO_OK;
}

template <class WAt>
static int pack_conv(ConvParams& L, WAt w_at, const float* p0, const float* p1, const float* p2, const float* p3) {
    std::vector<float> scale, bias;
    fold_bn(L.cout, L.bn, p0, p1, p2, p3, scale, bias);
    return pack_conv_sb(L, w_at, scale, bias);
}

static void fill_conv(const ConvParams& L, ConvLaunch& p) {
    p.w = L.w;
    p.bias = L.bias;
    p.Cin = L.cin;
    p.Cout = L.cout;
    p.ksize = L.k;
    p.stride = L.stride;
    p.pad = (L.k - 1) / 2;
    p.kpad_bytes = L.kpad_bytes;
    p.Ho = (p.H + 2 * p.pad - L.k) / L.stride + 1;
    p.Wo = (p.W + 2 * p.pad - L.k) / L.stride + 1;
    p.leaky = L.bn ? L.act : 0;
}

// conv2d(conv2d(x, 32, 3), 64, 3, strides=2) of darknet53_body (layer_utils.py:31-32) as one kernel
static int run_front(const ConvParams& L0, const ConvParams& L1, ConvLaunch p0, ConvLaunch p1, hipStream_t stream) {
    fill_conv(L0, p0);
    fill_conv(L1, p1);
    return launch_front(p0, p1, stream);
}

static int run_conv(const ConvParams& L, ConvLaunch p, hipStream_t stream) {
    p.w = L.w;
    p.bias = L.bias;
    p.Cin = L.cin;
    p.Cout = L.cout;
    p.ksize = L.k;
    p.stride = L.stride;
    p.pad = (L.k - 1) / 2;          // layer_utils.py:11-20: SAME (s1) / explicit symmetric pad + VALID (s2)
    p.kpad_bytes = L.kpad_bytes;
    p.Ho = (p.H + 2 * p.pad - L.k) / L.stride + 1;
    p.Wo = (p.W + 2 * p.pad - L.k) / L.stride + 1;
    p.leaky = L.bn ? L.act : 0;     // model.py:32-36 vs :42-44 (v2: yolo_v2.py:86 vs :88)
    return L.stem ? launch_stem(L.dtype, p, stream) : launch_conv_igemm(L.dtype, p, stream);
}

}  // namespace yolo

using namespace yolo;

// ---- architecture -------------------------------------------------------------------------------
struct Spec { int cin, cout, k, stride, bn; };

static std::vector<Spec> yolov3_specs(int class_num) {
    std::vector<Spec> v;
    auto conv = [&](int ci, int co, int k, int s = 1, int bn = 1) { v.push_back({ci, co, k, s, bn}); return co; };
    auto res = [&](int c, int f) { conv(c, f, 1); conv(f, 2 * f, 3); return 2 * f; };      // layer_utils.py:24-29
    int c = conv(3, 32, 3);                                                               // :31
    c = conv(c, 64, 3, 2);                                                                // :32
    c = res(c, 32);
    c = conv(c, 128, 3, 2);
    for (int i = 0; i < 2; ++i) c = res(c, 64);
    c = conv(c, 256, 3, 2);
    for (int i = 0; i < 8; ++i) c = res(c, 128);
    c = conv(c, 512, 3, 2);
    for (int i = 0; i < 8; ++i) c = res(c, 256);
    c = conv(c, 1024, 3, 2);
    for (int i = 0; i < 4; ++i) c = res(c, 512);
    auto block = [&](int ci, int f) {                                                     // layer_utils.py:56-65
        conv(ci, f, 1); conv(f, 2 * f, 3); conv(2 * f, f, 1); conv(f, 2 * f, 3); conv(2 * f, f, 1); conv(f, 2 * f, 3);
    };
    const int nout = 3 * (5 + class_num);
    block(1024, 512); conv(1024, nout, 1, 1, 0); conv(512, 256, 1);                        // model.py:41-47
    block(768, 256); conv(512, nout, 1, 1, 0); conv(256, 128, 1);                          // model.py:49-57
    block(384, 128); conv(256, nout, 1, 1, 0);                                             // model.py:59-64
    return v;
}

// One step of the static plan.  Tensor ids index Plan::tens; -1 = none, -2/-3/-4 = fm1/fm2/fm3, -5 = image.
struct Step {
    int layer;                 // conv index, or -1 for a 2x2 max-pool step
    int in, out, res;
    int in_coff, out_coff;
    int upsample;
    int reorg;
};
struct Tens { int h, w, ld; size_t bytes; int first, last; size_t offset; };
struct Plan { std::vector<Step> steps; std::vector<Tens> tens; size_t ws_bytes = 0; size_t scratch_off = 0, scratch_bytes = 0; };
// split-K partial tiles of the small-batch plan: at most 512 workgroups x 128 x 128 f32 per launch (conv_igemm.hip pick_splitk)
constexpr size_t kSplitKScratch = (size_t)512 * 128 * 128 * 4;

struct YoloNet {
    int kind = 3;              // 3: YOLOv3 (model.py), 2: YOLOv2 (yolo_v2.py)
    int class_num = 0, dtype = 0, box_per_cell = 5;
    bool small_plan = false;   // yolo_net_set_small_batch_plan
    std::vector<std::string> kernel_names;   // per layer, filled by yolo_net_forward_timed
    std::vector<Spec> specs;
    std::vector<std::unique_ptr<ConvParams>> layers;
    std::map<std::tuple<int, int, int>, Plan> plans;
    std::mutex mu;
    // lanes (forward_lanes): side streams + fork/join events, created on first use on the device of that call
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> side_done;
    hipEvent_t fork_ev = nullptr;
    ~YoloNet() {
        for (hipStream_t s : side) (void)hipStreamDestroy(s);
        for (hipEvent_t e : side_done) (void)hipEventDestroy(e);
        if (fork_ev) (void)hipEventDestroy(fork_ev);
    }
};
struct YoloConv { ConvParams p; };

static void assign_offsets(Plan& P) {
    // lifetimes -> offsets (first-fit over live intervals)
    for (size_t s = 0; s < P.steps.size(); ++s) {
        const Step& st = P.steps[s];
        for (int id : {st.in, st.out, st.res})
            if (id >= 0) {
                if (P.tens[id].first < 0) P.tens[id].first = (int)s;
                P.tens[id].last = (int)s;
            }
    }
    std::vector<int> placed;
    for (size_t s = 0; s < P.steps.size(); ++s)
        for (size_t id = 0; id < P.tens.size(); ++id) {
            Tens& t = P.tens[id];
            if (t.first != (int)s) continue;
            // concat buffers are created early (first written mid-backbone); plain first-fit
            size_t off = 0;
            bool moved = true;
            while (moved) {
                moved = false;
                for (int o : placed) {
                    const Tens& u = P.tens[o];
                    if (u.last < t.first || u.first > t.last) continue;          // disjoint lifetimes
                    if (off < u.offset + u.bytes && u.offset < off + t.bytes) { off = u.offset + u.bytes; moved = true; }
                }
            }
            t.offset = off;
            placed.push_back((int)id);
            if (off + t.bytes > P.ws_bytes) P.ws_bytes = off + t.bytes;
        }
}

// yolo_v2.build_networks (reference yolo_v2/yolo/yolo_v2.py:32-70): 23 convs, 5 max-pools, the passthrough
// 1x1 conv scattered space-to-depth into channels [0:256) of the concat buffer whose channels [256:1280)
// are written by 24_conv.
static std::vector<Spec> yolov2_specs(int num_class, int box_per_cell) {
    std::vector<Spec> v;
    const int s[22][3] = {{3, 3, 32}, {3, 32, 64}, {3, 64, 128}, {1, 128, 64}, {3, 64, 128}, {3, 128, 256}, {1, 256, 128},
                          {3, 128, 256}, {3, 256, 512}, {1, 512, 256}, {3, 256, 512}, {1, 512, 256}, {3, 256, 512},
                          {3, 512, 1024}, {1, 1024, 512}, {3, 512, 1024}, {1, 1024, 512}, {3, 512, 1024}, {3, 1024, 1024},
                          {3, 1024, 1024}, {1, 512, 64}, {3, 1280, 1024}};
    for (auto& e : s) v.push_back({e[1], e[2], e[0], 1, 1});
    v.push_back({1024, box_per_cell * (num_class + 5), 1, 1, 0});
    return v;
}

static Plan build_plan_v2(const YoloNet& net, int B, int H, int W) {
    Plan P;
    const int sz = net.dtype == YOLO_DTYPE_BF16 ? 2 : 4;
    auto tensor = [&](int h, int w, int ld) {
        P.tens.push_back({h, w, ld, align_up((size_t)B * h * w * ld * sz, 256), -1, -1, 0});
        return (int)P.tens.size() - 1;
    };
    int li = 0, h = H, w = W;
    auto conv = [&](int in, int out = -1, int out_coff = 0, int reorg = 0) {
        const Spec& s = net.specs[li];
        if (out == -1) out = tensor(h, w, s.cout);
        P.steps.push_back({li, in, out, -1, 0, out_coff, 0, reorg});
        ++li;
        return out;
    };
    auto pool = [&](int in) {
        const int c = P.tens[in].ld;
        h /= 2; w /= 2;
        const int out = tensor(h, w, c);
        P.steps.push_back({-1, in, out, -1, 0, 0, 0, 0});
        return out;
    };
    int x = pool(conv(-5));                      // 0_conv, 1_pool
    x = pool(conv(x));                           // 2_conv, 3_pool
    x = conv(conv(conv(x)));                     // 4..6
    x = pool(x);                                 // 7_pool
    x = conv(conv(conv(x)));                     // 8..10
    x = pool(x);                                 // 11_pool
    x = conv(conv(conv(conv(x))));               // 12..15
    const int net16 = conv(x);                   // 16_conv
    const int h16 = h, w16 = w;
    x = pool(net16);                             // 17_pool
    for (int i = 0; i < 6; ++i) x = conv(x);     // 18..23
    const int cat = tensor(h, w, 256 + 1024);    // [reorg(26_conv) | net24]   yolo_v2.py:64-65
    conv(x, cat, 256);                           // 24_conv -> channels [256:1280)
    const int hs = h, ws = w;
    h = h16; w = w16;
    conv(net16, cat, 0, 1);                      // 26_conv + reorg -> channels [0:256)
    h = hs; w = ws;
    x = conv(cat);                               // 29_conv
    conv(x, -2);                                 // 30_conv -> logits (f32, bias only)
    assign_offsets(P);
    return P;
}

static Plan build_plan(const YoloNet& net, int B, int H, int W) {
    Plan P;
    const int sz = net.dtype == YOLO_DTYPE_BF16 ? 2 : 4;
    auto tensor = [&](int h, int w, int ld) {
        P.tens.push_back({h, w, ld, align_up((size_t)B * h * w * ld * sz, 256), -1, -1, 0});
        return (int)P.tens.size() - 1;
    };
    int li = 0;
    // conv into a fresh tensor (or an existing one at a channel offset)
    auto conv = [&](int in, int in_coff, int h_in, int w_in, int res = -1, int out = -1, int out_coff = 0, int ups = 0) {
        const Spec& s = net.specs[li];
        const int ho = h_in / s.stride, wo = w_in / s.stride;
        if (out == -1) out = tensor(ups ? 2 * ho : ho, ups ? 2 * wo : wo, s.cout);
        P.steps.push_back({li, in, out, res, in_coff, out_coff, ups, 0});
        ++li;
        return out;
    };
    int h = H, w = W;
    int x = conv(-5, 0, h, w);                               // stem
    auto down = [&](int in, int coff) { int o = conv(in, coff, h, w); h /= 2; w /= 2; return o; };
    auto res_block = [&](int in, int coff, int out = -1, int out_coff = 0) {
        int t = conv(in, coff, h, w);
        return conv(t, 0, h, w, in, out, out_coff);      // residual = block input (always a dense tensor here)
    };
    x = down(x, 0);
    x = res_block(x, 0);
    x = down(x, 0);
    for (int i = 0; i < 2; ++i) x = res_block(x, 0);
    x = down(x, 0);
    const int g3h = h, g3w = w;                              // stride-8 grid
    const int concat2 = tensor(g3h, g3w, 128 + 256);         // [upsampled inter2 | route_1]   model.py:59
    for (int i = 0; i < 7; ++i) x = res_block(x, 0);
    x = res_block(x, 0, concat2, 128);                       // route_1 lives inside concat2
    int xc = 128;
    x = down(x, xc);
    const int g2h = h, g2w = w;
    const int concat1 = tensor(g2h, g2w, 256 + 512);         // [upsampled inter1 | route_2]   model.py:49
    for (int i = 0; i < 7; ++i) x = res_block(x, 0);
    x = res_block(x, 0, concat1, 256);                       // route_2 lives inside concat1
    xc = 256;
    x = down(x, xc);
    for (int i = 0; i < 4; ++i) x = res_block(x, 0);         // route_3
    auto yolo_block = [&](int in, int& route) {              // layer_utils.py:56-65
        int t = in;
        for (int i = 0; i < 5; ++i) t = conv(t, 0, h, w);
        route = t;
        return conv(t, 0, h, w);
    };
    int inter1, inter2, dummy;
    int net1 = yolo_block(x, inter1);
    conv(net1, 0, h, w, -1, -2);                             // feature_map_1   model.py:42-44
    conv(inter1, 0, h, w, -1, concat1, 0, 1);                // model.py:47-49 conv + upsample + concat
    h = g2h; w = g2w;
    int net2 = yolo_block(concat1, inter2);
    conv(net2, 0, h, w, -1, -3);                             // feature_map_2
    conv(inter2, 0, h, w, -1, concat2, 0, 1);                // model.py:57-59
    h = g3h; w = g3w;
    int net3 = yolo_block(concat2, dummy);
    conv(net3, 0, h, w, -1, -4);                             // feature_map_3
    assign_offsets(P);
    return P;
}

static int get_plan(YoloNet* net, int B, int H, int W, const Plan** out) {
    YOLO_REQUIRE(B > 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, YOLO_ERR_SHAPE,
                 "forward: need B > 0 and H, W positive multiples of 32 (got B=%d, %dx%d)", B, H, W);
    std::lock_guard<std::mutex> g(net->mu);
    auto key = std::make_tuple(B, H, W);
    auto it = net->plans.find(key);
    if (it == net->plans.end()) {
        Plan P = net->kind == 2 ? build_plan_v2(*net, B, H, W) : build_plan(*net, B, H, W);
        if (net->small_plan && net->dtype == YOLO_DTYPE_BF16) {
            P.scratch_off = align_up(P.ws_bytes, 256);
            P.scratch_bytes = kSplitKScratch;
            P.ws_bytes = P.scratch_off + P.scratch_bytes;
        }
        it = net->plans.emplace(key, std::move(P)).first;
    }
    *out = &it->second;
    return YOLO_OK;
}

extern "C" const char* yolo_last_error(void) { return g_err; }
extern "C" int yolo_abi_version(void) { return 1; }

extern "C" int yolo_net_create(int class_num, int dtype, const float* stream, size_t n_floats, YoloNet** out) {
    YOLO_REQUIRE(stream && out, YOLO_ERR_ARG, "net_create: null pointer");
    YOLO_REQUIRE(class_num > 0 && class_num <= 4096, YOLO_ERR_ARG, "net_create: class_num %d", class_num);
    YOLO_REQUIRE(dtype == YOLO_DTYPE_F32 || dtype == YOLO_DTYPE_BF16, YOLO_ERR_ARG, "net_create: dtype %d", dtype);
    std::unique_ptr<YoloNet> net(new YoloNet);
    net->class_num = class_num;
    net->dtype = dtype;
    net->specs = yolov3_specs(class_num);
    size_t need = 0;
    for (const Spec& s : net->specs) need += (size_t)(s.bn ? 4 : 1) * s.cout + (size_t)s.cout * s.cin * s.k * s.k;
    YOLO_REQUIRE(need == n_floats, YOLO_ERR_SHAPE, "net_create: weight stream has %zu floats, architecture needs %zu", n_floats, need);
    const float* ptr = stream;
    for (const Spec& s : net->specs) {
        std::unique_ptr<ConvParams> L(new ConvParams);
        L->cin = s.cin; L->cout = s.cout; L->k = s.k; L->stride = s.stride; L->bn = s.bn; L->dtype = dtype;
        // misc_utils.py:59-95: [beta, gamma, mean, var] or [bias], then kernel [Cout, Cin, kh, kw]
        const float *p0 = ptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
        if (s.bn) { p1 = ptr + s.cout; p2 = ptr + 2 * s.cout; p3 = ptr + 3 * s.cout; ptr += 4 * (size_t)s.cout; }
        else ptr += s.cout;
        const float* wk = ptr;
        const int ci = s.cin, k = s.k;
        int rc = pack_conv(*L, [=](int n, int c, int ky, int kx) { return wk[(((size_t)n * ci + c) * k + ky) * k + kx]; }, p0, p1, p2, p3);
        if (rc != YOLO_OK) return rc;
        ptr += (size_t)s.cout * s.cin * s.k * s.k;
        net->layers.push_back(std::move(L));
    }
    *out = net.release();
    return YOLO_OK;
}

static int forward_lanes(YoloNet* net, const float* images, int B, int H, int W, void* workspace, size_t workspace_bytes,
                         float* fm1, float* fm2, float* fm3, void* stream);

// YOLOv2 weights: the reference keeps them in a TF checkpoint; this entry takes the variables concatenated in
// creation order per conv (yolo_v2.py:75-83): weight [k,k,cin,cout] (HWIO), biases, then for batch-norm layers
// scale, shift, rolling_mean, rolling_variance.  Fold: y = scale*(x-mean)/sqrt(var+1e-5) + shift + biases.
extern "C" int yolo2_net_create(int num_class, int box_per_cell, int dtype, const float* vars, size_t n_floats, YoloNet** out) {
    YOLO_REQUIRE(vars && out, YOLO_ERR_ARG, "yolo2_net_create: null pointer");
    YOLO_REQUIRE(num_class > 0 && num_class <= 4096 && box_per_cell > 0 && box_per_cell <= 16, YOLO_ERR_ARG, "yolo2_net_create: bad sizes");
    YOLO_REQUIRE(dtype == YOLO_DTYPE_F32 || dtype == YOLO_DTYPE_BF16, YOLO_ERR_ARG, "yolo2_net_create: dtype %d", dtype);
    std::unique_ptr<YoloNet> net(new YoloNet);
    net->kind = 2;
    net->class_num = num_class;
    net->box_per_cell = box_per_cell;
    net->dtype = dtype;
    net->specs = yolov2_specs(num_class, box_per_cell);
    size_t need = 0;
    for (const Spec& s : net->specs) need += (size_t)s.k * s.k * s.cin * s.cout + (size_t)(s.bn ? 5 : 1) * s.cout;
    YOLO_REQUIRE(need == n_floats, YOLO_ERR_SHAPE, "yolo2_net_create: %zu floats given, architecture needs %zu", n_floats, need);
    const float* ptr = vars;
    for (const Spec& s : net->specs) {
        std::unique_ptr<ConvParams> L(new ConvParams);
        L->cin = s.cin; L->cout = s.cout; L->k = s.k; L->stride = 1; L->bn = s.bn; L->dtype = dtype; L->act = 2;
        const float* wk = ptr;
        ptr += (size_t)s.k * s.k * s.cin * s.cout;
        const float* biases = ptr;
        ptr += s.cout;
        std::vector<float> scale(s.cout), bias(s.cout);
        if (s.bn) {
            const float *sc = ptr, *sh = ptr + s.cout, *mean = ptr + 2 * (size_t)s.cout, *var = ptr + 3 * (size_t)s.cout;
            ptr += 4 * (size_t)s.cout;
            for (int n = 0; n < s.cout; ++n) {
                const float f = sc[n] / std::sqrt(var[n] + 1e-5f);       // yolo_v2.py:84
                scale[n] = f;
                bias[n] = sh[n] - mean[n] * f + biases[n];               // yolo_v2.py:85
            }
        } else {
            for (int n = 0; n < s.cout; ++n) { scale[n] = 1.0f; bias[n] = biases[n]; }   // yolo_v2.py:88
        }
        const int ci = s.cin, co = s.cout, k = s.k;
        int rc = pack_conv_sb(*L, [=](int n, int c, int ky, int kx) { return wk[(((size_t)ky * k + kx) * ci + c) * co + n]; }, scale, bias);
        if (rc != YOLO_OK) return rc;
        net->layers.push_back(std::move(L));
    }
    *out = net.release();
    return YOLO_OK;
}

// images [B,H,W,3] f32 in [-1,1] -> logits [B,H/32,W/32,box_per_cell*(5+num_class)] f32 (yolo_v2.py:32-70).
extern "C" int yolo2_net_forward(YoloNet* net, const float* images, int B, int H, int W, void* workspace,
                                 size_t workspace_bytes, float* logits, void* stream) {
    YOLO_REQUIRE(net && net->kind == 2, YOLO_ERR_ARG, "yolo2_net_forward: not a YOLOv2 handle");
    return forward_lanes(net, images, B, H, W, workspace, workspace_bytes, logits, nullptr, nullptr, stream);
}

extern "C" int yolo_net_destroy(YoloNet* net) {
    delete net;
    return YOLO_OK;
}

// Small-batch plan (no counterpart in the reference): see include/yolo_hip.h.
extern "C" int yolo_net_set_small_batch_plan(YoloNet* net, int on) {
    YOLO_REQUIRE(net, YOLO_ERR_ARG, "set_small_batch_plan: null handle");
    std::lock_guard<std::mutex> g(net->mu);
    if (net->small_plan != (on != 0)) {
        net->small_plan = on != 0;
        net->plans.clear();                    // workspace sizes change with the scratch area
    }
    return YOLO_OK;
}

extern "C" int yolo_net_num_layers(const YoloNet* net) { return net ? (int)net->layers.size() : 0; }

// ---- lanes ------------------------------------------------------------------------------------------
// Images are independent, so a batch runs as up to MAX_LANES sub-batches, each the whole layer chain on a stream of
// its own (lane 0 on the caller's stream, forked / joined with events).  Kernels of different lanes overlap on the
// GPU: the partly filled last round of workgroups of one launch (e.g. 676 tiles on 512 slots) and the drain / fill
// between dependent launches are covered by the other lane's kernels.  Results do not depend on the split: every
// output element is accumulated in the same order whatever tile or batch it is part of.
// Default, by the work of the call in 416 x 416 images (B * H * W / 416^2): 2 lanes from 26 up, 3 from 56 to 89, 2 again from 90 up
// (batch 72: 5.31 / 5.34 ms for 2 / 3 lanes, 80: 5.96 / 5.89, 88: 6.53 / 6.54, 96: 7.05 / 7.17, 128: 9.44 / 9.56, 160: 11.8 / 12.2; round-3 kernels,
// one MI355X, ms per detect call for 1 / 2 / 3 lanes: batch 16 2.17 / 2.21; 20 2.26 / 2.32 / 2.44; 24 2.40 / 2.45 / 2.59;
// 28 2.98 / 2.75 / 2.74; 40 3.47 / 3.28 / 3.32; 48 3.76 / 3.64 / 3.73; 56 4.73 / 4.29 / 4.20; 608 x 608 batch 8 (work 17) 2.33 / 2.44).
// YOLO_LANES=<n> in the environment (read per call) sets the number of lanes; 1 = one launch per layer for the whole batch.
constexpr int MAX_LANES = 8;
static int lane_count(int B, int H, int W) {
    const char* e = getenv("YOLO_LANES");
    const long long work = (long long)B * H * W / (416 * 416);
    int n = e ? atoi(e) : (work >= 90 ? 2 : work >= 56 ? 3 : work >= 26 ? 2 : 1);
    if (n < 1) n = 1;
    if (n > MAX_LANES) n = MAX_LANES;
    return n < B ? n : B;
}
static int lane_batch(int B, int lanes, int i) { return B / lanes + (i < B % lanes ? 1 : 0); }

extern "C" size_t yolo_net_workspace_bytes(const YoloNet* net, int B, int H, int W) {
    if (!net) return 0;
    const Plan* P = nullptr;
    if (get_plan(const_cast<YoloNet*>(net), B, H, W, &P) != YOLO_OK) return 0;
    size_t need = P->ws_bytes;                      // the unsplit plan always fits (YOLO_LANES=1, forward_timed)
    for (int lanes = 2; lanes <= MAX_LANES && lanes <= B; ++lanes) {
        size_t sum = 0;
        for (int i = 0; i < lanes; ++i) {
            if (get_plan(const_cast<YoloNet*>(net), lane_batch(B, lanes, i), H, W, &P) != YOLO_OK) return 0;
            sum += align_up(P->ws_bytes, 256);
        }
        if (sum > need) need = sum;
    }
    return need;
}

static int forward_impl(YoloNet* net, const float* images, int B, int H, int W, void* workspace, size_t workspace_bytes,
                        float* fm1, float* fm2, float* fm3, void* stream, hipEvent_t* events) {
    YOLO_REQUIRE(net && images && workspace && fm1 && (net->kind == 2 || (fm2 && fm3)), YOLO_ERR_ARG, "forward: null pointer");
    const Plan* P = nullptr;
    int rc = get_plan(net, B, H, W, &P);
    if (rc != YOLO_OK) return rc;
    YOLO_REQUIRE(workspace_bytes >= P->ws_bytes, YOLO_ERR_WORKSPACE, "forward: workspace %zu < %zu bytes", workspace_bytes, P->ws_bytes);
    YOLO_REQUIRE(((uintptr_t)workspace & 255) == 0, YOLO_ERR_ARG, "forward: workspace must be 256-byte aligned");
    const int nout = (net->kind == 2 ? net->box_per_cell : 3) * (5 + net->class_num);
    float* fms[3] = {fm1, fm2, fm3};
    int h_in = H, w_in = W;
    (void)h_in; (void)w_in;
    // bf16 YOLOv3: the stem and the first stride-2 conv run as ONE kernel; the stem output tensor of the plan stays
    // unused (YOLO_NO_FRONT_FUSION=1 in the environment keeps them apart, for A/B measurements)
    const char* nf_env = getenv("YOLO_NO_FRONT_FUSION");   // read per call: the tests toggle it
    const bool no_fusion = nf_env && atoi(nf_env) != 0;
    const bool fuse_front = net->kind == 3 && net->dtype == YOLO_DTYPE_BF16 && !no_fusion && P->steps.size() > 2 &&
                            P->steps[0].layer == 0 && P->steps[1].layer == 1 && P->steps[1].in == P->steps[0].out &&
                            P->steps[1].res < 0;
    // ... and the first residual block (1x1 64->32, 3x3 32->64, + shortcut) as ONE kernel (YOLO_NO_RES_FUSION=1: apart)
    const char* nr_env = getenv("YOLO_NO_RES_FUSION");
    const bool fuse_res = net->kind == 3 && net->dtype == YOLO_DTYPE_BF16 && !(nr_env && atoi(nr_env) != 0) &&
                          P->steps.size() > 4 && P->steps[2].layer == 2 && P->steps[3].layer == 3 &&
                          P->steps[3].in == P->steps[2].out && P->steps[3].res == P->steps[2].in && P->steps[2].res < 0;
    ConvLaunch front_p, res_p;
    memset(&front_p, 0, sizeof(front_p));
    memset(&res_p, 0, sizeof(res_p));
    // bf16 YOLOv2: the stem conv and the max-pool behind it run as ONE kernel (front.hip stem_pool_kernel) when the pool is the stem
    // output's only reader (YOLO_NO_STEM_POOL_FUSION=1 keeps them apart, for A/B measurements and the equality test)
    const char* np_env = getenv("YOLO_NO_STEM_POOL_FUSION");
    const bool fuse_stem_pool = net->dtype == YOLO_DTYPE_BF16 && !(np_env && atoi(np_env) != 0);
    bool skip_pool = false;
    for (const Step& st : P->steps) {
        if (st.layer < 0 && skip_pool) { skip_pool = false; continue; }     // done by the stem launch before it
        if (st.layer < 0) {                       // pooling_layer (yolo_v2.py:91-94)
            const Tens& ti = P->tens[st.in];
            const Tens& to = P->tens[st.out];
            rc = launch_maxpool2x2(net->dtype, (char*)workspace + ti.offset, B, ti.h, ti.w, ti.ld, (char*)workspace + to.offset, (hipStream_t)stream);
            if (rc != YOLO_OK) return rc;
            continue;
        }
        const ConvParams& L = *net->layers[st.layer];
        ConvLaunch p;
        memset(&p, 0, sizeof(p));
        p.B = B;
        if (st.in == -5) { p.in = images; p.H = H; p.W = W; p.in_ld = 3; p.in_coff = 0; }
        else {
            const Tens& t = P->tens[st.in];
            p.in = (char*)workspace + t.offset; p.H = t.h; p.W = t.w; p.in_ld = t.ld; p.in_coff = st.in_coff;
        }
        if (st.out <= -2) { p.out = fms[-2 - st.out]; p.out_ld = nout; p.out_coff = 0; p.out_f32 = 1; }
        else {
            const Tens& t = P->tens[st.out];
            p.out = (char*)workspace + t.offset; p.out_ld = t.ld; p.out_coff = st.out_coff;
        }
        if (st.res >= 0) {
            const Tens& t = P->tens[st.res];
            p.res = (char*)workspace + t.offset; p.res_ld = t.ld; p.res_coff = 0;
        }
        p.upsample = st.upsample;
        p.reorg = st.reorg;
        bool c32_pool = false;
        if (fuse_stem_pool && !L.stem && L.cin == 32 && L.cout == 64 && L.k == 3 && L.stride == 1 && L.bn && st.out >= 0 && st.res < 0 &&
            !st.upsample && !st.reorg && p.H % 8 == 0 && p.W % 16 == 0) {              // YOLOv2's second conv + pool (resblock.hip)
            const size_t si = (size_t)(&st - P->steps.data());
            if (si + 1 < P->steps.size() && P->steps[si + 1].layer < 0 && P->steps[si + 1].in == st.out) {
                int readers = 0;
                for (const Step& o : P->steps) readers += (o.in == st.out) + (o.res == st.out);
                if (readers == 1) {
                    const Tens& to = P->tens[P->steps[si + 1].out];
                    p.out = (char*)workspace + to.offset; p.out_ld = to.ld; p.out_coff = 0;
                    c32_pool = true;
                    skip_pool = true;
                }
            }
        }
        bool stem_pool = false;
        if (fuse_stem_pool && L.stem && st.out >= 0 && st.res < 0 && p.H % 16 == 0 && p.W % 32 == 0) {
            const size_t si = (size_t)(&st - P->steps.data());
            if (si + 1 < P->steps.size() && P->steps[si + 1].layer < 0 && P->steps[si + 1].in == st.out) {
                int readers = 0;
                for (const Step& o : P->steps) readers += (o.in == st.out) + (o.res == st.out);
                if (readers == 1) {
                    const Tens& to = P->tens[P->steps[si + 1].out];
                    p.out = (char*)workspace + to.offset; p.out_ld = to.ld; p.out_coff = 0;
                    stem_pool = true;
                    skip_pool = true;
                }
            }
        }
        if (P->scratch_bytes) {
            p.small_plan = 1;
            p.scratch = (char*)workspace + P->scratch_off;
            p.scratch_bytes = P->scratch_bytes;
        }
        if (events) YOLO_HIP_CHECK(hipEventRecord(events[st.layer], (hipStream_t)stream));
        if (fuse_front && st.layer == 0) {        // held back: launched together with layer 1 (front.hip)
            front_p = p;
            continue;
        }
        if (fuse_res && st.layer == 2) {          // held back: launched together with layer 3 (resblock.hip)
            res_p = p;
            continue;
        }
#ifdef YOLO_EXP
        {
            static const int skipf = [] { const char* e = getenv("YOLO_SKIP_CLASS"); return e ? atoi(e) : 0; }();
            if ((skipf & 1024) && fuse_front && st.layer == 1) continue;      // 1024: front kernel
            if ((skipf & 2048) && fuse_res && st.layer == 3) continue;        // 2048: fused first residual block
            static const int dblf = [] { const char* e = getenv("YOLO_DOUBLE_CLASS"); return e ? atoi(e) : 0; }();
            if ((dblf & 1024) && fuse_front && st.layer == 1) { rc = run_front(*net->layers[0], L, front_p, p, (hipStream_t)stream); if (rc != YOLO_OK) return rc; }
            if ((dblf & 2048) && fuse_res && st.layer == 3) {
                ConvLaunch q1 = res_p, q3 = p;
                fill_conv(*net->layers[2], q1);
                fill_conv(L, q3);
                rc = launch_res_block(q1, q3, (hipStream_t)stream);
                if (rc != YOLO_OK) return rc;
            }
        }
#endif
        if (fuse_res && st.layer == 3) {
            ConvLaunch p1 = res_p, p3 = p;
            fill_conv(*net->layers[2], p1);
            fill_conv(L, p3);
            rc = launch_res_block(p1, p3, (hipStream_t)stream);
            if (rc != YOLO_OK) return rc;
            if (record_kernel_names) {
                net->kernel_names[2] = "(fused into layer 3)";
                net->kernel_names[3] = last_kernel_name();
            }
            continue;
        }
        if (fuse_front && st.layer == 1) {
            rc = run_front(*net->layers[0], L, front_p, p, (hipStream_t)stream);
            if (rc != YOLO_OK) return rc;
            if (record_kernel_names) {
                net->kernel_names[0] = "(fused into layer 1)";     // its event pair brackets nothing
                net->kernel_names[1] = last_kernel_name();
            }
            continue;
        }
#ifdef YOLO_EXP
        {   // experiment builds only: YOLO_SKIP_CLASS=<bit mask> leaves the launches of whole layer classes out (outputs are
            // garbage; the step time without a class = its marginal cost).  1 s2 3x3, 2 1x1 @13, 4 1x1 @26, 8 1x1 @52, 16 1x1 @104,
            // 32 heads + scatters + concat 1x1, 64 3x3 @104, 128 3x3 @13, 256 3x3 @26, 512 3x3 @52
            static const int skip = [] { const char* e = getenv("YOLO_SKIP_CLASS"); return e ? atoi(e) : 0; }();
            int cls = 0;
            const bool special = st.out <= -2 || st.upsample || L.cin == 768 || L.cin == 384;
            if (special) cls = 32;
            else if (L.k == 3 && L.stride == 2) cls = 1;
            else if (L.k == 1) cls = p.W <= 16 ? 2 : p.W <= 32 ? 4 : p.W <= 64 ? 8 : 16;
            else cls = p.W <= 16 ? 128 : p.W <= 32 ? 256 : p.W <= 64 ? 512 : 64;
            if (skip & cls) continue;
            // YOLO_DOUBLE_CLASS=<mask>: launch the class twice (same outputs, valid data): step time minus the base = its cost
            static const int dbl = [] { const char* e = getenv("YOLO_DOUBLE_CLASS"); return e ? atoi(e) : 0; }();
            if (dbl & cls) { rc = run_conv(L, p, (hipStream_t)stream); if (rc != YOLO_OK) return rc; }
        }
#endif
        if (c32_pool) {
            fill_conv(L, p);
            rc = launch_conv3x3_c32_pool(p, (hipStream_t)stream);
        } else if (stem_pool) {
            fill_conv(L, p);
            rc = launch_stem_pool(p, (hipStream_t)stream);
        } else {
            rc = run_conv(L, p, (hipStream_t)stream);
        }
        if (rc != YOLO_OK) return rc;
        if (record_kernel_names) net->kernel_names[st.layer] = last_kernel_name();
    }
    if (events) YOLO_HIP_CHECK(hipEventRecord(events[P->steps.size()], (hipStream_t)stream));
    return YOLO_OK;
}

// Runs lane_fn(lane, first image, images, stream) for every lane: lane 0 on the caller's stream, the others on the
// handle's side streams, forked from / joined into the caller's stream with events.
template <class F>
static int run_lanes(YoloNet* net, int lanes, int B, hipStream_t main_s, F&& lane_fn) {
    if (lanes == 1) return lane_fn(0, 0, B, main_s);
    {
        std::lock_guard<std::mutex> g(net->mu);
        if (!net->fork_ev) YOLO_HIP_CHECK(hipEventCreateWithFlags(&net->fork_ev, hipEventDisableTiming));
        while ((int)net->side.size() < lanes - 1) {
            hipStream_t s;
            hipEvent_t e;
            YOLO_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            net->side.push_back(s);
            YOLO_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            net->side_done.push_back(e);
        }
    }
    YOLO_HIP_CHECK(hipEventRecord(net->fork_ev, main_s));
    int rc = YOLO_OK, b0 = 0, forked = 0;
    for (int i = 0; i < lanes && rc == YOLO_OK; ++i) {
        const int Bi = lane_batch(B, lanes, i);
        const hipStream_t s = i == 0 ? main_s : net->side[i - 1];
        if (i > 0) {
            if (hipStreamWaitEvent(s, net->fork_ev, 0) != hipSuccess) { set_error("forward: fork failed"); rc = YOLO_ERR_HIP; break; }
            forked = i;
        }
        rc = lane_fn(i, b0, Bi, s);
        b0 += Bi;
    }
    // join every lane that was forked, also after an error, so that the caller's stream orders after all launched work
    for (int i = 1; i <= forked; ++i) {
        if (hipEventRecord(net->side_done[i - 1], net->side[i - 1]) != hipSuccess ||
            hipStreamWaitEvent(main_s, net->side_done[i - 1], 0) != hipSuccess) {
            if (rc == YOLO_OK) { set_error("forward: join failed"); rc = YOLO_ERR_HIP; }
        }
    }
    return rc;
}

static int forward_lanes(YoloNet* net, const float* images, int B, int H, int W, void* workspace, size_t workspace_bytes,
                         float* fm1, float* fm2, float* fm3, void* stream) {
    YOLO_REQUIRE(net && B > 0, YOLO_ERR_ARG, "forward: bad handle or batch");
    const int lanes = lane_count(B, H, W);
    if (lanes == 1) return forward_impl(net, images, B, H, W, workspace, workspace_bytes, fm1, fm2, fm3, stream, nullptr);
    YOLO_REQUIRE(images && workspace && fm1 && (net->kind == 2 || (fm2 && fm3)), YOLO_ERR_ARG, "forward: null pointer");
    YOLO_REQUIRE(((uintptr_t)workspace & 255) == 0, YOLO_ERR_ARG, "forward: workspace must be 256-byte aligned");
    YOLO_REQUIRE(H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, YOLO_ERR_SHAPE, "forward: H, W must be multiples of 32 (got %dx%d)", H, W);
    size_t ws_off[MAX_LANES + 1] = {0};
    for (int i = 0; i < lanes; ++i) {
        const Plan* P = nullptr;
        int rc = get_plan(net, lane_batch(B, lanes, i), H, W, &P);
        if (rc != YOLO_OK) return rc;
        ws_off[i + 1] = ws_off[i] + align_up(P->ws_bytes, 256);
    }
    YOLO_REQUIRE(workspace_bytes >= ws_off[lanes], YOLO_ERR_WORKSPACE, "forward: workspace %zu < %zu bytes", workspace_bytes, ws_off[lanes]);
    const int nout = (net->kind == 2 ? net->box_per_cell : 3) * (5 + net->class_num);
    const size_t px1 = (size_t)(H / 32) * (W / 32) * nout;
    return run_lanes(net, lanes, B, (hipStream_t)stream, [&](int i, int b0, int Bi, hipStream_t s) {
        return forward_impl(net, images + (size_t)b0 * H * W * 3, Bi, H, W, (char*)workspace + ws_off[i], ws_off[i + 1] - ws_off[i],
                            fm1 + (size_t)b0 * px1, fm2 ? fm2 + (size_t)b0 * px1 * 4 : nullptr, fm3 ? fm3 + (size_t)b0 * px1 * 16 : nullptr,
                            s, nullptr);
    });
}

// ---- the whole chain in one call: forward -> predict -> conf*prob -> per-class NMS (test_single_image.py:31-40) ----------
// Per lane: [forward workspace | fm1 | fm2 | fm3 | yolo_detect_post workspace]; the post-processing of a lane runs on the
// lane's stream right behind its last conv, so it overlaps the other lane's convs.
struct DetectLayout { size_t fwd, fm[3], post, total; };
static int detect_layout(YoloNet* net, int Bi, int H, int W, int max_boxes, DetectLayout* d) {
    const Plan* P = nullptr;
    int rc = get_plan(net, Bi, H, W, &P);
    if (rc != YOLO_OK) return rc;
    const size_t px1 = (size_t)(H / 32) * (W / 32) * 3 * (5 + net->class_num) * sizeof(float);
    const size_t post = yolo_detect_post_workspace_bytes(Bi, H, W, net->class_num, max_boxes);
    YOLO_REQUIRE(post > 0, YOLO_ERR_ARG, "detect: bad sizes B=%d H=%d W=%d max_boxes=%d", Bi, H, W, max_boxes);
    size_t off = 0;
    d->fwd = off; off = align_up(off + P->ws_bytes, 256);
    d->fm[0] = off; off = align_up(off + (size_t)Bi * px1, 256);
    d->fm[1] = off; off = align_up(off + (size_t)Bi * px1 * 4, 256);
    d->fm[2] = off; off = align_up(off + (size_t)Bi * px1 * 16, 256);
    d->post = off; off = align_up(off + post, 256);
    d->total = off;
    return YOLO_OK;
}

extern "C" size_t yolo_net_detect_workspace_bytes(const YoloNet* net, int B, int H, int W, int max_boxes) {
    if (!net || net->kind != 3 || B <= 0 || max_boxes <= 0) return 0;
    size_t need = 0;
    for (int lanes = 1; lanes <= MAX_LANES && lanes <= B; ++lanes) {
        size_t sum = 0;
        for (int i = 0; i < lanes; ++i) {
            DetectLayout d;
            if (detect_layout(const_cast<YoloNet*>(net), lane_batch(B, lanes, i), H, W, max_boxes, &d) != YOLO_OK) return 0;
            sum += d.total;
        }
        if (sum > need) need = sum;
    }
    return need;
}

extern "C" int yolo_net_detect(YoloNet* net, const float* images, int B, int H, int W, const float* anchors, int max_boxes,
                               float score_thresh, float iou_thresh, float* out_boxes, float* out_scores, int32_t* out_labels,
                               int32_t* out_index, int32_t* out_counts, void* workspace, size_t workspace_bytes, void* stream) {
    YOLO_REQUIRE(net && net->kind == 3, YOLO_ERR_ARG, "yolo_net_detect: not a YOLOv3 handle");
    YOLO_REQUIRE(images && anchors && out_boxes && out_scores && out_labels && out_counts && workspace, YOLO_ERR_ARG,
                 "yolo_net_detect: null pointer");
    YOLO_REQUIRE(B > 0 && max_boxes > 0, YOLO_ERR_ARG, "yolo_net_detect: non-positive size");
    YOLO_REQUIRE(H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, YOLO_ERR_SHAPE, "yolo_net_detect: H, W must be multiples of 32 (got %dx%d)", H, W);
    YOLO_REQUIRE(((uintptr_t)workspace & 255) == 0, YOLO_ERR_ARG, "yolo_net_detect: workspace must be 256-byte aligned");
    const int lanes = lane_count(B, H, W);
    DetectLayout lay[MAX_LANES];
    size_t base[MAX_LANES + 1] = {0};
    for (int i = 0; i < lanes; ++i) {
        int rc = detect_layout(net, lane_batch(B, lanes, i), H, W, max_boxes, &lay[i]);
        if (rc != YOLO_OK) return rc;
        base[i + 1] = base[i] + lay[i].total;
    }
    YOLO_REQUIRE(workspace_bytes >= base[lanes], YOLO_ERR_WORKSPACE, "yolo_net_detect: workspace %zu < %zu bytes", workspace_bytes, base[lanes]);
    const size_t cap = (size_t)net->class_num * max_boxes;
    return run_lanes(net, lanes, B, (hipStream_t)stream, [&](int i, int b0, int Bi, hipStream_t s) {
        char* w = (char*)workspace + base[i];
        const DetectLayout& d = lay[i];
        float* f1 = (float*)(w + d.fm[0]);
        float* f2 = (float*)(w + d.fm[1]);
        float* f3 = (float*)(w + d.fm[2]);
        int rc = forward_impl(net, images + (size_t)b0 * H * W * 3, Bi, H, W, w + d.fwd, d.fm[0] - d.fwd, f1, f2, f3, s, nullptr);
        if (rc != YOLO_OK) return rc;
        return yolo_detect_post(f1, f2, f3, Bi, H, W, net->class_num, anchors, max_boxes, score_thresh, iou_thresh,
                                out_boxes + (size_t)b0 * cap * 4, out_scores + (size_t)b0 * cap, out_labels + (size_t)b0 * cap,
                                out_index ? out_index + (size_t)b0 * cap : nullptr, out_counts + b0, w + d.post, d.total - d.post, s);
    });
}

extern "C" int yolo_net_forward(YoloNet* net, const float* images, int B, int H, int W, void* workspace, size_t workspace_bytes,
                                float* fm1, float* fm2, float* fm3, void* stream) {
    YOLO_REQUIRE(net && net->kind == 3, YOLO_ERR_ARG, "yolo_net_forward: not a YOLOv3 handle");
    return forward_lanes(net, images, B, H, W, workspace, workspace_bytes, fm1, fm2, fm3, stream);
}

// Measurement aid: one forward with a hipEvent before every layer's launch (on `stream`), then a stream
// synchronisation; layer_ms[i] = device time between the launch of layer i and of layer i+1.
extern "C" int yolo_net_forward_timed(YoloNet* net, const float* images, int B, int H, int W, void* workspace,
                                      size_t workspace_bytes, float* fm1, float* fm2, float* fm3, void* stream,
                                      float* layer_ms, int n_layers) {
    YOLO_REQUIRE(net && layer_ms && net->kind == 3, YOLO_ERR_ARG, "forward_timed: needs a YOLOv3 handle and layer_ms");
    const int nl = (int)net->layers.size();
    YOLO_REQUIRE(n_layers == nl, YOLO_ERR_ARG, "forward_timed: layer_ms must hold %d floats", nl);
    std::vector<hipEvent_t> ev(nl + 1);
    for (auto& e : ev) YOLO_HIP_CHECK(hipEventCreate(&e));
    net->kernel_names.assign(nl, std::string());
    record_kernel_names = true;
    int rc = forward_impl(net, images, B, H, W, workspace, workspace_bytes, fm1, fm2, fm3, stream, ev.data());
    record_kernel_names = false;
    if (rc == YOLO_OK) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) { set_error("forward_timed: sync failed: %s", hipGetErrorString(e)); rc = YOLO_ERR_HIP; }
    }
    for (int i = 0; i < nl && rc == YOLO_OK; ++i)
        if (hipEventElapsedTime(&layer_ms[i], ev[i], ev[i + 1]) != hipSuccess) { set_error("forward_timed: elapsed"); rc = YOLO_ERR_HIP; }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

// Static description of layer i (tests, benchmarks): fills cin, cout, ksize, stride.
extern "C" int yolo_net_layer_info(const YoloNet* net, int layer, int* cin, int* cout, int* ksize, int* stride) {
    YOLO_REQUIRE(net && layer >= 0 && layer < (int)net->specs.size() && cin && cout && ksize && stride, YOLO_ERR_ARG, "layer_info: bad argument");
    const Spec& s = net->specs[layer];
    *cin = s.cin; *cout = s.cout; *ksize = s.k; *stride = s.stride;
    return YOLO_OK;
}

// Name of the kernel instantiation the last yolo_net_forward_timed launched for `layer` (as rocprofv3 prints
// it, without return type and argument list); "" before the first timed forward.
extern "C" const char* yolo_net_layer_kernel(const YoloNet* net, int layer) {
    if (!net || layer < 0 || layer >= (int)net->kernel_names.size()) return "";
    return net->kernel_names[layer].c_str();
}

// ---- single layer ---------------------------------------------------------------------------------
extern "C" int yolo_conv_create(int dtype, int cin, int cout, int ksize, int stride, int bn, const float* w_hwio,
                                const float* p0, const float* p1, const float* p2, const float* p3, YoloConv** out) {
    YOLO_REQUIRE(w_hwio && p0 && out, YOLO_ERR_ARG, "conv_create: null pointer");
    YOLO_REQUIRE(!bn || (p1 && p2 && p3), YOLO_ERR_ARG, "conv_create: bn needs beta, gamma, mean, var");
    YOLO_REQUIRE(dtype == YOLO_DTYPE_F32 || dtype == YOLO_DTYPE_BF16, YOLO_ERR_ARG, "conv_create: dtype %d", dtype);
    YOLO_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && cin > 0 && cout > 0, YOLO_ERR_UNSUPPORTED,
                 "conv_create: ksize %d stride %d", ksize, stride);
    const int sz = dtype == YOLO_DTYPE_BF16 ? 2 : 4;
    if (cin == 3)
        YOLO_REQUIRE(ksize == 3 && stride == 1 && cout == 32, YOLO_ERR_UNSUPPORTED, "conv_create: Cin=3 only as the 3x3 s1 3->32 stem");
    else
        YOLO_REQUIRE((cin * sz) % 64 == 0 && (128 % (cin * sz) == 0 || (cin * sz) % 128 == 0), YOLO_ERR_SHAPE,
                     "conv_create: Cin=%d unsupported (need Cin*sizeof a divisor or multiple of 128 bytes)", cin);
    std::unique_ptr<YoloConv> c(new YoloConv);
    ConvParams& L = c->p;
    L.cin = cin; L.cout = cout; L.k = ksize; L.stride = stride; L.bn = bn; L.dtype = dtype;
    int rc = pack_conv(L, [=](int n, int ci, int ky, int kx) { return w_hwio[(((size_t)ky * ksize + kx) * cin + ci) * cout + n]; },
                       p0, p1, p2, p3);
    if (rc != YOLO_OK) return rc;
    *out = c.release();
    return YOLO_OK;
}

// conv_layer of YOLOv2 (reference yolo_v2/yolo/yolo_v2.py:73-89) as a single layer: SAME conv (no bias) ->
// batch_normalization(mean, variance, shift, scale, 1e-5) -> + biases -> max(0.1x, x); batch_norm == 0: conv + biases.
extern "C" int yolo2_conv_create(int dtype, int cin, int cout, int ksize, int batch_norm, const float* weight_hwio,
                                 const float* biases, const float* scale, const float* shift, const float* mean,
                                 const float* variance, YoloConv** out) {
    YOLO_REQUIRE(weight_hwio && biases && out, YOLO_ERR_ARG, "conv2_create: null pointer");
    YOLO_REQUIRE(!batch_norm || (scale && shift && mean && variance), YOLO_ERR_ARG, "conv2_create: batch_norm needs scale, shift, mean, variance");
    YOLO_REQUIRE(dtype == YOLO_DTYPE_F32 || dtype == YOLO_DTYPE_BF16, YOLO_ERR_ARG, "conv2_create: dtype %d", dtype);
    YOLO_REQUIRE((ksize == 1 || ksize == 3) && cin > 0 && cout > 0, YOLO_ERR_UNSUPPORTED, "conv2_create: ksize %d", ksize);
    const int sz = dtype == YOLO_DTYPE_BF16 ? 2 : 4;
    if (cin == 3)
        YOLO_REQUIRE(ksize == 3 && cout == 32, YOLO_ERR_UNSUPPORTED, "conv2_create: Cin=3 only as the 3x3 3->32 stem");
    else
        YOLO_REQUIRE((cin * sz) % 64 == 0 && (128 % (cin * sz) == 0 || (cin * sz) % 128 == 0), YOLO_ERR_SHAPE,
                     "conv2_create: Cin=%d unsupported (need Cin*sizeof a divisor or multiple of 128 bytes)", cin);
    std::unique_ptr<YoloConv> c(new YoloConv);
    ConvParams& L = c->p;
    L.cin = cin; L.cout = cout; L.k = ksize; L.stride = 1; L.bn = batch_norm; L.dtype = dtype; L.act = 2;
    std::vector<float> sc(cout), bi(cout);
    for (int n = 0; n < cout; ++n) {
        if (batch_norm) {
            const float f = scale[n] / std::sqrt(variance[n] + 1e-5f);   // yolo_v2.py:84
            sc[n] = f;
            bi[n] = shift[n] - mean[n] * f + biases[n];                  // yolo_v2.py:85
        } else {
            sc[n] = 1.0f;
            bi[n] = biases[n];                                           // yolo_v2.py:88
        }
    }
    int rc = pack_conv_sb(L, [=](int n, int ci, int ky, int kx) { return weight_hwio[(((size_t)ky * ksize + kx) * cin + ci) * cout + n]; }, sc, bi);
    if (rc != YOLO_OK) return rc;
    *out = c.release();
    return YOLO_OK;
}

extern "C" int yolo_conv_destroy(YoloConv* c) {
    delete c;
    return YOLO_OK;
}

static void conv_ws_layout(const ConvParams& L, int B, int H, int W, size_t* in_b, size_t* out_b, size_t* res_b) {
    const int pad = (L.k - 1) / 2;
    const int ho = (H + 2 * pad - L.k) / L.stride + 1, wo = (W + 2 * pad - L.k) / L.stride + 1;
    const bool bf = L.dtype == YOLO_DTYPE_BF16;
    const int cpad = (int)align_up((size_t)L.cout, 8);
    *in_b = (bf && !L.stem) ? align_up((size_t)B * H * W * L.cin * 2, 256) : 0;
    *out_b = (bf || cpad != L.cout) ? align_up((size_t)B * ho * wo * cpad * (bf ? 2 : 4), 256) : 0;
    *res_b = bf ? align_up((size_t)B * ho * wo * L.cout * 2, 256) : 0;
}

extern "C" size_t yolo_conv_workspace_bytes(const YoloConv* c, int B, int H, int W) {
    if (!c || B <= 0 || H <= 0 || W <= 0) return 0;
    size_t a, b, r;
    conv_ws_layout(c->p, B, H, W, &a, &b, &r);
    return a + b + r + 256;
}

extern "C" int yolo_conv_forward(YoloConv* c, const float* in, int B, int H, int W, const float* residual, float* out,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
    YOLO_REQUIRE(c && in && out, YOLO_ERR_ARG, "conv_forward: null pointer");
    YOLO_REQUIRE(B > 0 && H > 0 && W > 0, YOLO_ERR_ARG, "conv_forward: non-positive shape");
    const ConvParams& L = c->p;
    if (L.stride == 2) YOLO_REQUIRE(H % 2 == 0 && W % 2 == 0, YOLO_ERR_SHAPE, "conv_forward: stride-2 conv needs even H, W");
    hipStream_t stream = (hipStream_t)stream_;
    size_t in_b, out_b, res_b;
    conv_ws_layout(L, B, H, W, &in_b, &out_b, &res_b);
    YOLO_REQUIRE(in_b + out_b + res_b == 0 || (workspace && workspace_bytes >= in_b + out_b + res_b), YOLO_ERR_WORKSPACE,
                 "conv_forward: workspace %zu < %zu bytes", workspace_bytes, in_b + out_b + res_b);
    YOLO_REQUIRE(((uintptr_t)workspace & 255) == 0, YOLO_ERR_ARG, "conv_forward: workspace must be 256-byte aligned");
    const bool bf = L.dtype == YOLO_DTYPE_BF16;
    const int pad = (L.k - 1) / 2;
    const int ho = (H + 2 * pad - L.k) / L.stride + 1, wo = (W + 2 * pad - L.k) / L.stride + 1;
    char* ws = (char*)workspace;
    ConvLaunch p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.in_ld = L.cin;
    int rc;
    if (in_b) {
        rc = launch_cast_f32_to_bf16(in, ws, (size_t)B * H * W * L.cin, stream);
        if (rc != YOLO_OK) return rc;
        p.in = ws;
    } else p.in = in;
    const bool head = (L.cout % 8) != 0;          // detection conv: direct f32 scalar stores
    if (head) { p.out = out; p.out_ld = L.cout; p.out_f32 = 1; }
    else if (bf) { p.out = ws + in_b; p.out_ld = L.cout; }
    else { p.out = out; p.out_ld = L.cout; }
    if (residual) {
        YOLO_REQUIRE(!head, YOLO_ERR_UNSUPPORTED, "conv_forward: residual on a detection conv");
        if (bf) {
            rc = launch_cast_f32_to_bf16(residual, ws + in_b + out_b, (size_t)B * ho * wo * L.cout, stream);
            if (rc != YOLO_OK) return rc;
            p.res = ws + in_b + out_b;
        } else p.res = residual;
        p.res_ld = L.cout;
    }
    rc = run_conv(L, p, stream);
    if (rc != YOLO_OK) return rc;
    if (bf && !head) return launch_cast_bf16_to_f32(ws + in_b, out, (size_t)B * ho * wo * L.cout, stream);
    return YOLO_OK;
}

// Measurement aid (no counterpart in the reference): run one conv layer `iters` times on device
// buffers of the layer's own dtype (allocated and freed here, random contents) and report the average
// kernel time from HIP events.  Used by scripts/bench_layer.py to tune tile shapes per layer.
extern "C" int yolo_conv_benchmark(YoloConv* c, int B, int H, int W, int with_residual, int iters, float* avg_ms) {
    YOLO_REQUIRE(c && avg_ms && B > 0 && H > 0 && W > 0 && iters > 0, YOLO_ERR_ARG, "conv_benchmark: bad argument");
    const ConvParams& L = c->p;
    const int sz = L.dtype == YOLO_DTYPE_BF16 ? 2 : 4;
    const int pad = (L.k - 1) / 2;
    const int ho = (H + 2 * pad - L.k) / L.stride + 1, wo = (W + 2 * pad - L.k) / L.stride + 1;
    const size_t in_bytes = align_up((size_t)B * H * W * L.cin * (L.stem ? 4 : sz), 256);
    const size_t out_bytes = align_up((size_t)B * ho * wo * align_up((size_t)L.cout, 8) * 4, 256);
    // A ring of input / output / residual sets larger than L2 + MALL (YOLO_BENCH_RING_MB, default 768): launch i works on set
    // i % R, so that - as in the network, where the input was just written by another kernel - nothing a launch reads is left
    // in the caches by the launch before it.  (With one set the 1x1 layers of the small maps ran out of L2 and looked up to
    // 20 % better than they are in the network.)
    static const size_t ring_mb = [] { const char* e = getenv("YOLO_BENCH_RING_MB"); return e ? (size_t)atoi(e) : (size_t)768; }();
    const size_t set_bytes = in_bytes + 2 * out_bytes;
    int R = (int)((ring_mb << 20) / set_bytes) + 1;
    if (R > 64) R = 64;
    char* base = nullptr;
    YOLO_HIP_CHECK(hipMalloc((void**)&base, set_bytes * R));
    {   // random finite contents (bf16 pattern 0x3cxx..0x3fxx / small f32)
        std::vector<uint16_t> h(in_bytes / 2);
        uint32_t s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u + ((s >> 16) & 0x3ffu) + ((s >> 3) & 0x8000u)); }
        int rc0 = YOLO_OK;
        for (int r = 0; r < R && rc0 == YOLO_OK; ++r)
            if (hipMemcpy(base + r * set_bytes, h.data(), in_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemset(base + r * set_bytes + in_bytes, 0, 2 * out_bytes) != hipSuccess) rc0 = YOLO_ERR_HIP;
        if (rc0 != YOLO_OK) { (void)hipFree(base); set_error("conv_benchmark: fill failed"); return rc0; }
    }
    ConvLaunch p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.in_ld = L.cin; p.out_ld = L.cout;
    if (L.cout % 8) p.out_f32 = 1;
    auto set = [&](int i) {
        char* b = base + (size_t)(i % R) * set_bytes;
        p.in = b; p.out = b + in_bytes;
        if (with_residual && !p.out_f32) { p.res = b + in_bytes + out_bytes; p.res_ld = L.cout; }
    };
    int rc = YOLO_OK;
    // YOLO_BENCH_STREAMS=<n> (n > 1): launches go round-robin over n streams (independent buffer sets), host-timed between two
    // device synchronisations - the rate of the kernel when other launches fill its partly filled rounds and its ramps
    static const int nstreams = [] { const char* e = getenv("YOLO_BENCH_STREAMS"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n > 8 ? 8 : n; }();
    if (nstreams > 1 && R >= nstreams) {
        hipStream_t st[8];
        for (int i = 0; i < nstreams; ++i) YOLO_HIP_CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        for (int i = 0; i < 2 * nstreams && rc == YOLO_OK; ++i) { set(i); rc = run_conv(L, p, st[i % nstreams]); }
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters && rc == YOLO_OK; ++i) { set(i); rc = run_conv(L, p, st[i % nstreams]); }
        if (hipDeviceSynchronize() != hipSuccess) { set_error("conv_benchmark: sync failed"); rc = YOLO_ERR_HIP; }
        *avg_ms = (float)(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters);
        for (int i = 0; i < nstreams; ++i) (void)hipStreamDestroy(st[i]);
        (void)hipFree(base);
        return rc;
    }
    hipEvent_t e0, e1;
    YOLO_HIP_CHECK(hipEventCreate(&e0));
    YOLO_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3 && rc == YOLO_OK; ++i) { set(i); rc = run_conv(L, p, nullptr); }
    if (rc == YOLO_OK) {
        (void)hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters && rc == YOLO_OK; ++i) { set(i + 3); rc = run_conv(L, p, nullptr); }
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess) { set_error("conv_benchmark: sync failed"); rc = YOLO_ERR_HIP; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_ms = ms / iters;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(base);
    return rc;
}
