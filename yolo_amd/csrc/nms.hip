// Per-class NMS for gfx950.  Replaces gpu_nms (reference yolo_v3/utils/nms_utils.py:7-37: score mask +
// tf.boolean_mask + tf.image.non_max_suppression + tf.gather per class) and, in YOLO_NMS_PY mode,
// cpu_nms / py_nms (nms_utils.py:40-90).  Integer/compare work plus IEEE single ops only: results are
// bit-identical to the CPU oracle (every float op is a single correctly rounded +,-,*,/; this TU is
// built with -ffp-contract=off and uses the _rn intrinsics).
//
// Three kernels per call:
//  1. nms_compact : streams scores [B,n,C] once (HBM-bound), tile -> LDS, one wave per class column,
//                   ballot + prefix compaction, ONE atomic per (tile, class) -> candidate keys
//                   key = (~sortable(score) << 32) | index   (TF mode; PY mode stores ~index so that
//                   equal scores come out in descending index order like argsort()[::-1]).
//  2. nms_select  : one workgroup per (image, class).
//        k <= 2048 : keys sorted in LDS (bitonic), then wave 0 walks the sorted list 64 candidates
//                    at a time: test against the selected boxes (LDS), build the 64x64 IoU bitmask
//                    of the chunk (one 64-bit word per lane), run the LDS-resident suppression scan,
//                    append the survivors; stops at max_boxes.
//        k  > 2048 : (eval-mode stress, score_thresh 0.001) the same walk over successive key ranges of at most
//                    2048 candidates, best first, cut out by a radix histogram of the keys; stops as soon as
//                    max_boxes boxes are selected (normally inside the first range).
//  3. nms_gather  : per image, prefix over class counts -> class-ordered dense output rows.
#include <mutex>
#include "decode_common.h"

namespace yolo {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned long long u64;

constexpr int kSortCap = 2048;   // candidates per (image, class) handled by the LDS sort path
constexpr int kSelCap = 1024;    // max_boxes limit (selected boxes are kept in LDS)
constexpr int kTileJ = 128;      // boxes per compaction tile
constexpr int kTileC = 96;       // classes per compaction tile (LDS: 128 x 97 floats)
constexpr u64 kDead = ~0ull;

// ---- exact-arithmetic overlap tests ------------------------------------------------------------
// TF 1.x non_max_suppression_op.cc, IOUGreaterThanThreshold (boxes are read as pairs (0,2), (1,3)).
__device__ __forceinline__ bool tf_suppresses(const f32x4 a, const f32x4 b, float thr) {
    const float a_lo0 = fminf(a[0], a[2]), a_hi0 = fmaxf(a[0], a[2]);
    const float a_lo1 = fminf(a[1], a[3]), a_hi1 = fmaxf(a[1], a[3]);
    const float b_lo0 = fminf(b[0], b[2]), b_hi0 = fmaxf(b[0], b[2]);
    const float b_lo1 = fminf(b[1], b[3]), b_hi1 = fmaxf(b[1], b[3]);
    const float area_a = __fmul_rn(__fsub_rn(a_hi0, a_lo0), __fsub_rn(a_hi1, a_lo1));
    const float area_b = __fmul_rn(__fsub_rn(b_hi0, b_lo0), __fsub_rn(b_hi1, b_lo1));
    if (area_a <= 0.0f || area_b <= 0.0f) return false;
    const float d0 = fmaxf(__fsub_rn(fminf(a_hi0, b_hi0), fmaxf(a_lo0, b_lo0)), 0.0f);
    const float d1 = fmaxf(__fsub_rn(fminf(a_hi1, b_hi1), fmaxf(a_lo1, b_lo1)), 0.0f);
    const float inter = __fmul_rn(d0, d1);
    const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
    return iou > thr;
}

// nms_utils.py:48,54-62 (areas without min/max, +1 on w/h, keep iff ovr <= thr).
__device__ __forceinline__ bool py_drops(const f32x4 s, const f32x4 b, float thr) {
    const float area_s = __fmul_rn(__fsub_rn(s[2], s[0]), __fsub_rn(s[3], s[1]));
    const float area_b = __fmul_rn(__fsub_rn(b[2], b[0]), __fsub_rn(b[3], b[1]));
    const float xx1 = fmaxf(s[0], b[0]), yy1 = fmaxf(s[1], b[1]);
    const float xx2 = fminf(s[2], b[2]), yy2 = fminf(s[3], b[3]);
    const float w = fmaxf(0.0f, __fadd_rn(__fsub_rn(xx2, xx1), 1.0f));
    const float h = fmaxf(0.0f, __fadd_rn(__fsub_rn(yy2, yy1), 1.0f));
    const float inter = __fmul_rn(w, h);
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_s, area_b), inter));
    return !(ovr <= thr);
}

template <int MODE>
__device__ __forceinline__ bool suppresses(const f32x4 sel, const f32x4 b, float thr) {
    return MODE == YOLO_NMS_TF ? tf_suppresses(sel, b, thr) : py_drops(sel, b, thr);
}

template <int MODE>
__device__ __forceinline__ u64 make_key(float score, unsigned idx) {
    const unsigned bits = __float_as_uint(score + 0.0f);                 // -0 -> +0
    const unsigned u = bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // monotone in the float order
    return ((u64)(~u) << 32) | (MODE == YOLO_NMS_TF ? idx : ~idx);
}
template <int MODE>
__device__ __forceinline__ unsigned key_index(u64 key) {
    const unsigned lo = (unsigned)key;
    return MODE == YOLO_NMS_TF ? lo : ~lo;
}

// score stored in a candidate key (inverse of make_key's monotone map)
__device__ __forceinline__ float key_score(u64 key) {
    const unsigned u = ~(unsigned)(key >> 32);
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

// ---- 1. compaction -----------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void nms_compact_kernel(const float* __restrict__ scores, int n, int C, float thr,
                                                          int* __restrict__ cand_count, u64* __restrict__ cand_keys) {
    __shared__ float tile[kTileJ * (kTileC + 1)];
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * kTileJ;
    const int nj = min(kTileJ, n - j0);
    const int c0 = blockIdx.z * kTileC;
    const int cw = min(kTileC, C - c0);
    constexpr int ldt = kTileC + 1;              // odd stride: column reads are bank-conflict free
    const float* src = scores + ((long long)b * n + j0) * C + c0;
    for (int e = threadIdx.x; e < nj * cw; e += 256) {
        const int j = e / cw;
        const int c = e - j * cw;
        tile[j * ldt + c] = src[(long long)j * C + c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int cl = wave; cl < cw; cl += 4) {
        const int c = c0 + cl;
        int* cnt = cand_count + (long long)b * C + c;
        u64* keys = cand_keys + ((long long)b * C + c) * n;
#pragma unroll
        for (int h = 0; h < kTileJ / 64; ++h) {
            const int j = h * 64 + lane;
            const float s = j < nj ? tile[j * ldt + cl] : 0.f;
            const bool pass = j < nj && s >= thr;                        // nms_utils.py:22 / :74
            const u64 m = __ballot(pass);
            if (m == 0) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(cnt, __popcll(m));
            base = __shfl(base, 0);
            if (pass) keys[base + __popcll(m & ((1ull << lane) - 1))] = make_key<MODE>(s, (unsigned)(j0 + j));
        }
    }
}

// ---- 1b. decode + score + compaction in one pass (yolo_detect_post) ----------------------------------------
// The chain predict -> pred_confs * pred_probs -> score mask of gpu_nms (reference yolo_v3/model.py:110-153,
// yolo_v3/test_single_image.py:38, yolo_v3/utils/nms_utils.py:22) without the [B,n,C] probs / scores tensors ever
// reaching HBM: a tile of 128 boxes x (5+C) logits is read once (contiguous), sigmoid-ed into LDS, the four box
// attributes are decoded and stored, and every class column is compacted into candidate keys exactly as
// nms_compact_kernel does.  The score of a candidate is sigmoid_fast(t4) * sigmoid_fast(t5+c), the expression
// decode_kernel evaluates, so keys (and therefore the NMS result) are bit-identical to the unfused path.
template <int MODE>
__global__ __launch_bounds__(256) void decode_compact_kernel(DecodeK k, float thr, int* __restrict__ cand_count,
                                                             u64* __restrict__ cand_keys) {
    extern __shared__ float dtile[];            // [kTileJ][ldt]
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * kTileJ;
    const int nj = min(kTileJ, k.n - j0);
    const int A = k.A, C = k.C;
    const int ldt = A | 1;                      // odd stride: column reads are bank-conflict free
    const float inv_a = 1.0f / (float)A;
    for (int e = threadIdx.x; e < nj * A; e += 256) {
        int j = (int)((float)e * inv_a);
        int attr = e - j * A;
        if (attr < 0) { attr += A; --j; }
        if (attr >= A) { attr -= A; ++j; }
        const int jg = j0 + j;
        const int s = (jg >= k.off[1]) + (jg >= k.off[2]);
        const int ns = k.off[s + 1] - k.off[s];
        const float v = k.fm[s][((long long)b * ns + (jg - k.off[s])) * A + attr];
        dtile[j * ldt + attr] = attr < 4 ? v : sigmoid_fast(v);            // model.py:134-135
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nj * 4; e += 256) {                      // boxes (model.py:76-108,147-152)
        const int j = e >> 2, attr = e & 3;
        const int jg = j0 + j;
        const int s = (jg >= k.off[1]) + (jg >= k.off[2]);
        k.boxes[((long long)b * k.n + jg) * 4 + attr] = decode_box_attr(k, s, jg - k.off[s], dtile + j * ldt, attr);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C; c += 4) {
        int* cnt = cand_count + (long long)b * C + c;
        u64* keys = cand_keys + ((long long)b * C + c) * k.n;
#pragma unroll
        for (int h = 0; h < kTileJ / 64; ++h) {
            const int j = h * 64 + lane;
            const float s = j < nj ? dtile[j * ldt + 4] * dtile[j * ldt + 5 + c] : 0.f;   // test_single_image.py:38
            const bool pass = j < nj && s >= thr;                        // nms_utils.py:22
            const u64 m = __ballot(pass);
            if (m == 0) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(cnt, __popcll(m));
            base = __shfl(base, 0);
            if (pass) keys[base + __popcll(m & ((1ull << lane) - 1))] = make_key<MODE>(s, (unsigned)(j0 + j));
        }
    }
}

// ---- 1c. the same for detection-style thresholds: box-level gate, no LDS tile ---------------------------------
// score = conf * prob with prob <= 1, so a box whose conf = sigmoid(t4) is below the threshold cannot contribute a
// candidate in ANY class (rounding to nearest is monotone: conf * prob <= conf).  One thread per box reads only the
// five leading logits of its row, decodes and stores the box, and votes; the wave then walks its active boxes one
// at a time, lanes = classes (coalesced row read), one atomic per candidate.  With a detector's usual threshold a
// few per cent of the boxes are active: the class logits of all others are never read.  Keys are the ones
// decode_compact_kernel produces (same expression), in a different order inside a (image, class) list, which the
// selection kernel does not depend on.  The host picks this kernel for score_thresh >= 0.05.
template <int MODE>
__global__ __launch_bounds__(256) void decode_sparse_kernel(DecodeK k, float thr, int* __restrict__ cand_count,
                                                            u64* __restrict__ cand_keys) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool valid = j < k.n;
    const int jc = valid ? j : k.n - 1;
    const int s = (jc >= k.off[1]) + (jc >= k.off[2]);
    const int local = jc - k.off[s];
    const int ns = k.off[s + 1] - k.off[s];
    const float* row = k.fm[s] + ((long long)b * ns + local) * k.A;
    float t[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i] = row[i];
    if (valid) {
        f32x4 box;
#pragma unroll
        for (int a = 0; a < 4; ++a) box[a] = decode_box_attr(k, s, local, t, a);         // model.py:76-108,147-152
        ((f32x4*)k.boxes)[(long long)b * k.n + j] = box;
    }
    const float conf = sigmoid_fast(t[4]);                                                // model.py:134
    u64 todo = __ballot(valid && conf >= thr);
    const int C = k.C;
    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const float* rowl = (const float*)(((unsigned long long)(unsigned)__shfl((int)((unsigned long long)row >> 32), l) << 32) |
                                           (unsigned)__shfl((int)(unsigned long long)row, l));
        const float confl = __shfl(conf, l);
        const unsigned jl = (unsigned)__shfl(j, l);
        for (int c = lane; c < C; c += 64) {
            const float sc = confl * sigmoid_fast(rowl[5 + c]);                           // model.py:135, test_single_image.py:38
            if (sc >= thr) {                                                              // nms_utils.py:22
                const int pos = atomicAdd(cand_count + (long long)b * C + c, 1);
                cand_keys[((long long)b * C + c) * k.n + pos] = make_key<MODE>(sc, jl);
            }
        }
    }
}

// ---- 2. selection ------------------------------------------------------------------------------
// One workgroup per (image, class).  The candidates are consumed in key order (best first) in ROUNDS of at most
// kSortCap keys: a round takes the key range [lo, hi), sorts it in LDS and lets wave 0 walk it.  Greedy NMS only ever
// looks back (a candidate is judged against better-ranked ones), so the first max_boxes survivors of the whole list
// are found as soon as a prefix of it has produced them - with k = 22743 candidates per class (score_thresh 0.001)
// the first round normally does.  hi comes from a radix histogram of the keys >= lo: 12 key bits per level, descending
// into the first non-empty bin while that bin alone holds more than kSortCap keys (equal scores differ only in the
// index bits); keys are unique, so the last level always separates them.
template <int MODE>
__device__ __forceinline__ void nms_select_block(const int bc, const float* __restrict__ boxes, int n, int C, int max_boxes,
                                                 float iou_thr, const int* __restrict__ cand_count,
                                                 u64* __restrict__ cand_keys, int* __restrict__ sel_count,
                                                 int* __restrict__ sel_idx, float* __restrict__ sel_score, f32x4* selbox,
                                                 u64* skeys) {
    // skeys[kSortCap] (16 KB, the kernel's): the round's keys; before they are gathered, the radix histogram
    __shared__ u64 masks[64];
    __shared__ int part[256];
    int* const hist = (int*)skeys;             // 4096 bins = the same 16 KB; never live together with the keys
    static_assert(sizeof(u64) * kSortCap >= sizeof(int) * 4096, "histogram must fit the key buffer");
    __shared__ int s_take, s_first, s_cnt, s_nsel;

    __syncthreads();                           // the shared arrays are free again (previous class of this workgroup)
    const int b = bc / C;                      // bc = b * C + c
    const int k = cand_count[bc];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* out_idx = sel_idx + (long long)bc * max_boxes;
    float* out_sc = sel_score ? sel_score + (long long)bc * max_boxes : nullptr;   // scores live only in the keys (fused path)
    const f32x4* bx = (const f32x4*)boxes + (long long)b * n;
    const u64* keys = cand_keys + (long long)bc * n;
    if (k == 0) {
        if (tid == 0) sel_count[bc] = 0;
        return;
    }

    int nsel = 0;
    u64 lo = 0;                                // every key < lo has been walked
    while (true) {
        // ---- the round's key range [lo, hi) and its keys -> skeys[0, cnt)
        u64 hi = kDead;
        if (k > kSortCap) {
            u64 prefix = 0, pmask = 0;         // keys under consideration: key >= lo and (key & pmask) == prefix
            int shift = 52, bits = 12;
            while (true) {
                const int nbins = 1 << bits;
                for (int i = tid; i < nbins; i += 256) hist[i] = 0;
                if (tid == 0) { s_take = -1; s_first = nbins; }
                __syncthreads();
                for (int i = tid; i < k; i += 256) {
                    const u64 key = keys[i];
                    if (key >= lo && (key & pmask) == prefix) atomicAdd(&hist[(int)((key >> shift) & (u64)(nbins - 1))], 1);
                }
                __syncthreads();
                // inclusive prefix sums over the bins: thread t owns bins [t * per, (t + 1) * per)
                const int per = nbins >= 256 ? nbins / 256 : 1;
                const int nown = nbins >= 256 ? 256 : nbins;
                int mine = 0;
                if (tid < nown)
                    for (int j = 0; j < per; ++j) mine += hist[tid * per + j];
                part[tid] = mine;
                __syncthreads();
                if (tid == 0) {
                    int acc = 0;
                    for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = acc; acc += v; }   // exclusive
                }
                __syncthreads();
                if (tid < nown) {
                    int acc = part[tid];
                    for (int j = 0; j < per; ++j) {
                        const int h = hist[tid * per + j];
                        acc += h;
                        if (h > 0) {
                            atomicMin(&s_first, tid * per + j);
                            if (acc <= kSortCap) atomicMax(&s_take, tid * per + j);       // bins 0..this one fit a round
                        }
                    }
                }
                __syncthreads();
                const int take = s_take, first = s_first;
                __syncthreads();
                if (first == nbins) { hi = lo; break; }                                    // no key >= lo left
                if (take >= 0) {                                                           // whole bins [.., take] fit
                    hi = prefix + ((u64)(take + 1) << shift);
                    if ((shift == 52 && take + 1 == nbins) || hi < prefix) hi = kDead;      // past the largest key
                    break;
                }
                // the first non-empty bin alone exceeds a round: split it by the next key bits
                prefix |= (u64)first << shift;
                pmask |= (u64)(nbins - 1) << shift;
                if (shift >= 12) { shift -= 12; bits = 12; }
                else { bits = shift; shift = 0; }                                          // 64 = 5 x 12 + 4
            }
            if (hi == lo) break;
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            for (int i = tid; i < k; i += 256) {
                const u64 key = keys[i];
                if (key >= lo && key < hi) skeys[atomicAdd(&s_cnt, 1)] = key;
            }
            __syncthreads();
        } else {
            for (int i = tid; i < k; i += 256) skeys[i] = keys[i];
            if (tid == 0) s_cnt = k;
            __syncthreads();
        }
        const int cnt = s_cnt;
        int np2 = 64;
        while (np2 < cnt) np2 <<= 1;
        for (int i = cnt + tid; i < np2; i += 256) skeys[i] = kDead;
        __syncthreads();
        for (int size = 2; size <= np2; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (np2 >> 1); t += 256) {
                    const int l0 = 2 * t - (t & (stride - 1));
                    const int h0 = l0 + stride;
                    const bool up = (l0 & size) == 0;
                    const u64 x = skeys[l0], y = skeys[h0];
                    if ((x > y) == up) { skeys[l0] = y; skeys[h0] = x; }
                }
                __syncthreads();
            }
        if (wave == 0) {
            // chunked greedy walk over the sorted candidates of the round
            for (int c0 = 0; c0 < cnt && nsel < max_boxes; c0 += 64) {
                const int i = c0 + lane;
                const bool valid = i < cnt;
                const u64 mykey = valid ? skeys[i] : 0ull;
                const unsigned idx = valid ? key_index<MODE>(mykey) : 0u;
                f32x4 box = {0.f, 0.f, 0.f, 0.f};
                if (valid) box = bx[idx];
                bool alive = valid;
                for (int sidx = 0; sidx < nsel; ++sidx)
                    if (alive && suppresses<MODE>(selbox[sidx], box, iou_thr)) alive = false;
                u64 mask = 0;                       // bit j: chunk member j (better ranked) suppresses me
                for (int j = 0; j < 63; ++j) {
                    f32x4 o;
                    o[0] = __shfl(box[0], j); o[1] = __shfl(box[1], j); o[2] = __shfl(box[2], j); o[3] = __shfl(box[3], j);
                    if (j < lane && suppresses<MODE>(o, box, iou_thr)) mask |= 1ull << j;
                }
                masks[lane] = mask;
                const u64 alive_mask = __ballot(alive);
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): masks[] written by this wave only
                __builtin_amdgcn_wave_barrier();
                u64 selected = 0;
                int cntsel = nsel;
                for (int t = 0; t < 64; ++t) {      // suppression scan (wave-uniform)
                    if (((alive_mask >> t) & 1) && cntsel < max_boxes && (masks[t] & selected) == 0) {
                        selected |= 1ull << t;
                        ++cntsel;
                    }
                }
                if ((selected >> lane) & 1) {
                    const int pos = nsel + __popcll(selected & ((1ull << lane) - 1));
                    selbox[pos] = box;
                    out_idx[pos] = (int)idx;
                    if (out_sc) out_sc[pos] = key_score(mykey);
                }
                nsel = cntsel;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) s_nsel = nsel;
        }
        __syncthreads();
        nsel = s_nsel;
        if (nsel >= max_boxes || hi == kDead) break;      // enough boxes, or that was the last key range
        lo = hi;
    }
    if (tid == 0) sel_count[bc] = nsel;
}

// The kernel: a workgroup takes four (image, class) lists.  With a detector's threshold nearly all of them hold a handful of
// candidates, so each WAVE settles one list of <= 64 candidates by itself - keys ranked by 64 wave shuffles (keys are
// unique), placed through a 64-entry LDS row, then the same single-chunk greedy step as above (IoU bit masks by shuffles,
// wave-uniform suppression scan): no workgroup barrier, and four lists per workgroup instead of one with three idle waves.
// Lists with more than 64 candidates go through nms_select_block, one workgroup per list (the second part of the grid: a
// workgroup per list that exits at once when its list is short - the stress configuration keeps its 640 parallel workgroups).
// Same keys, same order, same comparisons: the selected indices are the ones the block path gives.
template <int MODE>
__global__ __launch_bounds__(256) void nms_select_kernel(const float* __restrict__ boxes, int n, int C, int nbc, int max_boxes,
                                                         float iou_thr, const int* __restrict__ cand_count,
                                                         u64* __restrict__ cand_keys, int* __restrict__ sel_count,
                                                         int* __restrict__ sel_idx, float* __restrict__ sel_score) {
    extern __shared__ f32x4 selbox[];          // [max_boxes] selected boxes of the block path (dynamic: ~18 KB + 16 max_boxes bytes
                                               // of LDS per workgroup, 8 workgroups per CU at max_boxes 50)
    // one 16 KB array for both paths (a workgroup runs exactly one of them): the block path's key buffer; the wave path's
    // rank rows wsort[4][64] and mask rows wmask[4][64] in its first 4 KB
    __shared__ u64 skeys[kSortCap];
    u64 (*const wsort)[64] = (u64 (*)[64])skeys;
    u64 (*const wmask)[64] = (u64 (*)[64])(skeys + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsmall = (nbc + 3) >> 2;              // workgroups [0, nsmall): four short lists each; [nsmall, nsmall + nbc): one
    if ((int)blockIdx.x >= nsmall) {                // workgroup per list, which leaves at once unless the list is a long one
        const int bc = (int)blockIdx.x - nsmall;
        if (cand_count[bc] > 64)
            nms_select_block<MODE>(bc, boxes, n, C, max_boxes, iou_thr, cand_count, cand_keys, sel_count, sel_idx, sel_score, selbox, skeys);
        return;
    }
    const int bc0 = blockIdx.x * 4;
    {
        const int bc = bc0 + wave;
        const int k = bc < nbc ? cand_count[bc] : -1;
        if (k == 0) {
            if (lane == 0) sel_count[bc] = 0;
        } else if (k > 0 && k <= 64) {
            const int b = bc / C;
            const f32x4* bx = (const f32x4*)boxes + (long long)b * n;
            const u64 key = lane < k ? cand_keys[(long long)bc * n + lane] : kDead;
            int rank = 0;
            for (int j = 0; j < k; ++j) {
                const u64 kj = ((u64)(unsigned)__shfl((int)(key >> 32), j) << 32) | (unsigned)__shfl((int)key, j);
                rank += kj < key;
            }
            if (lane < k) wsort[wave][rank] = key;
            __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): the row is written and read by this wave only
            __builtin_amdgcn_wave_barrier();
            const bool valid = lane < k;
            const u64 mykey = valid ? wsort[wave][lane] : 0ull;
            const unsigned idx = valid ? key_index<MODE>(mykey) : 0u;
            f32x4 box = {0.f, 0.f, 0.f, 0.f};
            if (valid) box = bx[idx];
            u64 mask = 0;                           // bit j: the better-ranked candidate j suppresses me
            for (int j = 0; j + 1 < k; ++j) {
                f32x4 o;
                o[0] = __shfl(box[0], j); o[1] = __shfl(box[1], j); o[2] = __shfl(box[2], j); o[3] = __shfl(box[3], j);
                if (j < lane && suppresses<MODE>(o, box, iou_thr)) mask |= 1ull << j;
            }
            wmask[wave][lane] = mask;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            u64 selected = 0;
            int cntsel = 0;
            for (int t = 0; t < k; ++t) {           // suppression scan (wave-uniform)
                if (cntsel < max_boxes && (wmask[wave][t] & selected) == 0) {
                    selected |= 1ull << t;
                    ++cntsel;
                }
            }
            if ((selected >> lane) & 1) {
                const int pos = __popcll(selected & ((1ull << lane) - 1));
                sel_idx[(long long)bc * max_boxes + pos] = (int)idx;
                if (sel_score) sel_score[(long long)bc * max_boxes + pos] = key_score(mykey);
            }
            if (lane == 0) sel_count[bc] = cntsel;
        }
    }
}

// ---- 3. class-ordered dense output ---------------------------------------------------------------
__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                         int n, int C, int max_boxes, const int* __restrict__ sel_count,
                                                         const int* __restrict__ sel_idx, const float* __restrict__ sel_score,
                                                         float* __restrict__ out_boxes,
                                                         float* __restrict__ out_scores, int* __restrict__ out_labels,
                                                         int* __restrict__ out_index, int* __restrict__ out_counts) {
    extern __shared__ int prefix[];            // [C + 1]
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) { prefix[c] = acc; acc += sel_count[(long long)b * C + c]; }
        prefix[C] = acc;
        out_counts[b] = acc;
    }
    __syncthreads();
    const long long row0 = (long long)b * C * max_boxes;
    for (int e = threadIdx.x; e < C * max_boxes; e += 256) {
        const int c = e / max_boxes, r = e - c * max_boxes;
        if (r >= prefix[c + 1] - prefix[c]) continue;
        const int idx = sel_idx[((long long)b * C + c) * max_boxes + r];
        const long long o = row0 + prefix[c] + r;
        ((f32x4*)out_boxes)[o] = ((const f32x4*)boxes)[(long long)b * n + idx];      // nms_utils.py:32
        out_scores[o] = scores ? scores[((long long)b * n + idx) * C + c]            // nms_utils.py:33
                               : sel_score[((long long)b * C + c) * max_boxes + r];
        out_labels[o] = c;                                                           // nms_utils.py:31
        if (out_index) out_index[o] = idx;
    }
}

struct NmsWs { int* cand_count; int* sel_count; int* sel_idx; u64* cand_keys; size_t total; float* sel_score; float* boxes; };

static NmsWs nms_layout(void* base, int B, int n, int C, int max_boxes) {
    NmsWs w;
    char* p = (char*)base;
    size_t off = 0;
    const size_t bc = (size_t)B * C;
    w.cand_count = (int*)(p + off); off = align_up(off + bc * 4, 256);
    w.sel_count = (int*)(p + off); off = align_up(off + bc * 4, 256);
    w.sel_idx = (int*)(p + off); off = align_up(off + bc * (size_t)max_boxes * 4, 256);
    w.cand_keys = (u64*)(p + off); off = align_up(off + bc * (size_t)n * 8, 256);
    w.total = off;
    w.sel_score = nullptr;
    w.boxes = nullptr;
    return w;
}

// yolo_detect_post: the NMS arrays + selected scores + decoded boxes [B,n,4]
static NmsWs post_layout(void* base, int B, int n, int C, int max_boxes) {
    NmsWs w = nms_layout(base, B, n, C, max_boxes);
    char* p = (char*)base;
    size_t off = w.total;
    w.sel_score = (float*)(p + off); off = align_up(off + (size_t)B * C * max_boxes * 4, 256);
    w.boxes = (float*)(p + off); off = align_up(off + (size_t)B * n * 16, 256);
    w.total = off;
    return w;
}

template <int MODE>
static int nms_run(const float* boxes, const float* scores, int B, int n, int C, int max_boxes, float st, float it,
                   float* ob, float* os, int* ol, int* oi, int* oc, const NmsWs& w, hipStream_t stream) {
    YOLO_HIP_CHECK(hipMemsetAsync(w.cand_count, 0, (size_t)B * C * 4, stream));
    const unsigned tiles = (unsigned)((n + kTileJ - 1) / kTileJ);
    const unsigned ctiles = (unsigned)((C + kTileC - 1) / kTileC);
    hipLaunchKernelGGL(nms_compact_kernel<MODE>, dim3(tiles, B, ctiles), dim3(256), 0, stream, scores, n, C, st, w.cand_count, w.cand_keys);
    YOLO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(nms_select_kernel<MODE>, dim3((unsigned)((B * C + 3) / 4 + B * C)), dim3(256), (size_t)max_boxes * 16, stream, boxes, n, C, B * C, max_boxes, it,
                       w.cand_count, w.cand_keys, w.sel_count, w.sel_idx, (float*)nullptr);
    YOLO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(nms_gather_kernel, dim3(B), dim3(256), (size_t)(C + 1) * 4, stream, boxes, scores, n, C, max_boxes,
                       w.sel_count, w.sel_idx, (const float*)nullptr, ob, os, ol, oi, oc);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

// decode + score + compaction fused, then the same selection / gather kernels (TF semantics = gpu_nms)
static int post_run(DecodeK k, int B, int max_boxes, float st, float it, float* ob, float* os, int* ol, int* oi, int* oc,
                    const NmsWs& w, hipStream_t stream) {
    const int n = k.n, C = k.C;
    YOLO_HIP_CHECK(hipMemsetAsync(w.cand_count, 0, (size_t)B * C * 4, stream));
    const unsigned tiles = (unsigned)((n + kTileJ - 1) / kTileJ);
    const size_t lds = (size_t)kTileJ * (k.A | 1) * 4;
    auto fn = decode_compact_kernel<YOLO_NMS_TF>;
    static std::atomic<bool> attr_done[16];   // zero-initialised; setting the attribute twice is harmless, the flag only must not tear
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 16 || !attr_done[dev]) {
        YOLO_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev < 16) attr_done[dev] = true;
    }
    k.boxes = w.boxes;
    if (st >= 0.05f)      // box-level conf gate pays when few boxes pass; below, the dense tile kernel (1 atomic per 64 boxes)
        hipLaunchKernelGGL(decode_sparse_kernel<YOLO_NMS_TF>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, stream, k, st,
                           w.cand_count, w.cand_keys);
    else
        hipLaunchKernelGGL(fn, dim3(tiles, B), dim3(256), lds, stream, k, st, w.cand_count, w.cand_keys);
    YOLO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(nms_select_kernel<YOLO_NMS_TF>, dim3((unsigned)((B * C + 3) / 4 + B * C)), dim3(256), (size_t)max_boxes * 16, stream, w.boxes, n, C, B * C, max_boxes, it,
                       w.cand_count, w.cand_keys, w.sel_count, w.sel_idx, w.sel_score);
    YOLO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(nms_gather_kernel, dim3(B), dim3(256), (size_t)(C + 1) * 4, stream, w.boxes, (const float*)nullptr, n, C,
                       max_boxes, w.sel_count, w.sel_idx, w.sel_score, ob, os, ol, oi, oc);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

}  // namespace yolo

extern "C" size_t yolo_nms_workspace_bytes(int B, int n, int class_num, int max_boxes) {
    if (B <= 0 || n <= 0 || class_num <= 0 || max_boxes <= 0) return 0;
    if (max_boxes > yolo::kSelCap || class_num > 4096 || B > 65535) return 0;      // the limits yolo_nms enforces
    return yolo::nms_layout(nullptr, B, n, class_num, max_boxes).total;
}

extern "C" int yolo_nms(const float* boxes, const float* scores, int B, int n, int class_num, int max_boxes,
                        float score_thresh, float iou_thresh, int mode, float* out_boxes, float* out_scores,
                        int32_t* out_labels, int32_t* out_index, int32_t* out_counts, void* workspace,
                        size_t workspace_bytes, void* stream) {
    using namespace yolo;
    YOLO_REQUIRE(boxes && scores && out_boxes && out_scores && out_labels && out_counts && workspace, YOLO_ERR_ARG, "nms: null pointer");
    YOLO_REQUIRE(B > 0 && n > 0 && class_num > 0 && max_boxes > 0, YOLO_ERR_ARG, "nms: non-positive size");
    YOLO_REQUIRE(max_boxes <= kSelCap, YOLO_ERR_UNSUPPORTED, "nms: max_boxes %d > %d", max_boxes, kSelCap);
    YOLO_REQUIRE(class_num <= 4096 && B <= 65535, YOLO_ERR_UNSUPPORTED, "nms: class_num/B out of range");
    YOLO_REQUIRE((long long)B * class_num < (1ll << 31) && (long long)B * n < (1ll << 31), YOLO_ERR_SHAPE, "nms: problem too large");
    YOLO_REQUIRE(mode == YOLO_NMS_TF || mode == YOLO_NMS_PY, YOLO_ERR_ARG, "nms: bad mode %d", mode);
    const NmsWs w = nms_layout(workspace, B, n, class_num, max_boxes);
    YOLO_REQUIRE(workspace_bytes >= w.total, YOLO_ERR_WORKSPACE, "nms: workspace %zu < %zu bytes", workspace_bytes, w.total);
    YOLO_REQUIRE(((uintptr_t)workspace & 255) == 0, YOLO_ERR_ARG, "nms: workspace must be 256-byte aligned");
    if (mode == YOLO_NMS_TF)
        return nms_run<YOLO_NMS_TF>(boxes, scores, B, n, class_num, max_boxes, score_thresh, iou_thresh, out_boxes, out_scores,
                                    out_labels, out_index, out_counts, w, (hipStream_t)stream);
    return nms_run<YOLO_NMS_PY>(boxes, scores, B, n, class_num, max_boxes, score_thresh, iou_thresh, out_boxes, out_scores,
                                out_labels, out_index, out_counts, w, (hipStream_t)stream);
}

// ---- yolo_nms for callers that cannot allocate -----------------------------------------------------------------
// The candidate / selection arrays of nms_layout live in a file-scope device array instead of a caller workspace.
// Every GPU the library runs on holds its own copy at its own address; `scratch` itself is only the host-side
// stand-in of the symbol, so host code uses the per-device address from hipGetSymbolAddress, never the name.
namespace yolo {
constexpr size_t kStaticScratchBytes = YOLO_NMS_STATIC_BYTES;      // 16 MiB: one 416 or 608 image with 80 classes, or
                                                              // any B * n * class_num * 8 + small change below it
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static void* g_scratch_ptr[64] = {nullptr};
static std::mutex g_scratch_mu;

static int static_scratch(void** out) {
    int dev = 0;
    YOLO_HIP_CHECK(hipGetDevice(&dev));
    YOLO_REQUIRE(dev >= 0 && dev < 64, YOLO_ERR_UNSUPPORTED, "nms_static: device ordinal %d not below 64", dev);
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {                                // once per device, with that device current
        void* ptr = nullptr;
        YOLO_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        YOLO_REQUIRE(ptr != nullptr && ((uintptr_t)ptr & 255) == 0, YOLO_ERR_HIP, "nms_static: bad scratch address %p", ptr);
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return YOLO_OK;
}
}  // namespace yolo

extern "C" int yolo_nms_static(const float* boxes, const float* scores, int B, int n, int class_num, int max_boxes,
                               float score_thresh, float iou_thresh, int mode, float* out_boxes, float* out_scores,
                               int32_t* out_labels, int32_t* out_index, int32_t* out_counts, void* stream) {
    using namespace yolo;
    YOLO_REQUIRE(boxes && scores && out_boxes && out_scores && out_labels && out_counts, YOLO_ERR_ARG, "nms_static: null pointer");
    YOLO_REQUIRE(B > 0 && n > 0 && class_num > 0 && max_boxes > 0, YOLO_ERR_ARG, "nms_static: non-positive size");
    YOLO_REQUIRE(max_boxes <= kSelCap, YOLO_ERR_UNSUPPORTED, "nms_static: max_boxes %d > %d", max_boxes, kSelCap);
    YOLO_REQUIRE(class_num <= 4096 && B <= 65535, YOLO_ERR_UNSUPPORTED, "nms_static: class_num/B out of range");
    YOLO_REQUIRE((long long)B * class_num < (1ll << 31) && (long long)B * n < (1ll << 31), YOLO_ERR_SHAPE, "nms_static: problem too large");
    YOLO_REQUIRE(mode == YOLO_NMS_TF || mode == YOLO_NMS_PY, YOLO_ERR_ARG, "nms_static: bad mode %d", mode);
    const size_t need = nms_layout(nullptr, B, n, class_num, max_boxes).total;
    YOLO_REQUIRE(need <= kStaticScratchBytes, YOLO_ERR_UNSUPPORTED,
                 "nms_static: B=%d n=%d class_num=%d max_boxes=%d needs %zu bytes of scratch, the built-in buffer holds %zu; "
                 "use yolo_nms with a workspace", B, n, class_num, max_boxes, need, kStaticScratchBytes);
    void* ptr = nullptr;
    const int rc = static_scratch(&ptr);
    if (rc != YOLO_OK) return rc;
    const NmsWs w = nms_layout(ptr, B, n, class_num, max_boxes);     // cand_count is cleared per call by nms_run
    if (mode == YOLO_NMS_TF)
        return nms_run<YOLO_NMS_TF>(boxes, scores, B, n, class_num, max_boxes, score_thresh, iou_thresh, out_boxes, out_scores,
                                    out_labels, out_index, out_counts, w, (hipStream_t)stream);
    return nms_run<YOLO_NMS_PY>(boxes, scores, B, n, class_num, max_boxes, score_thresh, iou_thresh, out_boxes, out_scores,
                                out_labels, out_index, out_counts, w, (hipStream_t)stream);
}

// ---- detections -> wire rows -------------------------------------------------------------------------
// The step after gpu_nms in the reference's drivers: boxes back to the original image's pixels
// (yolo_v3/test_single_image.py:42-45, video_test.py:67-71: boxes_[:, 0|2] *= width_ori / new_w, [:, 1|3] *= height_ori / new_h,
// a float32 multiply by the float32-rounded ratio) and one dense list for the whole batch, rows
// (image index, x0, y0, x1, y1, score, label) in image order - the format the shards of a multi-GPU run gather.
namespace yolo {
__global__ __launch_bounds__(256) void pack_offsets_kernel(const int* __restrict__ counts, int B, int cap, int* __restrict__ offsets) {
    __shared__ int part[256];
    __shared__ int carry;
    if (threadIdx.x == 0) { carry = 0; offsets[0] = 0; }
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 256) {                      // blocked inclusive scan, 256 images at a time
        const int b = b0 + threadIdx.x;
        int c = b < B ? counts[b] : 0;
        c = c < 0 ? 0 : (c > cap ? cap : c);
        part[threadIdx.x] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (b < B) offsets[b + 1] = carry + part[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                        const int* __restrict__ labels, const int* __restrict__ offsets, int cap,
                                                        const float* __restrict__ scale_xy, int image_base, float* __restrict__ rows) {
    const int b = blockIdx.x;
    const int lo = offsets[b], m = offsets[b + 1] - lo;
    const float sx = scale_xy ? scale_xy[2 * b] : 1.0f, sy = scale_xy ? scale_xy[2 * b + 1] : 1.0f;
    for (int r = threadIdx.x; r < m; r += 256) {
        const long long src = (long long)b * cap + r;
        const f32x4 bx = ((const f32x4*)boxes)[src];
        float* o = rows + (long long)(lo + r) * 7;
        o[0] = (float)(image_base + b);
        o[1] = scale_xy ? __fmul_rn(bx[0], sx) : bx[0];
        o[2] = scale_xy ? __fmul_rn(bx[1], sy) : bx[1];
        o[3] = scale_xy ? __fmul_rn(bx[2], sx) : bx[2];
        o[4] = scale_xy ? __fmul_rn(bx[3], sy) : bx[3];
        o[5] = scores[src];
        o[6] = (float)labels[src];
    }
}
}  // namespace yolo

extern "C" int yolo_pack_detections(const float* boxes, const float* scores, const int32_t* labels, const int32_t* counts, int B,
                                    int cap, const float* scale_xy, int image_base, float* rows, int32_t* offsets, void* stream) {
    using namespace yolo;
    YOLO_REQUIRE(boxes && scores && labels && counts && rows && offsets, YOLO_ERR_ARG, "pack_detections: null pointer");
    YOLO_REQUIRE(B > 0 && cap > 0 && B <= (1 << 20) && image_base >= 0, YOLO_ERR_ARG, "pack_detections: bad size");
    YOLO_REQUIRE(((uintptr_t)boxes & 15) == 0, YOLO_ERR_ARG, "pack_detections: boxes must be 16-byte aligned (one box per load)");
    YOLO_REQUIRE((long long)B * cap < (1ll << 31) / 8 && (long long)image_base + B < (1 << 24), YOLO_ERR_SHAPE,
                 "pack_detections: problem too large (image indices are carried as float32)");
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, B, cap, offsets);
    YOLO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, boxes, scores, labels, offsets, cap,
                       scale_xy, image_base, rows);
    YOLO_HIP_CHECK(hipGetLastError());
    return YOLO_OK;
}

static int post_shape(int H, int W) { return 3 * ((H / 32) * (W / 32) + (H / 16) * (W / 16) + (H / 8) * (W / 8)); }

extern "C" size_t yolo_detect_post_workspace_bytes(int B, int H, int W, int class_num, int max_boxes) {
    if (B <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32 || class_num <= 0 || max_boxes <= 0) return 0;
    if (max_boxes > yolo::kSelCap || class_num > 250 || B > 65535) return 0;       // the limits yolo_detect_post enforces
    return yolo::post_layout(nullptr, B, post_shape(H, W), class_num, max_boxes).total;
}

extern "C" int yolo_detect_post(const float* fm1, const float* fm2, const float* fm3, int B, int H, int W, int class_num,
                                const float* anchors, int max_boxes, float score_thresh, float iou_thresh, float* out_boxes,
                                float* out_scores, int32_t* out_labels, int32_t* out_index, int32_t* out_counts,
                                void* workspace, size_t workspace_bytes, void* stream) {
    using namespace yolo;
    YOLO_REQUIRE(fm1 && fm2 && fm3 && anchors && out_boxes && out_scores && out_labels && out_counts && workspace, YOLO_ERR_ARG,
                 "detect_post: null pointer");
    YOLO_REQUIRE(B > 0 && B <= 65535 && class_num > 0 && max_boxes > 0, YOLO_ERR_ARG, "detect_post: non-positive size");
    YOLO_REQUIRE(H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, YOLO_ERR_SHAPE, "detect_post: H, W must be positive multiples of 32 (got %d x %d)", H, W);
    YOLO_REQUIRE(max_boxes <= kSelCap, YOLO_ERR_UNSUPPORTED, "detect_post: max_boxes %d > %d", max_boxes, kSelCap);
    YOLO_REQUIRE(class_num <= 250, YOLO_ERR_UNSUPPORTED, "detect_post: class_num %d > 250 (LDS tile); use yolo_decode + yolo_nms", class_num);
    DecodeK k;
    k.fm[0] = fm1; k.fm[1] = fm2; k.fm[2] = fm3;
    k.boxes = nullptr; k.confs = nullptr; k.probs = nullptr; k.scores = nullptr;
    decode_setup(k, H, W, class_num, anchors);
    YOLO_REQUIRE((long long)k.n * k.A < (1ll << 24) && (long long)B * k.n < (1ll << 31), YOLO_ERR_SHAPE, "detect_post: problem too large");
    const NmsWs w = post_layout(workspace, B, k.n, class_num, max_boxes);
    YOLO_REQUIRE(workspace_bytes >= w.total, YOLO_ERR_WORKSPACE, "detect_post: workspace %zu < %zu bytes", workspace_bytes, w.total);
    YOLO_REQUIRE(((uintptr_t)workspace & 255) == 0, YOLO_ERR_ARG, "detect_post: workspace must be 256-byte aligned");
    return post_run(k, B, max_boxes, score_thresh, iou_thresh, out_boxes, out_scores, out_labels, out_index, out_counts, w,
                    (hipStream_t)stream);
}
