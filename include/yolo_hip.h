/* yolo_hip.h - C-ABI of libyolo_hip.so: the MI355X (gfx950) YOLOv3 inference hot path.
 *
 * The reference (FenHua/yolo, yolo_v3/) has no FFI of its own: its boundary is a set of Python
 * functions that build TF-1 graphs.  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference checkout).  The Python mirror in yolo_amd/ binds
 * these with ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - All tensor pointers are DEVICE pointers (hipMalloc / torch-ROCm memory) unless marked HOST.
 *  - Caller allocates inputs, outputs and workspaces; the library keeps no reference to them after
 *    the call returns.  Only the weight handle (YoloNet / YoloConv) owns device memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every entry enqueues
 *    asynchronously and never synchronises the device.
 *  - Layout NHWC, float32 at the boundary (images RGB in [0,1]; boxes x_min,y_min,x_max,y_max in
 *    network-input pixels; labels int32) exactly like the reference.
 *  - Return value: YOLO_OK or a negative YOLO_ERR_*; yolo_last_error() gives a message (thread-local).
 */
#ifndef YOLO_HIP_H_
#define YOLO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    YOLO_OK = 0,
    YOLO_ERR_ARG = -1,         /* null pointer / non-positive size */
    YOLO_ERR_SHAPE = -2,       /* shape not supported (H,W must be multiples of 32, ...) */
    YOLO_ERR_HIP = -3,         /* a HIP runtime call failed */
    YOLO_ERR_UNSUPPORTED = -4, /* parameter outside the implemented range */
    YOLO_ERR_WORKSPACE = -5    /* workspace too small */
};

enum { YOLO_DTYPE_F32 = 0, YOLO_DTYPE_BF16 = 1 };   /* arithmetic type of the conv stack */
enum { YOLO_NMS_TF = 0, YOLO_NMS_PY = 1 };          /* gpu_nms (TF op) / cpu_nms+py_nms semantics */

typedef struct YoloNet YoloNet;
typedef struct YoloConv YoloConv;

const char* yolo_last_error(void);
int yolo_abi_version(void);

/* ---- whole network: replaces yolov3.forward (yolo_v3/model.py:21-67), i.e. darknet53_body +
 * yolo_block + upsample_layer + concat (yolo_v3/utils/layer_utils.py:9-72), and the weight
 * ingestion of load_weights (yolo_v3/utils/misc_utils.py:45-96).
 * darknet_stream: HOST float32 body of a Darknet .weights file (after the 5 x int32 header), conv
 * order = call order of forward; n_floats must match the architecture exactly. */
int yolo_net_create(int class_num, int dtype, const float* darknet_stream, size_t n_floats, YoloNet** out);
int yolo_net_destroy(YoloNet* net);
/* Bytes of activation workspace forward needs for a [B,H,W,3] input (0 on bad shape). */
size_t yolo_net_workspace_bytes(const YoloNet* net, int B, int H, int W);
/* images [B,H,W,3] f32 -> fm1 [B,H/32,W/32,3*(5+C)], fm2 [B,H/16,W/16,.], fm3 [B,H/8,W/8,.] f32.
 * Ordered on `stream` like a single launch.  By the work of the call in 416 x 416 images (B * H * W / 416^2): from 26 up the
 * images run as two, from 56 to 89 as three, from 90 up as two again sub-batches ("lanes"), one on `stream` and the others on streams the handle owns,
 * forked and joined with events; the result does not depend on the split.  YOLO_LANES=<1..8> in the environment overrides
 * the lane count (1: none; never more lanes than images).
 * One forward at a time per handle: concurrent calls on one handle would share its side streams and plans. */
int yolo_net_forward(YoloNet* net, const float* images, int B, int H, int W, void* workspace,
                     size_t workspace_bytes, float* fm1, float* fm2, float* fm3, void* stream);
/* Small-batch ("latency") plan, bf16 handles, no counterpart in the reference (whose published operating point is batch 1,
 * yolo_v3/README.md:66).  At batch 1 - 4 a layer's grid covers a fraction of the 256 CUs and its time is the chain of its K-steps;
 * with the plan on, every 3x3 layer (and 1x1 layer of K >= 1024) whose 128 x 128 tiles would occupy fewer than half the CUs runs its K loop in S slices
 * on S times as many workgroups (f32 partial tiles in the workspace) followed by one launch that adds the slices in order and
 * applies bias / activation / shortcut.  Deterministic, within the bf16 tolerance of the default plan but NOT bit-identical to it
 * (the f32 accumulation order differs), which is why it is opt-in: the default plan gives every image the same bits whatever
 * batch it is part of.  Changes yolo_net_workspace_bytes / yolo_net_detect_workspace_bytes (+32 MiB per lane): query them again.
 * Call it while no forward of the handle is in flight. */
int yolo_net_set_small_batch_plan(YoloNet* net, int on);
/* Number of conv layers of the plan (75 for YOLOv3) and the static shape of layer i. */
int yolo_net_num_layers(const YoloNet* net);
int yolo_net_layer_info(const YoloNet* net, int layer, int* cin, int* cout, int* ksize, int* stride);
/* Measurement aid (no counterpart in the reference): yolo_net_forward with a hipEvent recorded on
 * `stream` before every layer; SYNCHRONISES the stream, then writes per-layer device milliseconds to
 * HOST layer_ms[n_layers] (n_layers must equal yolo_net_num_layers). */
int yolo_net_forward_timed(YoloNet* net, const float* images, int B, int H, int W, void* workspace,
                           size_t workspace_bytes, float* fm1, float* fm2, float* fm3, void* stream,
                           float* layer_ms, int n_layers);

/* Measurement aid (no counterpart in the reference): name of the kernel instantiation the last
 * yolo_net_forward_timed launched for `layer`, spelled as rocprofv3 prints it (no return type / arguments). */
const char* yolo_net_layer_kernel(const YoloNet* net, int layer);

/* ---- one conv layer: replaces conv2d (yolo_v3/utils/layer_utils.py:9-21) under the arg_scope of
 * yolo_v3/model.py:32-36 (conv no-bias + BN(eps 1e-5, inference) + leaky 0.1), or, with bn == 0,
 * the detection conv of model.py:42-44 (conv + bias, no activation).
 * weights_hwio: HOST [k,k,Cin,Cout] f32 (TF variable layout).  bn != 0: p0..p3 = beta, gamma,
 * moving_mean, moving_variance (HOST, [Cout]); bn == 0: p0 = bias, p1..p3 ignored. */
int yolo_conv_create(int dtype, int cin, int cout, int ksize, int stride, int bn,
                     const float* weights_hwio, const float* p0, const float* p1, const float* p2,
                     const float* p3, YoloConv** out);
int yolo_conv_destroy(YoloConv* conv);
/* in: [B,H,W,Cin] f32; residual (optional, same shape as out, added AFTER the activation like
 * res_block, layer_utils.py:24-29) -> out [B,Ho,Wo,Cout] f32.  In bf16 mode the input is rounded to
 * bf16 first (scratch: workspace of yolo_conv_workspace_bytes). */
size_t yolo_conv_workspace_bytes(const YoloConv* conv, int B, int H, int W);
int yolo_conv_forward(YoloConv* conv, const float* in, int B, int H, int W, const float* residual,
                      float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement aid (no counterpart in the reference): average device time of `iters` launches of this
 * layer on internally allocated buffers of the layer's dtype.  Allocates and synchronises. */
int yolo_conv_benchmark(YoloConv* conv, int B, int H, int W, int with_residual, int iters, float* avg_ms);

/* Test / measurement aid (no counterpart in the reference): force tile configuration `cfg` of the conv kernels for every later
 * conv launch of the process (ids: csrc/conv_igemm.hip launch_dtype; a layer the configuration does not apply to keeps the
 * default choice); cfg < 0 restores the default choice.  The initial value is the environment variable YOLO_CONV_CFG, read once. */
int yolo_debug_set_conv_cfg(int cfg);

/* ---- replaces upsample_layer (layer_utils.py:67-72) for the exact-2x case the network uses. */
int yolo_upsample2x(const float* in, int B, int H, int W, int C, float* out, void* stream);
/* ... and for any output size: tf.image.resize_nearest_neighbor(align_corners=True) of TF 1.x (layer_utils.py:71):
 * src = min(round(dst * (in - 1) / (out - 1)), in - 1) per axis.  in [B,H,W,C] f32 -> out [B,Ho,Wo,C] f32. */
int yolo_upsample_nearest(const float* in, int B, int H, int W, int C, int Ho, int Wo, float* out, void* stream);

/* ---- decode: replaces yolov3.reorg_layer + yolov3.predict (yolo_v3/model.py:69-153) and,
 * when scores != NULL, the caller's `pred_confs * pred_probs` (yolo_v3/test_single_image.py:38).
 * anchors: HOST float32 [9,2] (w,h) as parsed by parse_anchors (misc_utils.py:10-13).
 * boxes [B,n,4], confs [B,n,1], probs [B,n,C], scores [B,n,C]; n = 3*(g1^2+g2^2+g3^2). */
int yolo_decode(const float* fm1, const float* fm2, const float* fm3, int B, int H, int W,
                int class_num, const float* anchors, float* boxes, float* confs, float* probs,
                float* scores, void* stream);

/* reorg_layer alone (model.py:69-108) for ONE feature map [B,gh,gw,3*(5+C)] and its 3 anchors (HOST [3,2]):
 * boxes [B,gh,gw,3,4] = (center_x, center_y, w, h) in input pixels.  conf/prob logits are slices of fm. */
int yolo_reorg_boxes(const float* fm, int B, int gh, int gw, int H, int W, int class_num,
                     const float* anchors3, float* boxes, void* stream);

/* ---- per-class NMS: replaces gpu_nms (yolo_v3/utils/nms_utils.py:7-37, i.e. the score mask +
 * tf.image.non_max_suppression per class) with mode YOLO_NMS_TF, and cpu_nms/py_nms
 * (nms_utils.py:40-90) with mode YOLO_NMS_PY.
 * boxes [B,n,4], scores [B,n,C]: each of the B images is processed independently (the reference's
 * batch-flattening behaviour is obtained by calling with B=1, n=B*n).
 * Outputs per image i at rows [i*C*max_boxes, ...): class-ascending, selection order inside a class;
 * out_counts[i] = number of valid rows; out_index = index of the box inside its image (keep-index).
 * Limits (beyond the reference, which has none): max_boxes <= 1024 (the selected boxes of a class live in LDS) and
 * class_num <= 4096; yolo_detect_post / yolo_net_detect also need class_num <= 250.  The *_workspace_bytes functions return 0
 * for arguments outside these limits, the calls YOLO_ERR_UNSUPPORTED. */
size_t yolo_nms_workspace_bytes(int B, int n, int class_num, int max_boxes);
int yolo_nms(const float* boxes, const float* scores, int B, int n, int class_num, int max_boxes,
             float score_thresh, float iou_thresh, int mode, float* out_boxes, float* out_scores,
             int32_t* out_labels, int32_t* out_index, int32_t* out_counts, void* workspace,
             size_t workspace_bytes, void* stream);

/* yolo_nms for callers that cannot allocate device memory or pass a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, results and
 * error codes as yolo_nms without `workspace`: the candidate / selection arrays live in a YOLO_NMS_STATIC_BYTES
 * (16 MiB) device array inside the library, of which every GPU the library runs on holds its own copy until the
 * process ends.  Sizes whose yolo_nms_workspace_bytes exceeds YOLO_NMS_STATIC_BYTES (about B * n * class_num > 2 M,
 * e.g. more than one 608 x 608 image of 80 classes) are refused with YOLO_ERR_UNSUPPORTED.  Calls no hipMalloc /
 * hipFree.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two
 * threads) race; serialise them.  The FIRST call on a device resolves that device's copy of the array (hipGetSymbolAddress,
 * which loads the code object) under a lock: make one call per device - any valid small problem - before capturing a graph
 * or entering a loop that must not touch the runtime; later calls only enqueue kernels and a memset node. */
#define YOLO_NMS_STATIC_BYTES ((size_t)16 << 20)
int yolo_nms_static(const float* boxes, const float* scores, int B, int n, int class_num, int max_boxes,
                    float score_thresh, float iou_thresh, int mode, float* out_boxes, float* out_scores,
                    int32_t* out_labels, int32_t* out_index, int32_t* out_counts, void* stream);

/* ---- the post-processing chain of the driver in ONE call: replaces predict (yolo_v3/model.py:110-153), the
 * caller's pred_confs * pred_probs (yolo_v3/test_single_image.py:38) and gpu_nms (yolo_v3/utils/nms_utils.py:7-37),
 * applied to each of the B images independently.  Decode, score product and the score mask run as one pass over
 * the feature maps; the [B,n,C] probs / scores tensors are never materialised.  Outputs as yolo_nms (mode
 * YOLO_NMS_TF); results are bit-identical to yolo_decode (with scores) followed by yolo_nms. */
size_t yolo_detect_post_workspace_bytes(int B, int H, int W, int class_num, int max_boxes);
int yolo_detect_post(const float* fm1, const float* fm2, const float* fm3, int B, int H, int W, int class_num,
                     const float* anchors, int max_boxes, float score_thresh, float iou_thresh, float* out_boxes,
                     float* out_scores, int32_t* out_labels, int32_t* out_index, int32_t* out_counts,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The whole hot chain of yolo_v3/test_single_image.py:31-40 for a batch as ONE call: yolov3.forward -> predict ->
 * pred_confs * pred_probs -> gpu_nms, = yolo_net_forward followed by yolo_detect_post (same results, bit for bit);
 * the feature maps live in the workspace, and with lanes (see yolo_net_forward) each lane's post-processing runs
 * behind its own convs, overlapping the other lane.  anchors: HOST float32 [9][2]; outputs as yolo_detect_post. */
size_t yolo_net_detect_workspace_bytes(const YoloNet* net, int B, int H, int W, int max_boxes);
int yolo_net_detect(YoloNet* net, const float* images, int B, int H, int W, const float* anchors, int max_boxes,
                    float score_thresh, float iou_thresh, float* out_boxes, float* out_scores, int32_t* out_labels,
                    int32_t* out_index, int32_t* out_counts, void* workspace, size_t workspace_bytes, void* stream);

/* ---- detections of a batch -> one dense row list: replaces the box rescale the reference's drivers do
 * on the host after sess.run (yolo_v3/test_single_image.py:42-45, yolo_v3/video_test.py:67-71: x *= width_ori / new_w,
 * y *= height_ori / new_h in float32) and fixes the wire format a sharded run gathers.  boxes [B,cap,4], scores [B,cap],
 * labels [B,cap], counts [B]: the outputs of yolo_nms / yolo_detect_post / yolo_net_detect (cap = class_num * max_boxes).
 * scale_xy: DEVICE float32 [B,2] = (width_ori / new_w, height_ori / new_h) per image, or NULL (boxes stay in network pixels).
 * rows: DEVICE float32 [B*cap, 7], filled densely in image order with (image_base + image index, x0, y0, x1, y1, score,
 * label); offsets: DEVICE int32 [B+1], rows of image i are [offsets[i], offsets[i+1]); offsets[B] = number of rows.
 * boxes must be 16-byte aligned; image_base + B < 2^24 (the index travels as float32). */
int yolo_pack_detections(const float* boxes, const float* scores, const int32_t* labels, const int32_t* counts, int B,
                         int cap, const float* scale_xy, int image_base, float* rows, int32_t* offsets, void* stream);

/* ---- image preprocessing: replaces the host-side steps before the network,
 * cv2.resize(img, new_size) [INTER_LINEAR, 8-bit fixed point] -> cv2.cvtColor(BGR2RGB) -> float32(img) / 255.
 * (yolo_v3/test_single_image.py:25-28, yolo_v3/video_test.py:61-64); mode 1 ends in `/ 255.0 * 2.0 - 1.0`
 * (yolo_v2/Detector.py:37-39).  images: DEVICE uint8 [B, src_h, src_w, 3] in BGR, rows src_row_bytes apart
 * (0: packed), images src_row_bytes * src_h apart; out: DEVICE float32 [B, H, W, 3] RGB = the `inputs` of
 * yolo_net_forward / yolo2_net_forward.  OpenCV is an un-vendored dependency of the reference and not installed:
 * the filter restates its algorithm (see preprocess.hip), parity with cv2 itself is unpinned. */
int yolo_preprocess(const uint8_t* images, int B, int src_h, int src_w, int src_row_bytes, float* out, int H, int W,
                    int mode, void* stream);

/* ---- YOLOv2 (config 4).  yolo2_net_create/forward replace yolo_v2.build_networks + conv_layer + pooling_layer +
 * reorg (yolo_v2/yolo/yolo_v2.py:32-113).  vars: HOST float32, the TF variables concatenated in creation order per
 * conv (yolo_v2.py:75-83): weight [k,k,cin,cout], biases, then (batch-norm layers) scale, shift, rolling_mean,
 * rolling_variance.  images [B,H,W,3] f32 in [-1,1] (Detector.py:39) -> logits [B,H/32,W/32,box_per_cell*(5+C)].
 * Workspace size: yolo_net_workspace_bytes; destroy with yolo_net_destroy. */
int yolo2_net_create(int num_class, int box_per_cell, int dtype, const float* vars, size_t n_floats, YoloNet** out);
int yolo2_net_forward(YoloNet* net, const float* images, int B, int H, int W, void* workspace,
                      size_t workspace_bytes, float* logits, void* stream);
/* One YOLOv2 layer: replaces conv_layer (yolo_v2.py:73-89): SAME conv -> batch_normalization(mean, variance, shift,
 * scale, 1e-5) -> + biases -> max(0.1x, x); batch_norm == 0: conv + biases.  All parameter pointers HOST f32;
 * weight_hwio [k,k,cin,cout].  Run / size / free with yolo_conv_forward / yolo_conv_workspace_bytes / yolo_conv_destroy. */
int yolo2_conv_create(int dtype, int cin, int cout, int ksize, int batch_norm, const float* weight_hwio,
                      const float* biases, const float* scale, const float* shift, const float* mean,
                      const float* variance, YoloConv** out);
/* 2x2 stride-2 max-pool on NHWC f32 (pooling_layer, yolo_v2.py:91-94). */
int yolo_maxpool2x2(const float* in, int B, int H, int W, int C, float* out, void* stream);
/* Space-to-depth [B,H,W,C] -> [B,H/2,W/2,4C], channel block (dy*2+dx)*C (reorg, yolo_v2.py:97-113). */
int yolo_reorg2x(const float* in, int B, int H, int W, int C, float* out, void* stream);
/* yolo2_decode + yolo2_nms replace Detector.calc_output / get_boxes / calc_iou (yolo_v2/Detector.py:51-115).
 * anchors: HOST double[2*NB], widths then heights (yolo_v2/yolo/config.py:13).  boxes are float64 (cx,cy,w,h) in
 * input pixels, probs float32 = softmax(classes) * sigmoid(conf), exactly the reference's numpy dtypes. */
int yolo2_decode(const float* logits, int B, int S, int NB, int C, int image_size, const double* anchors,
                 double* boxes, float* probs, void* stream);
/* yolo2_nms: out_counts[i] <= max_det is the number of rows written for image i, always.  out_truncated (optional, [B]) is set
 * to 1 for an image with more survivors than rows (degenerate boxes; the reference has no cap: call again with max_det = n * C,
 * which always suffices), else 0. */
size_t yolo2_nms_workspace_bytes(int B, int n, int C);
int yolo2_nms(const double* boxes, const float* probs, int B, int n, int C, float threshold, double iou_thresh,
              int max_det, int32_t* out_counts, int32_t* out_cls, double* out_boxes, float* out_probs,
              int32_t* out_flat, int32_t* out_truncated, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H_ */
