/*
 * rtn.h — C-ABI of librtn.so: the MI355X (gfx950) RetinaNet detection hot path.
 *
 * The reference (jabhinav/RetinaNet-for-Table-Detection) has no FFI layer: the path is
 * Python calling Keras/TensorFlow.  Each entry point below replaces the TF/Keras/NumPy
 * work done at the cited reference lines; the Python host in
 * `retinanet-for-table-detection_amd/model/` keeps the reference's call surface and binds
 * these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary.
 *   - every entry returns int: 0 = RTN_OK, <0 = RTN_E*; rtn_last_error(h) gives the text.
 *   - the CALLER owns every buffer (device memory unless a parameter says "host");
 *     the library never allocates outputs.  Work is enqueued on the handle's stream and
 *     returns without synchronising.
 *   - activations are NHWC; "elements" are units of the tensor dtype.
 *   - all device pointers must be 16-byte aligned unless stated otherwise.
 */
#ifndef RTN_H
#define RTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTN_OK            0
#define RTN_EINVAL       -1   /* bad argument / shape the kernels do not support        */
#define RTN_EHIP         -2   /* a HIP runtime call failed (text has hipGetErrorString) */
#define RTN_ENOMEM       -3   /* caller-provided workspace too small                     */
#define RTN_EBOUNDS      -4   /* described access would leave a caller buffer           */

typedef struct rtn_ctx* rtn_handle_t;

typedef enum { RTN_BF16 = 0, RTN_F32 = 1, RTN_FP8 = 3 /* OCP e4m3fn bytes; rtn_conv2d_fp8_fwd / rtn_quantize_fp8 only */ } rtn_dtype_t;

#define RTN_MAX_GROUPS 5      /* pyramid levels P3..P7 in one grouped launch */
#define RTN_MAX_GT     64     /* ground-truth boxes per image (anchor targets) */
#define RTN_MAX_DET    300    /* reference max_detections (model/layers.py:277) */

/* ---- lifetime ------------------------------------------------------------------- */
int  rtn_create(rtn_handle_t* out, int device);
int  rtn_destroy(rtn_handle_t h);
/* why the last rtn_create of this process failed (which HIP call, its error string); "" after a success */
const char* rtn_create_error(void);
/* stream: a hipStream_t (as void*); NULL = the default stream. */
int  rtn_set_stream(rtn_handle_t h, void* stream);
const char* rtn_last_error(rtn_handle_t h);
const char* rtn_version(void);

/* ---- convolution (implicit GEMM on MFMA) ---------------------------------------------
 * Replaces every keras.layers.Conv2D / keras_resnet conv executed by TF:
 *   model/defineModel.py:101-117,155-163 (head stacks), :183-203 (FPN),
 *   :376-380 (keras_resnet backbone; frozen BN folded into w/bias by the host).
 *
 * out[b, oy, ox, n] = act( bias[n] + res(...) +
 *        sum_{kh,kw,c} in[b, oy*sy - pad_t + kh, ox*sx - pad_l + kw, c] * w[n, (kh*KW+kw)*Crun + c] )
 * Out-of-image taps read zero (TF zero padding; pad_t/pad_l carry TF 'same' asymmetry,
 * i.e. pad_before = floor(pad_total/2), SURVEY §8a notes).
 *
 * A launch covers up to RTN_MAX_GROUPS "groups" that share w/bias (the pyramid levels of a
 * head layer, model/defineModel.py:217); ordinary layers use one group.
 */
#define RTN_CONV_RELU         0x01  /* max(x,0) after bias+residual                        */
#define RTN_CONV_SIGMOID      0x02  /* 1/(1+exp(-x))  (model/defineModel.py:123)           */
#define RTN_CONV_RES_SAME     0x04  /* += res[b,oy,ox,n]  (keras Add; ResNet shortcut)     */
#define RTN_CONV_RES_UPSAMPLE 0x08  /* += res[b, floor(oy*rs_h), floor(ox*rs_w), n]:
                                       UpsampleLike + Add, model/layers.py:89-98,
                                       model/defineModel.py:184,189-190,195                */
#define RTN_CONV_OUT_F32      0x10  /* store f32 even when dtype is bf16                   */
#define RTN_CONV_RELU_MASK    0x20  /* backward of a ReLU: result = mask[b,oy,ox,n] > 0 ? result : 0, applied after
                                       the residual add (the residual carries the other gradient contributions) */
#define RTN_CONV_MASK_PRE     0x40  /* with RELU_MASK: mask the convolution result BEFORE the residual add (C6_relu:
                                       the residual holds gradients of the un-rectified tensor)            */

typedef struct {
    const void* in;          /* NHWC activations                                       */
    void*       out;
    const void* res;         /* residual source or NULL                                */
    int64_t in_elems;        /* size of the `in` buffer (bounds validation)            */
    int64_t out_elems;       /* size of the `out` buffer                               */
    int64_t res_elems;
    int64_t in_img_stride;   /* elements between images of `in`                        */
    int64_t out_img_stride;  /* elements between images of `out`                       */
    int64_t out_off;         /* element offset of this group inside an image of `out`
                                (level offset of the concatenated head output,
                                model/defineModel.py:217)                              */
    int64_t res_img_stride;
    int32_t in_row_stride;   /* elements between rows of `in`                          */
    int32_t Hin, Win;        /* taps with iy>=Hin or ix>=Win (or <0) read zero         */
    int32_t Hout, Wout;
    int32_t Hres, Wres;      /* residual map extent (RES_UPSAMPLE)                     */
    int32_t res_ld;          /* elements per pixel of `res`                            */
    const void* mask;        /* RELU_MASK: the forward activation whose ReLU is being
                                differentiated; indexed like `out` (mask_ld per pixel)  */
    int64_t mask_elems;
    int64_t mask_img_stride;
    int32_t mask_ld;
    int32_t out_step;        /* 0/1 = dense; s > 1 scatters output pixel (oy,ox) to pixel
                                (oy*s, ox*s) of an image out_pix_w pixels wide: the
                                data gradient of a stride-s 1x1 'valid' convolution
                                (caller zero-fills `out`); res/mask use the same pixel  */
    int32_t out_pix_w;       /* width in pixels of the `out` image when out_step > 1   */
    int32_t reserved_;
} rtn_conv_group_t;

typedef struct {
    rtn_conv_group_t g[RTN_MAX_GROUPS];
    int32_t ngroups;
    int32_t batch;
    int32_t dtype;           /* rtn_dtype_t of in / w / res (and out unless OUT_F32)   */
    const void*  w;          /* [w_rows][Ktot], K contiguous; rows >= N are zero       */
    const float* bias;       /* [w_rows] f32 or NULL                                   */
    int32_t w_rows;          /* N rounded up to a multiple of 128                      */
    int32_t N;               /* valid output channels                                  */
    int32_t KH, KW;
    int32_t Crun;            /* contiguous input elements per tap: Cin for ordinary
                                layers (power of two, Crun*sizeof >= 128 B); 32 for the
                                packed stem (rtn_stem_pack)                            */
    int32_t pix_stride;      /* elements between horizontally adjacent taps (= Cin;
                                4 for the packed stem)                                 */
    int32_t sy, sx;          /* stride                                                 */
    int32_t pad_t, pad_l;
    int32_t out_ld;          /* elements per output pixel (= N for NHWC; 4*A / K*A for
                                the head outputs)                                      */
    int32_t flags;
    int32_t reserved_;
    void*   workspace;       /* caller-owned scratch for the launches that split a K loop over workgroups (f32 partial-sum
                                slabs of the split-K, tail-split, K-slice and stream-K paths), >= rtn_conv2d_workspace_bytes(h, d)
                                bytes, 16-byte aligned: launches that may overlap in time (other streams) need different buffers.
                                Its first RTN_CONV_SYNC_BYTES are the SYNC BLOCK of the in-launch reductions (stream-K: one flag
                                per workgroup): they must be zero when a launch starts - rtn_conv_workspace_init() once after
                                allocation does that - and every launch that completes leaves them zero again; everything behind
                                them is scratch used only while a launch runs.  NULL / too small = no K split for this launch
                                (slower on the layers that want one, same results up to f32 summation order).  */
    int64_t workspace_bytes;
} rtn_conv_desc_t;
#define RTN_CONV_SYNC_BYTES 4096

/* Bytes of `workspace` the launch described by `d` can use on this device (0 for most layers).  The library never allocates:
 * every forward / dgrad entry point below takes its scratch from the descriptor. */
size_t rtn_conv2d_workspace_bytes(rtn_handle_t h, const rtn_conv_desc_t* d);
/* Zero the sync block of a freshly allocated convolution workspace.  Once per buffer; the block is zero on return (the call
 * synchronises the handle's stream), so the buffer may then serve launches on any stream. */
int rtn_conv_workspace_init(rtn_handle_t h, void* workspace, size_t workspace_bytes);
/* Workgroups of the last convolution launch on this handle if it ran in STREAM-K form (csrc/rtn_conv_gemm8.hip, rtn_conv_halo8.hip:
 * the launch's tiles x K steps shared evenly by the workgroups, partial tiles handed over and added in a fixed order inside the
 * launch), 0 otherwise.  rtn_debug_conv_sync_timeouts: synchronises the stream and reads the sync block's error word - the number of
 * bounded flag polls that gave up (always 0 unless a workgroup of a launch never ran). */
int rtn_debug_last_conv_streamk(rtn_handle_t h);
/* (tile rows << 16) | tile columns of the last convolution launch on this handle (the persistent kernels: 128 / 192 / 256 staged rows
 * x 128 / 256 columns; generations 1-3: 128 or 256 x 64 / 128 / 256).  For the launch-plan table of DESIGN.md (tools/launch_plan.py). */
int rtn_debug_last_conv_tile(rtn_handle_t h);
int rtn_debug_conv_sync_timeouts(rtn_handle_t h, const void* workspace, unsigned* count);
/* Which kernel generation the last convolution launch on this handle ran (1: 128-row register-staged, 2: 256-row LDS-DMA per tap,
 * 3: 256-row shared halo, 4: persistent 8-phase halo kernel, 5: persistent 1x1 kernel, 6: narrow-N head-output kernel).  For tests and profiles: proves which native path executed. */
int rtn_debug_last_conv_impl(rtn_handle_t h);
/* Which weight-gradient kernel the last wgrad call of this handle ran: 2 the 256 x 256
 * LDS-DMA kernel, 3 the 128 x 128 LDS-DMA kernel, 4 the nine-tap window kernel (csrc/rtn_wgrad_win.hip: stride-1 3x3 'same' layers
 * with 64 or a multiple of 128 filters), 0 the register-staged kernel (fp32 and odd shapes). */
int rtn_debug_last_wgrad_impl(rtn_handle_t h);
int rtn_conv2d_fwd(rtn_handle_t h, const rtn_conv_desc_t* d);

/* Two 1x1 convolutions that are ADDED, as one GEMM over the concatenated K: out = W1 . in + W2 . in2[stepped] + bias (+ the
 * epilogue of `d`).  This is the first bottleneck block of a ResNet stage (keras_resnet: res*a_branch2c + res*a_branch1,
 * joined by keras Add; model/defineModel.py:357-389 builds the backbone): the projection shortcut never becomes a tensor.
 * `d`: one group, KH = KW = 1, stride 1, no padding, Crun = channels of `in`; w = [w_rows][Crun + C] with the second
 * layer's filters behind the first's along K, bias = the sum of the two biases.  `s2`: the block input, sampled at pixel
 * (oy*step, ox*step) (step 2 = the stride-2 'valid' 1x1 shortcut of stages 3-5). */
typedef struct {
    const void* in;          /* NHWC, same dtype as d->dtype                          */
    int64_t in_elems;
    int64_t in_img_stride;   /* elements                                               */
    int32_t in_row_stride;   /* elements                                               */
    int32_t pix_stride;      /* elements between horizontally adjacent pixels          */
    int32_t Hin, Win;
    int32_t C;               /* channels read per pixel (C * sizeof multiple of 128 B) */
    int32_t step;            /* 1 or the shortcut's stride                             */
} rtn_conv_src2_t;
int rtn_conv1x1_dual_fwd(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_src2_t* s2);
/* rtn_conv2d_workspace_bytes for that launch (its stream-K form; the same sync-block contract). */
size_t rtn_conv1x1_dual_workspace_bytes(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_src2_t* s2);

/* fp8 (OCP e4m3fn) convolution for BASELINE.json configs[4] ("fp8 MFMA convs", the head towers default_classification_model /
 * default_regression_model, model/defineModel.py:78-167): v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, f32 accumulate.
 *   d->dtype = RTN_FP8: `in` and `w` are e4m3 bytes, i.e. quantised x * s_x and w * s_w with per-tensor scales the caller chose;
 *   d: stride-1 'same' KHxKW layer (KW 2..4) over dense NHWC inputs, Crun a multiple of 128, N and out_ld multiples of 8,
 *      flags = 0 or RTN_CONV_RELU, bias f32 (unscaled) or NULL; grouped levels as for rtn_conv2d_fwd.
 *   y = acc * acc_scale + bias  (acc_scale = 1 / (s_x * s_w));  ReLU if flagged;
 *   out_dtype RTN_BF16: out = bf16(y);   RTN_FP8: out = e4m3(clamp(y * out_scale, -448, 448)), round to nearest even.
 * rtn_quantize_fp8: dst[i] = e4m3(clamp(src[i] * scale, -448, 448)) for a bf16 / f32 tensor of n elements (n % 8 == 0). */
typedef struct {
    float acc_scale;
    float out_scale;
    int32_t out_dtype;
} rtn_conv_fp8_t;
int rtn_conv2d_fp8_fwd(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_fp8_t* q);
/* A bf16 layer of rtn_conv2d_fwd (any kernel generation, residual / ReLU epilogues included; not OUT_F32 / SIGMOID) whose
 * output tensor is written as e4m3(clamp(y * out_scale, -448, 448)) instead of bf16: the producer of an fp8 layer's input
 * (keras_resnet's branch2a in front of the 3x3 branch2b), so that no separate quantise pass touches the tensor.
 * `out`, out_elems, out_img_stride, out_off, out_ld count e4m3 bytes. */
int rtn_conv2d_fwd_fp8out(rtn_handle_t h, const rtn_conv_desc_t* d, float out_scale);
int rtn_quantize_fp8(rtn_handle_t h, const void* src, int src_dtype, void* dst, int64_t n, float scale);

/* One keras_resnet identity bottleneck block of the 64-channel stage (res2b, res2c) as ONE launch (inference, bf16):
 *   h1    = relu(conv3x3(a_in;  w2b) + b2b)                 branch2b + BN + ReLU   (3x3 'same', 64 -> 64)
 *   x_out = relu(conv1x1(h1;    w2c) + b2c + x_in)          branch2c + BN + Add + ReLU  (64 -> 256)
 *   a_out = relu(conv1x1(x_out; w2a) + b2a)                 the NEXT block's branch2a + BN + ReLU (256 -> 64); skipped if a_out is NULL
 * (the projection-shortcut form of the stage's first block: see p_in / wproj below)
 * (keras_resnet bottleneck_2d as instantiated by model/defineModel.py:376-380; BN folded into w / b as for rtn_conv2d_fwd.)
 * h1 never reaches memory and x_out is read back by nobody: 0.68 GB of HBM traffic per block at batch 8 instead of 1.09 GB.
 * Tensors are dense NHWC bf16, weights [N][KH*KW*Cin] K-contiguous as for rtn_conv2d_fwd (no row padding needed), biases f32.
 * Intermediate roundings are those of the three separate launches (h1 and x_out are rounded to bf16 before they are multiplied
 * again); only the f32 summation order differs. */
typedef struct {
    const void* a_in;  int64_t a_in_elems;    /* [batch][H][W][mid]                      */
    const void* x_in;  int64_t x_in_elems;    /* [batch][H][W][4*mid]  shortcut           */
    void*       x_out; int64_t x_out_elems;   /* [batch][H][W][4*mid]                     */
    void*       a_out; int64_t a_out_elems;   /* [batch][H][W][mid] or NULL               */
    const void* w2b;   const float* b2b;      /* [mid][3*3*mid], [mid]                    */
    const void* w2c;   const float* b2c;      /* [4*mid][mid],   [4*mid]                  */
    const void* w2a;   const float* b2a;      /* [mid][4*mid],   [mid]   (with a_out)     */
    int32_t batch, H, W;
    int32_t mid;                               /* bottleneck channels: 64                  */
    int32_t dtype;                             /* RTN_BF16                                 */
    int32_t w2c_ld;                            /* elements between rows of w2c / wproj; 0 = mid (dense)                */
    /* The stage's FIRST block (res2a), whose shortcut is a projection (model/defineModel.py:376-380, keras_resnet
     * bottleneck_2d with block == 0):  x_out = relu(conv1x1(h1; w2c) + conv1x1(p_in; wproj) + b2c)  with b2c = b2c + b1 summed by
     * the caller and p_in the block input [batch][H][W][mid].  x_in is not read.  With a_out the next block's branch2a rides along
     * as in the identity form (its filters and the two of this block are then streamed through the LDS chunk by chunk).  With the K-concatenated
     * filters of rtn_conv1x1_dual_fwd ([4*mid][mid + mid]): w2c = that matrix, wproj = w2c + mid elements, w2c_ld = 2 * mid. */
    const void* p_in;  int64_t p_in_elems;
    const void* wproj;
    /* optional: branch2b's output h1 [batch][H][W][mid] is also written (the training forward keeps it for the backward pass) */
    void*       h1_out; int64_t h1_out_elems;
} rtn_bottleneck_desc_t;
int rtn_bottleneck64_fwd(rtn_handle_t h, const rtn_bottleneck_desc_t* d);

/* The seam between two keras_resnet identity bottleneck blocks of the 128-channel stage (res3b..d) as ONE
 * launch (inference and training forward, bf16):
 *   x_out = relu(conv1x1(h_in;  w2c) + b2c + x_in)          this block's branch2c + BN + Add + ReLU   (mid -> out = 4 mid)
 *   a_out = relu(conv1x1(x_out; w2a) + b2a)                 the NEXT block's branch2a + BN + ReLU     (out -> next = mid)
 * (keras_resnet bottleneck_2d as instantiated by model/defineModel.py:376-380; replaces one rtn_conv2d_fwd with
 * RTN_CONV_RES_SAME | RTN_CONV_RELU and the rtn_conv2d_fwd + RTN_CONV_RELU behind it.)  Both tensors are written; x_out is not read
 * back.  Tensors are dense [pixels][channels] bf16 (NHWC with batch x H x W flattened: the two convs are pointwise, stride 1),
 * weights [N][K] K-contiguous, biases f32.  Roundings are those of the two separate launches (x_out is rounded to bf16 before it is
 * multiplied again) and so is the f32 summation order: the results are bit-identical to them.
 * Built for (mid, out, next) = (128, 512, 128): rtn_chain1x1_supported. */
typedef struct {
    const void* h_in;  int64_t h_in_elems;    /* [pixels][mid]                            */
    const void* x_in;  int64_t x_in_elems;    /* [pixels][out]   shortcut                 */
    void*       x_out; int64_t x_out_elems;   /* [pixels][out]                            */
    void*       a_out; int64_t a_out_elems;   /* [pixels][next]                           */
    const void* w2c;   const float* b2c;      /* [out][mid],  [out]                       */
    const void* w2a;   const float* b2a;      /* [next][out], [next]                      */
    int64_t pixels;
    int32_t mid, out, next;
    int32_t dtype;                             /* RTN_BF16                                 */
} rtn_chain_desc_t;
int rtn_chain1x1_fwd(rtn_handle_t h, const rtn_chain_desc_t* d);
int rtn_chain1x1_supported(int mid, int out, int next);

/* Data gradient (what TF autodiff emits as Conv2DBackpropInput under fit_generator, RetinaNet.py:280).  The same
 * implicit GEMM run on dY: `in` = dY, `w` = the forward weights re-packed by rtn_pack_dgrad_weights
 * (w_d[c][(KH-1-kh, KW-1-kw, n)] = w[n][(kh,kw,c)]), pad = K-1-pad_fwd, stride 1 (a stride-2 1x1 forward conv uses
 * out_step = 2; a stride-2 3x3 forward conv is differentiated on the zero-inserted dY built by rtn_zero_insert2).
 * RES_SAME accumulates into an existing gradient, RELU_MASK applies the ReLU of the tensor being differentiated. */
int rtn_conv2d_dgrad(rtn_handle_t h, const rtn_conv_desc_t* d);
int rtn_pack_dgrad_weights(rtn_handle_t h, const void* w_fwd, void* w_dgrad, int dtype, int N, int w_rows_fwd,
                           int KH, int KW, int Cin, int Cout_run, int w_rows_dgrad);
/* Every layer's dgrad weights in ONE launch (after an optimizer step rewrote the forward weights).  table_dev: device array of
 * nlayers x 10 int64 {w_fwd pointer, w_dgrad pointer, N, KH, KW, Cin, Cout_run, w_rows_dgrad, first flat element of the layer in
 * the concatenation of all w_dgrad arrays, 0}; total_elems = sum of w_rows_dgrad * KH*KW*Cout_run. */
int rtn_pack_dgrad_weights_multi(rtn_handle_t h, const int64_t* table_dev, int nlayers, int64_t total_elems, int dtype);

/* Weight gradient (Conv2DBackpropFilter).  `d` describes the FORWARD convolution with two re-interpretations:
 * g[i].out / out_elems / out_img_stride / out_off and out_ld describe dY (same dtype as `in`; N and out_ld must span whole
 * 16-byte chunks — the head outputs' 36- and 9-channel gradients are padded to 64 channels with rtn_pad_cast_rows), and
 * w / bias are ignored.  dW is f32 [w_rows][KH*KW*Crun] in the forward weight layout and is ACCUMULATED into (zero it
 * once per step; the five pyramid levels and any number of micro-batches add into the same buffer).  */
size_t rtn_conv2d_wgrad_workspace_bytes(const rtn_conv_desc_t* d);
int rtn_conv2d_wgrad(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, void* workspace, size_t workspace_bytes);
/* the same with BiasAddGrad fused: db[0..db_n) += sum over pixels of dY[:, n] (db_n <= N) */
int rtn_conv2d_wgrad_bias(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, void* workspace,
                          size_t workspace_bytes);
/* The row-info table of a weight-gradient launch depends only on the descriptor (geometry, strides, dY / X offsets): build it
 * once per layer with rtn_conv2d_wgrad_rowinfo and reuse it with rtn_conv2d_wgrad_prepared (db may be NULL). */
int rtn_conv2d_wgrad_rowinfo(rtn_handle_t h, const rtn_conv_desc_t* d, void* workspace, size_t workspace_bytes);
int rtn_conv2d_wgrad_prepared(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, const void* workspace,
                              size_t workspace_bytes);
/* db[n] += sum over rows of dy[row][n]   (rows x N matrix with leading dimension ld, dtype bf16/f32) */
int rtn_bias_grad(rtn_handle_t h, const void* dy, int dtype, int64_t rows, int N, int64_t ld, float* db);
/* out[r][0..cout) = cast(in[r][0..cin)), zero beyond cin */
int rtn_pad_cast_rows(rtn_handle_t h, const float* in, void* out, int dtype, int64_t rows, int cin, int cout);
/* out[b][2y][2x] = in[b][y][x], zero elsewhere; out is [B][Hu][Wu][C] with Hu >= 2H-1, Wu >= 2W-1 */
int rtn_zero_insert2(rtn_handle_t h, const void* in, void* out, int dtype, int B, int H, int W, int C, int Hu, int Wu);
/* adjoint of UpsampleLike+Add (model/layers.py:89-98): d_src[sy][sx] (+)= sum of d_dst over the pixels that read it */
int rtn_upsample_add_bwd(rtn_handle_t h, const void* d_dst, void* d_src, int dtype, int B, int Hd, int Wd, int Hs, int Ws,
                         int C, int accumulate);
/* keras_resnet pool1 backward; scratch_f32 holds B*Hin*Win*C floats */
int rtn_maxpool3x3s2_tfsame_bwd(rtn_handle_t h, const void* x, const void* dy, void* dx, int dtype, int B, int Hin, int Win,
                                int C, float* scratch_f32, int relu_mask /* also zero dx where x <= 0 (x is a ReLU output) */);

/* Training-mode pool1: the forward also records the winning tap (kh*3+kw, first maximum in scan order) per output element
 * (idx: B*Hout*Wout*C bytes) and the backward is an atomic-free gather over the <= 4 windows that contain an input pixel.
 * relu_mask of the backward: 0 none; 1: x = the pool's INPUT [B][Hin][Win][C] (a ReLU output), dx zeroed where x <= 0; 2: x = the
 * pool's OUTPUT [B][Hout][Wout][C]: a window passes its gradient only when its maximum is > 0 - the same mask, for a forward that
 * never wrote the pool's input (rtn_stem_conv_pool_branch2a with pool_idx). */
int rtn_maxpool3x3s2_tfsame_fwd_idx(rtn_handle_t h, const void* in, void* out, uint8_t* idx, int dtype, int B, int Hin, int Win, int C);
int rtn_maxpool3x3s2_tfsame_bwd_idx(rtn_handle_t h, const void* dy, const uint8_t* idx, const void* x, void* dx, int dtype,
                                    int B, int Hin, int Win, int C, int relu_mask);

/* ---- optimizer: Adam(lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=0.001)  (RetinaNet.py:130) ---------------
 * Parameters live in ONE flat f32 buffer (forward weight layout per layer).  gscale[i] multiplies the raw gradient
 * (frozen-BN fold factor of the layer's output channel; 0 for structurally-zero slots), fold[i] re-creates the forward
 * weight w_fwd[i] = cast(w[i] * fold[i]).  Clipping is by the GLOBAL norm of all gradients (standalone Keras 2.x,
 * SURVEY §8a a20): sumsq = sum((g*gscale)^2) from rtn_sumsq (all-reduced by the caller under data parallelism),
 * factor = min(1, clipnorm / (sqrt(sumsq) * |grad_mul|)); grad_mul rescales g (e.g. 1/world_size).  clipnorm <= 0: off. */
size_t rtn_sumsq_workspace_bytes(void);
int rtn_sumsq(rtn_handle_t h, const float* g, const float* scale, int64_t n, double* out, void* workspace, size_t workspace_bytes);
int rtn_adam_clipnorm_step(rtn_handle_t h, float* w, float* m, float* v, const float* g, const float* gscale, const float* fold,
                           void* w_fwd, int fwd_dtype, int64_t n, int64_t step, float lr, float beta1, float beta2, float eps,
                           const double* sumsq, float clipnorm, float grad_mul);
/* The same step with PER-TENSOR clipping (tf.keras / Keras >= 2.4 semantics of Adam(clipnorm=c), RetinaNet.py:130 under those
 * versions): tensor t, the segment [seg_begin[t], seg_begin[t+1]) of the flat vector (int64 table on the device,
 * nseg + 1 entries, at most 2048 tensors), is scaled by clipnorm / max(norm_t, clipnorm).  rtn_sumsq_segments writes the nseg sums of
 * (g*scale)^2 in a fixed order (one workgroup per tensor); `elem_offset` is the flat index of w[0] when the step runs on a sub-range. */
int rtn_sumsq_segments(rtn_handle_t h, const float* g, const float* scale, const int64_t* seg_begin_dev, int nseg, double* out_dev);
int rtn_adam_clipnorm_step_segments(rtn_handle_t h, float* w, float* m, float* v, const float* g, const float* gscale, const float* fold,
                                    void* w_fwd, int fwd_dtype, int64_t n, int64_t step, float lr, float beta1, float beta2, float eps,
                                    const int64_t* seg_begin_dev, int nseg, const double* sumsq_seg_dev, int64_t elem_offset,
                                    float clipnorm, float grad_mul);

/* ---- stem input packing ------------------------------------------------------------
 * NHWC C=3 image batch -> zero-padded [B][Hp][Wp][4] so that the 7x7/2 stem conv
 * (keras_resnet conv1 + ZeroPadding2D(3)) becomes an implicit GEMM with one
 * contiguous 32-element run per kernel row.  Hp = H+6 rounded so 2*(Hout-1)+8 <= Hp,
 * Wp likewise and even.  src_dtype: RTN_F32, RTN_BF16, or 2 = uint8 (then the reference
 * normalisation x/127.5-1 is fused: model/utils.py:43-46, model/Parameters.py:23-24).
 */
int rtn_stem_pack(rtn_handle_t h, const void* src, int src_dtype, void* dst, int dst_dtype,
                  int B, int H, int W, int Hp, int Wp);

/* ---- conv1 + ReLU + pool1 in one kernel (inference, bf16 path) -------------------------
 * ZeroPadding2D(3) + conv1 7x7/2 (frozen BN folded into w / bias) + ReLU + MaxPool 3x3/2 'same' of keras_resnet
 * (model/defineModel.py:357-389 builds it) on the packed image rtn_stem_pack writes ([B][Hp][Wp][4] bf16).
 * w_packed: the packed stem filters [w_rows >= 64][8 kernel rows][32] bf16 (kernel row kh = 7 pixels x 4 channels + 4 zero
 * elements, row 7 all zero) as rtn_conv2d_fwd takes them; out: [B][H2][W2][64] bf16, H1 = (H-1)/2+1, H2 = (H1+1)/2 (W alike).
 * Bit-identical to rtn_conv2d_fwd (RELU) -> rtn_maxpool3x3s2_tfsame_fwd on the same packed image: one MFMA per kernel row in
 * the same order, one bf16 rounding of the ReLU output; the 34 MB/image conv1 tensor never reaches HBM. */
int rtn_stem_conv_pool(rtn_handle_t h, const void* packed, int Hp, int Wp, const void* w_packed, int w_rows,
                       const float* bias, void* out, int B, int H, int W);
/* The same kernel with the first bottleneck's branch2a appended (keras_resnet res2a_branch2a 1x1 64 -> 64 + bn2a_branch2a + ReLU,
 * model/defineModel.py:376-380): a_out [B][H2][W2][64] bf16 = relu(w2a . pool1 + b2a), computed from the pooled pixels while they
 * are in registers.  w2a: [>= 64][64] bf16 K-contiguous (BN folded), b2a: f32 [64].  `out` (pool1) is still written: the block's
 * projection shortcut reads it.  pool_idx (optional, u8 [B][H2][W2][64]): the winning tap kh * 3 + kw of every pooled element, as
 * rtn_maxpool3x3s2_tfsame_fwd_idx records it - the training forward then needs neither conv1's output nor a pooling launch
 * (rtn_maxpool3x3s2_tfsame_bwd_idx with relu_mask = 2 takes the ReLU mask from pool1 itself).  a_out may be NULL when pool_idx is given. */
int rtn_stem_conv_pool_branch2a(rtn_handle_t h, const void* packed, int Hp, int Wp, const void* w_packed, int w_rows,
                                const float* bias, void* out, int B, int H, int W, const void* w2a, const float* b2a, void* a_out,
                                uint8_t* pool_idx);

/* ---- MaxPool 3x3 / 2, TF 'same' (keras_resnet pool1; -inf padding) ----------------- */
int rtn_maxpool3x3s2_tfsame_fwd(rtn_handle_t h, const void* in, void* out, int dtype,
                                int B, int Hin, int Win, int C);

/* ---- elementwise ReLU (C6_relu, model/defineModel.py:202) --------------------------- */
int rtn_relu(rtn_handle_t h, const void* in, void* out, int dtype, int64_t n);

/* ---- anchors ------------------------------------------------------------------------
 * Host helper: the 9 base anchors of one level, f64 (x1,y1,x2,y2), bit-identical to
 * generate_anchors (model/anchors.py:243-278).  ratios/scales are the float32-valued
 * parameters promoted to f64 (model/anchors.py:31-32).
 */
int rtn_generate_anchors(double base_size, const double* ratios, int nratios,
                         const double* scales, int nscales, double* out /* [nr*ns][4] */);

typedef struct {
    int32_t nlevels;
    int32_t A;                               /* anchors per cell                          */
    int32_t H[RTN_MAX_GROUPS], W[RTN_MAX_GROUPS];   /* guess_shapes, model/anchors.py:155 */
    int32_t stride[RTN_MAX_GROUPS];
    int32_t anchor_off[RTN_MAX_GROUPS + 1];  /* first anchor index of each level          */
    double  base[RTN_MAX_GROUPS][16][4];     /* base anchors per level (f64)              */
} rtn_anchor_cfg_t;

/* Materialise anchors_for_shape (model/anchors.py:169-204): out is f64 [N][4] (device). */
int rtn_anchors_f64(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, double* out);
/* The in-graph float32 Anchors layer (model/layers.py:42-53, model/utils.py:51-80). */
int rtn_anchors_f32(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, float* out);

/* ---- anchor -> ground-truth assignment -----------------------------------------------
 * Replaces anchor_targets_bbox + compute_gt_annotations + compute_overlap +
 * bbox_transform (model/anchors.py:36-117,282-313; model/utils.py:180-211), f64 math,
 * f32 IoU compare, first-max argmax: bit-exact with the NumPy reference.
 *   gt_boxes  : device f64 [B][RTN_MAX_GT][4]   (x1,y1,x2,y2)
 *   gt_labels : device i32 [B][RTN_MAX_GT]
 *   gt_count  : device i32 [B]
 *   img_hw    : device i32 [B][2]  each image's own (unpadded) height,width
 *               (model/anchors.py:85-90)
 *   regression_batch : device f32 [B][N][5];  labels_batch : device f32 [B][N][K+1]
 */
int rtn_anchor_targets(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, int B, int num_classes,
                       const double* gt_boxes, const int32_t* gt_labels,
                       const int32_t* gt_count, const int32_t* img_hw,
                       double negative_overlap, double positive_overlap,
                       float* regression_batch, float* labels_batch);

/* Stand-alone pieces of the same path, for callers that use the reference's individual functions:
 *   rtn_anchor_targets_explicit : anchor_targets_bbox on a caller-supplied anchors array (device f64 [N][4])
 *   rtn_compute_overlap         : utils.compute_overlap (model/utils.py:180-211) -> f32 [N][G]
 *   rtn_gt_annotations          : compute_gt_annotations (model/anchors.py:96-117) on that matrix
 *   rtn_bbox_transform          : bbox_transform (model/anchors.py:282-313), f64 */
int rtn_anchor_targets_explicit(rtn_handle_t h, const double* anchors, int N, int B, int num_classes, const double* gt_boxes,
                                const int32_t* gt_labels, const int32_t* gt_count, const int32_t* img_hw,
                                double negative_overlap, double positive_overlap, float* regression_batch, float* labels_batch);
int rtn_compute_overlap(rtn_handle_t h, const double* boxes, const double* gts, int N, int G, float* out);
int rtn_gt_annotations(rtn_handle_t h, const float* overlaps, int N, int G, double negative_overlap, double positive_overlap,
                       uint8_t* positive, uint8_t* ignore, int64_t* argmax);
int rtn_bbox_transform(rtn_handle_t h, const double* anchors, const double* gt_boxes, int N, const double* mean4 /* host */,
                       const double* std4 /* host */, double* out);

/* ---- focal + smooth-L1 (model/losses.py:5-46, 49-91) --------------------------------
 * fwd: sums[0] = sum of focal terms over non-ignored anchors, sums[1] = sum of smooth-L1
 * terms over positive anchors, sums[2] = number of positive anchors (state == 1); the
 * caller divides by max(1, count) (all-reduced across ranks under data parallelism so the
 * normaliser is the merged-batch count).  `sums` is device f64[4].
 * bwd: d_cls = dLoss/dclassification * inv_norm_cls (w.r.t. the PROBABILITY, or w.r.t.
 * the pre-sigmoid logit when wrt_logits != 0), d_reg = dLoss/dregression * inv_norm_reg.
 */
int rtn_retina_loss_fwd(rtn_handle_t h, int64_t rows /* B*N */, int num_classes,
                        const float* labels_batch, const float* regression_batch,
                        const float* classification, const float* regression,
                        float alpha, float gamma, float sigma, double* sums,
                        void* workspace, size_t workspace_bytes);
size_t rtn_retina_loss_workspace_bytes(int64_t rows);
int rtn_retina_loss_bwd(rtn_handle_t h, int64_t rows, int num_classes,
                        const float* labels_batch, const float* regression_batch,
                        const float* classification, const float* regression,
                        float alpha, float gamma, float sigma,
                        float inv_norm_cls, float inv_norm_reg, int wrt_logits,
                        float* d_cls, float* d_reg);

/* Same as rtn_retina_loss_bwd with the normalisers read on the device: inv_norm = 1 / max(1, sums[2]) for the focal
 * term and 1 / max(1, sums[3]) for smooth-L1, `sums` being rtn_retina_loss_fwd's output (all-reduced over ranks first
 * under data parallelism) — no host round trip between loss forward and backward. */
int rtn_retina_loss_bwd_dev(rtn_handle_t h, int64_t rows, int num_classes,
                            const float* labels_batch, const float* regression_batch,
                            const float* classification, const float* regression,
                            float alpha, float gamma, float sigma, const double* sums, int wrt_logits,
                            float* d_cls, float* d_reg);

/* ---- decode + clip + score threshold + NMS + top-k + pad ---------------------------
 * Replaces Anchors/RegressBoxes/ClipBoxes/FilterDetections for num_classes = K with
 * class_specific_filter=True, nms=True (model/layers.py:42-53,136-138,157-171,177-264;
 * model/utils.py:84-112).  Anchors are recomputed from the index in float32 exactly as
 * the in-graph Anchors layer does; boxes are clipped to [0,W]x[0,H] of the padded canvas.
 *   regression     : device f32 [B][N][4]
 *   classification : device f32 [B][N][K]
 *   boxes [B][max_det][4] f32, scores [B][max_det] f32, labels [B][max_det] i32 (pad -1)
 */
size_t rtn_detect_workspace_bytes(int B, int64_t N, int num_classes);
int rtn_decode_filter_nms(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, int B, int num_classes,
                          const float* regression, const float* classification,
                          int canvas_h, int canvas_w,
                          float score_threshold, float nms_threshold, int max_detections,
                          float* boxes, float* scores, int32_t* labels,
                          void* workspace, size_t workspace_bytes);

/* ---- page preprocessing -------------------------------------------------------------------------------
 * rtn_preprocess_dt3: DetectTablesUtils.preProcessSampleImages (DetectTablesUtils.py:251-261) for B equally sized pages:
 *   src uint8 [B][H][W][channels] (3 = BGR as cv2.imread gives, 1 = gray) -> dst uint8 [B][H][W][3] in OpenCV channel order
 *   b = DIST_L2 (mask 5), g = DIST_L1, r = DIST_C of the Gaussian adaptive threshold (block 11, C 2), saturated like imwrite.
 *   binary_out (optional, [B][H][W]) receives the thresholded image.  W <= 4096.
 * rtn_distance_transform3: the three distance transforms + merge + saturate of a caller-provided binary image.
 * rtn_resize_cubic: utils.resize_image (model/utils.py:140-154) = cv2.resize(INTER_CUBIC) of one HxWxC image; src f32, or
 *   uint8 with the 'custom_tf' normalisation x/127.5-1 (model/utils.py:43-46) fused; dst f32/bf16 rows dst_row_stride apart,
 *   i.e. written straight into the zero-padded batch canvas of Generator.compute_inputs (csv_generator.py:320-336). */
size_t rtn_preprocess_dt3_workspace_bytes(int B, int H, int W);
int rtn_preprocess_dt3(rtn_handle_t h, const uint8_t* src, int channels, int B, int H, int W, uint8_t* dst, uint8_t* binary_out,
                       void* workspace, size_t workspace_bytes);
int rtn_distance_transform3(rtn_handle_t h, const uint8_t* binary, int B, int H, int W, uint8_t* dst, void* workspace,
                            size_t workspace_bytes);
int rtn_resize_cubic(rtn_handle_t h, const void* src, int src_dtype, int H, int W, int C, double scale, void* dst, int dst_dtype,
                     int Ho, int Wo, int64_t dst_row_stride);

/* transform.apply_transform (model/transform.py:343-362) = cv2.warpAffine(image, matrix[:2], dsize = image size, flags, borderMode,
 * borderValue) of one uint8 HxWxC page (C <= 4), as Generator.random_transform_group_entry calls it (csv_generator.py:248-265).
 *   inv_map6 (host, 6 doubles, row-major 2x3): the destination->source map, i.e. the matrix already inverted the way warpAffine
 *     inverts it when WARP_INVERSE_MAP is absent;  interpolation 0 = INTER_NEAREST, 1 = INTER_LINEAR (TransformParameters
 *     default, model/transform.py:289-299);  border_mode 0 constant / 1 replicate ('nearest', the default) / 2 reflect101 / 3 wrap
 *     (model/transform.py:301-309);  cval4 (host, may be NULL = zeros): per-channel fill for border_mode 0. */
int rtn_warp_affine_u8(rtn_handle_t h, const uint8_t* src, int H, int W, int C, const double* inv_map6, int interpolation,
                       int border_mode, const uint8_t* cval4, uint8_t* dst);

/* The graph layers as separate calls (model/layers.py): FilterDetections on explicit boxes, RegressBoxes, ClipBoxes,
 * UpsampleLike; and utils.preprocess_image (model/utils.py:19-47; mode 0 'tf', 1 'caffe', 2 'custom_tf'). */
int rtn_filter_detections(rtn_handle_t h, int B, int64_t N, int num_classes, const float* in_boxes, const float* classification,
                          float score_threshold, float nms_threshold, int max_detections, float* boxes, float* scores,
                          int32_t* labels, void* workspace, size_t workspace_bytes);
int rtn_regress_boxes(rtn_handle_t h, const float* anchors, const float* deltas, int64_t n_boxes, const float* mean4 /* host */,
                      const float* std4 /* host */, float* out);
int rtn_clip_boxes(rtn_handle_t h, const float* in, int64_t n_boxes, float width, float height, float* out);
int rtn_upsample_nearest(rtn_handle_t h, const void* src, void* dst, int dtype, int B, int Hs, int Ws, int Hd, int Wd, int C);
int rtn_preprocess_image(rtn_handle_t h, const void* src, int src_dtype, float* dst, int64_t n, int mode, float scale, float sub);

#ifdef __cplusplus
}
#endif
#endif /* RTN_H */
