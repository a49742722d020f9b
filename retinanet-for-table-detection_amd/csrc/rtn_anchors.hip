// rtn_anchors.hip — anchor generation and anchor -> ground-truth assignment.
//
// Bit-exact device restatement of the reference's NumPy host path:
//   generate_anchors      model/anchors.py:243-278
//   shift / anchors_for_shape  model/anchors.py:169-238
//   compute_overlap       model/utils.py:180-211   (f64 math, result stored as f32)
//   compute_gt_annotations model/anchors.py:96-117 (first-max argmax, f32 compares)
//   bbox_transform        model/anchors.py:282-313 (f64, stored as f32)
//   anchor_targets_bbox   model/anchors.py:36-92
// and of the in-graph float32 Anchors layer (model/layers.py:42-53, model/utils.py:51-80).
// Anchors are never read from memory: every thread rebuilds its anchor from its index.
// FP contraction is OFF in this file: the reference rounds every product and sum.
#include "rtn_internal.h"
#include "rtn_anchor_dev.h"
#include <cmath>

#pragma clang fp contract(off)

extern "C" int rtn_generate_anchors(double base_size, const double* ratios, int nratios, const double* scales, int nscales,
                                    double* out) {
    if (!ratios || !scales || !out || nratios < 1 || nscales < 1 || nratios * nscales > 16) return RTN_EINVAL;
    const int num = nratios * nscales;
    for (int i = 0; i < num; ++i) {
        // anchors[:, 2:] = base_size * tile(scales, (2, len(ratios))).T  — NumPy evaluates this product in
        // the dtype of `scales`: float32 for the reference's parameters (model/anchors.py:32), then widens.
        const double sc = scales[i % nscales];
        volatile double side;
        if ((double)(float)sc == sc) { volatile float sf = (float)base_size * (float)sc; side = (double)sf; }
        else side = base_size * sc;
        volatile double w = side;
        volatile double hh = side;
        volatile double area = w * hh;
        const double ratio = ratios[i / nscales];  // np.repeat(ratios, len(scales))
        volatile double q = area / ratio;
        volatile double ww = std::sqrt(q);
        volatile double hn = ww * ratio;
        volatile double hw = ww * 0.5, hhh = hn * 0.5;
        out[4 * i + 0] = 0.0 - hw;
        out[4 * i + 1] = 0.0 - hhh;
        out[4 * i + 2] = ww - hw;
        out[4 * i + 3] = hn - hhh;
    }
    return RTN_OK;
}

namespace {

// anchors_for_shape, f64: base + ((i + 0.5) * stride)
__device__ __forceinline__ void anchor_f64(const DevAnchorCfg& c, int n, double a[4]) {
    const AnchorIdx ai = locate(c, n);
    const double sx = ((double)ai.x + 0.5) * (double)c.stride[ai.level];
    const double sy = ((double)ai.y + 0.5) * (double)c.stride[ai.level];
    const double* b = c.base[ai.level][ai.a];
    a[0] = b[0] + sx; a[1] = b[1] + sy; a[2] = b[2] + sx; a[3] = b[3] + sy;
}

__global__ __launch_bounds__(256) void anchors_f64_kernel(const DevAnchorCfg c, double* __restrict__ out) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < c.total; n += gridDim.x * blockDim.x) {
        double a[4];
        anchor_f64(c, n, a);
        out[4ll * n + 0] = a[0]; out[4ll * n + 1] = a[1]; out[4ll * n + 2] = a[2]; out[4ll * n + 3] = a[3];
    }
}

__global__ __launch_bounds__(256) void anchors_f32_kernel(const DevAnchorCfg c, float* __restrict__ out) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < c.total; n += gridDim.x * blockDim.x) {
        const AnchorIdx ai = locate(c, n);
        // keras.backend.variable(generate_anchors(...)) -> float32; shifts in float32
        const float sx = ((float)ai.x + 0.5f) * (float)c.stride[ai.level];
        const float sy = ((float)ai.y + 0.5f) * (float)c.stride[ai.level];
        const double* b = c.base[ai.level][ai.a];
        reinterpret_cast<float4*>(out)[n] = make_float4((float)b[0] + sx, (float)b[1] + sy, (float)b[2] + sx, (float)b[3] + sy);
    }
}

// one thread per (image, anchor); grid.y = image
__global__ __launch_bounds__(256) void anchor_targets_kernel(const DevAnchorCfg c, int K, const double* __restrict__ gt_boxes,
                                                             const int* __restrict__ gt_labels, const int* __restrict__ gt_count,
                                                             const int* __restrict__ img_hw, float neg_ov, float pos_ov,
                                                             float* __restrict__ reg_out, float* __restrict__ lab_out,
                                                             const double* __restrict__ anchors_explicit, int n_explicit) {
    __shared__ double s_gt[RTN_MAX_GT][4];
    __shared__ double s_area[RTN_MAX_GT];
    __shared__ int s_lab[RTN_MAX_GT];
    const int b = blockIdx.y;
    int G = gt_count[b];
    G = G < 0 ? 0 : (G > RTN_MAX_GT ? RTN_MAX_GT : G);
    for (int i = threadIdx.x; i < G; i += blockDim.x) {
        const double* g = gt_boxes + ((long long)b * RTN_MAX_GT + i) * 4;
        s_gt[i][0] = g[0]; s_gt[i][1] = g[1]; s_gt[i][2] = g[2]; s_gt[i][3] = g[3];
        s_area[i] = (g[2] - g[0]) * (g[3] - g[1]);
        s_lab[i] = gt_labels[(long long)b * RTN_MAX_GT + i];
    }
    __syncthreads();
    const int img_h = img_hw[2 * b], img_w = img_hw[2 * b + 1];
    const int N = anchors_explicit ? n_explicit : c.total;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        double a[4];
        if (anchors_explicit) {
            a[0] = anchors_explicit[4ll * n]; a[1] = anchors_explicit[4ll * n + 1];
            a[2] = anchors_explicit[4ll * n + 2]; a[3] = anchors_explicit[4ll * n + 3];
        } else {
            anchor_f64(c, n, a);
        }
        float state = 0.f;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        int label = -1;
        if (G > 0) {
            const double area1 = (a[2] - a[0]) * (a[3] - a[1]);
            float best = 0.f;
            int arg = 0;
            for (int i = 0; i < G; ++i) {
                const double x1 = fmax(a[0], s_gt[i][0]);
                const double y1 = fmax(a[1], s_gt[i][1]);
                const double x2 = fmin(a[2], s_gt[i][2]);
                const double y2 = fmin(a[3], s_gt[i][3]);
                const double w = fmax(0.0, x2 - x1);
                const double hh = fmax(0.0, y2 - y1);
                const double inter = w * hh;
                const double uni = area1 + s_area[i] - inter;
                const float iou = (float)(inter / uni);          // result array is float32
                if (i == 0 || iou > best) { best = iou; arg = i; }  // np.argmax: first maximum
            }
            const bool positive = best >= pos_ov;
            const bool ignore = (best > neg_ov) && !positive;
            state = positive ? 1.f : (ignore ? -1.f : 0.f);
            if (positive) label = s_lab[arg];
            // bbox_transform over ALL anchors with their argmax box
            const double aw = a[2] - a[0], ah = a[3] - a[1];
            t[0] = (float)(((s_gt[arg][0] - a[0]) / aw) / 0.2);
            t[1] = (float)(((s_gt[arg][1] - a[1]) / ah) / 0.2);
            t[2] = (float)(((s_gt[arg][2] - a[2]) / aw) / 0.2);
            t[3] = (float)(((s_gt[arg][3] - a[3]) / ah) / 0.2);
        }
        // anchors whose centre lies outside this image's own extent are ignored
        const double cx = (a[0] + a[2]) / 2, cy = (a[1] + a[3]) / 2;
        if (cx >= (double)img_w || cy >= (double)img_h) state = -1.f;
        float* r = reg_out + ((long long)b * N + n) * 5;
        r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3]; r[4] = state;
        float* l = lab_out + ((long long)b * N + n) * (K + 1);
        for (int k = 0; k < K; ++k) l[k] = (k == label) ? 1.f : 0.f;
        l[K] = state;
    }
}

}  // namespace

extern "C" int rtn_anchors_f64(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, double* out) {
    if (!h) return RTN_EINVAL;
    if (!out) return rtn_fail(h, RTN_EINVAL, "anchors_f64: null out");
    DevAnchorCfg d;
    const int rc = make_dev_cfg(h, cfg, &d);
    if (rc) return rc;
    const unsigned g = (unsigned)((d.total + 255) / 256);
    hipLaunchKernelGGL(anchors_f64_kernel, dim3(g > 2048 ? 2048 : g), dim3(256), 0, h->stream, d, out);
    RTN_CHECK_LAUNCH(h, "anchors_f64_kernel");
    return RTN_OK;
}

extern "C" int rtn_anchors_f32(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, float* out) {
    if (!h) return RTN_EINVAL;
    if (!out || ((uintptr_t)out & 15)) return rtn_fail(h, RTN_EINVAL, "anchors_f32: null/unaligned out");
    DevAnchorCfg d;
    const int rc = make_dev_cfg(h, cfg, &d);
    if (rc) return rc;
    const unsigned g = (unsigned)((d.total + 255) / 256);
    hipLaunchKernelGGL(anchors_f32_kernel, dim3(g > 2048 ? 2048 : g), dim3(256), 0, h->stream, d, out);
    RTN_CHECK_LAUNCH(h, "anchors_f32_kernel");
    return RTN_OK;
}

extern "C" int rtn_anchor_targets(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, int B, int num_classes, const double* gt_boxes,
                                  const int32_t* gt_labels, const int32_t* gt_count, const int32_t* img_hw,
                                  double negative_overlap, double positive_overlap, float* regression_batch,
                                  float* labels_batch) {
    if (!h) return RTN_EINVAL;
    if (B < 1 || B > 65535 || num_classes < 1) return rtn_fail(h, RTN_EINVAL, "anchor_targets: B %d / classes %d", B, num_classes);
    if (!gt_boxes || !gt_labels || !gt_count || !img_hw || !regression_batch || !labels_batch)
        return rtn_fail(h, RTN_EINVAL, "anchor_targets: null pointer");
    DevAnchorCfg d;
    const int rc = make_dev_cfg(h, cfg, &d);
    if (rc) return rc;
    unsigned gx = (unsigned)((d.total + 255) / 256);
    if (gx > 1024) gx = 1024;
    // thresholds compare against a float32 array: NumPy casts the Python float to float32
    hipLaunchKernelGGL(anchor_targets_kernel, dim3(gx, B), dim3(256), 0, h->stream, d, num_classes, gt_boxes, gt_labels,
                       gt_count, img_hw, (float)negative_overlap, (float)positive_overlap, regression_batch, labels_batch,
                       (const double*)nullptr, 0);
    RTN_CHECK_LAUNCH(h, "anchor_targets_kernel");
    return RTN_OK;
}

extern "C" int rtn_anchor_targets_explicit(rtn_handle_t h, const double* anchors, int N, int B, int num_classes, const double* gt_boxes,
                                           const int32_t* gt_labels, const int32_t* gt_count, const int32_t* img_hw,
                                           double negative_overlap, double positive_overlap, float* regression_batch,
                                           float* labels_batch) {
    if (!h) return RTN_EINVAL;
    if (N < 1 || B < 1 || B > 65535 || num_classes < 1) return rtn_fail(h, RTN_EINVAL, "anchor_targets_explicit: N %d B %d classes %d", N, B, num_classes);
    if (!anchors || !gt_boxes || !gt_labels || !gt_count || !img_hw || !regression_batch || !labels_batch)
        return rtn_fail(h, RTN_EINVAL, "anchor_targets_explicit: null pointer");
    DevAnchorCfg d;
    memset(&d, 0, sizeof(d));
    unsigned gx = (unsigned)((N + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(anchor_targets_kernel, dim3(gx, B), dim3(256), 0, h->stream, d, num_classes, gt_boxes, gt_labels, gt_count,
                       img_hw, (float)negative_overlap, (float)positive_overlap, regression_batch, labels_batch, anchors, N);
    RTN_CHECK_LAUNCH(h, "anchor_targets_kernel");
    return RTN_OK;
}

namespace {
// model/utils.py:180-211: IoU of every (box, gt) pair, f64 math, f32 result
__global__ __launch_bounds__(256) void compute_overlap_kernel(const double* __restrict__ boxes, const double* __restrict__ gts, int N,
                                                              int G, float* __restrict__ out) {
    const long long total = (long long)N * G;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / G), g = (int)(i - (long long)n * G);
        const double* a = boxes + 4ll * n;
        const double* b = gts + 4ll * g;
        const double area1 = (a[2] - a[0]) * (a[3] - a[1]);
        const double area2 = (b[2] - b[0]) * (b[3] - b[1]);
        const double w = fmax(0.0, fmin(a[2], b[2]) - fmax(a[0], b[0]));
        const double hh = fmax(0.0, fmin(a[3], b[3]) - fmax(a[1], b[1]));
        const double inter = w * hh;
        out[i] = (float)(inter / (area1 + area2 - inter));
    }
}

// model/anchors.py:96-117 on a precomputed f32 overlap matrix: first-max argmax, >= pos, > neg & !pos
__global__ __launch_bounds__(256) void gt_annotations_kernel(const float* __restrict__ ov, int N, int G, float neg_ov, float pos_ov,
                                                             unsigned char* __restrict__ positive, unsigned char* __restrict__ ignore,
                                                             long long* __restrict__ argmax) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        float best = ov[(long long)n * G];
        int arg = 0;
        for (int g = 1; g < G; ++g) {
            const float v = ov[(long long)n * G + g];
            if (v > best) { best = v; arg = g; }
        }
        const bool pos = best >= pos_ov;
        positive[n] = pos ? 1 : 0;
        ignore[n] = (best > neg_ov && !pos) ? 1 : 0;
        argmax[n] = arg;
    }
}

// model/anchors.py:282-313: ((gt - anchor) / anchor extent - mean) / std, f64
__global__ __launch_bounds__(256) void bbox_transform_kernel(const double* __restrict__ anchors, const double* __restrict__ gt, int N,
                                                             double m0, double m1, double m2, double m3, double s0, double s1, double s2,
                                                             double s3, double* __restrict__ out) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const double* a = anchors + 4ll * n;
        const double* g = gt + 4ll * n;
        const double aw = a[2] - a[0], ah = a[3] - a[1];
        out[4ll * n + 0] = (((g[0] - a[0]) / aw) - m0) / s0;
        out[4ll * n + 1] = (((g[1] - a[1]) / ah) - m1) / s1;
        out[4ll * n + 2] = (((g[2] - a[2]) / aw) - m2) / s2;
        out[4ll * n + 3] = (((g[3] - a[3]) / ah) - m3) / s3;
    }
}
}  // namespace

extern "C" int rtn_compute_overlap(rtn_handle_t h, const double* boxes, const double* gts, int N, int G, float* out) {
    if (!h) return RTN_EINVAL;
    if (!boxes || !gts || !out || N < 1 || G < 1) return rtn_fail(h, RTN_EINVAL, "compute_overlap: bad argument");
    long long g = ((long long)N * G + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(compute_overlap_kernel, dim3((unsigned)g), dim3(256), 0, h->stream, boxes, gts, N, G, out);
    RTN_CHECK_LAUNCH(h, "compute_overlap_kernel");
    return RTN_OK;
}

extern "C" int rtn_gt_annotations(rtn_handle_t h, const float* overlaps, int N, int G, double negative_overlap, double positive_overlap,
                                  uint8_t* positive, uint8_t* ignore, int64_t* argmax) {
    if (!h) return RTN_EINVAL;
    if (!overlaps || !positive || !ignore || !argmax || N < 1 || G < 1) return rtn_fail(h, RTN_EINVAL, "gt_annotations: bad argument");
    unsigned g = (unsigned)((N + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(gt_annotations_kernel, dim3(g), dim3(256), 0, h->stream, overlaps, N, G, (float)negative_overlap,
                       (float)positive_overlap, positive, ignore, (long long*)argmax);
    RTN_CHECK_LAUNCH(h, "gt_annotations_kernel");
    return RTN_OK;
}

extern "C" int rtn_bbox_transform(rtn_handle_t h, const double* anchors, const double* gt_boxes, int N, const double* mean4,
                                  const double* std4, double* out) {
    if (!h) return RTN_EINVAL;
    if (!anchors || !gt_boxes || !mean4 || !std4 || !out || N < 1) return rtn_fail(h, RTN_EINVAL, "bbox_transform: bad argument");
    unsigned g = (unsigned)((N + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(bbox_transform_kernel, dim3(g), dim3(256), 0, h->stream, anchors, gt_boxes, N, mean4[0], mean4[1], mean4[2], mean4[3],
                       std4[0], std4[1], std4[2], std4[3], out);
    RTN_CHECK_LAUNCH(h, "bbox_transform_kernel");
    return RTN_OK;
}
