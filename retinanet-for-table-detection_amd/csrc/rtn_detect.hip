// rtn_detect.hip — box decode + clip + score threshold + greedy NMS + top-k + pad.
//
// Restates (float32, no FP contraction) what the reference's inference graph does after the
// two head outputs (model/defineModel.py:329-350):
//   Anchors layer      model/layers.py:42-53 + model/utils.py:51-80  (anchor rebuilt from index)
//   RegressBoxes       model/utils.py:84-112   x1 = ax1 + (d0*0.2 + 0)*w ...
//   ClipBoxes          model/layers.py:157-171 clip to [0,W] x [0,H] of the padded canvas
//   filter_detections  model/layers.py:200-264 per class: score > 0.05 ->
//                      tf.image.non_max_suppression(max 300, IoU > 0.5 suppresses, score-descending,
//                      equal scores: lower index first) -> concat -> tf.nn.top_k -> pad -1.
//
// Stage 1 (grid-wide): threshold, append 64-bit keys {score bits | ~index} per (image, class).
// Stage 2 (one 1024-thread workgroup per (image, class)): take candidates in exact descending
//   key order in batches of <= 4096 (radix select on the keys when more remain), bitonic-sort a
//   batch in LDS, decode its boxes into LDS, drop those overlapping already-kept boxes, then
//   resolve 64 candidates at a time inside one wavefront (ballot / readlane) and broadcast the
//   newly kept boxes to the rest of the batch.  Exact TF semantics for ANY candidate count.
// Stage 3 (one workgroup per image): top-k merge over classes, pad with -1.
#include "rtn_internal.h"
#include "rtn_anchor_dev.h"

#pragma clang fp contract(off)

namespace {

constexpr int NMS_T = 1024;
constexpr int CAP = 4096;       // candidates resolved per batch
constexpr int MCAP = 8192;      // classes * max_detections handled by the merge

typedef unsigned long long u64;

__device__ __forceinline__ float4 decode_box(const DevAnchorCfg& c, const float* __restrict__ regression, long long img_base,
                                             int idx, float cw, float ch) {
    const AnchorIdx ai = locate(c, idx);
    const float sx = ((float)ai.x + 0.5f) * (float)c.stride[ai.level];
    const float sy = ((float)ai.y + 0.5f) * (float)c.stride[ai.level];
    const double* bb = c.base[ai.level][ai.a];
    const float ax1 = (float)bb[0] + sx, ay1 = (float)bb[1] + sy, ax2 = (float)bb[2] + sx, ay2 = (float)bb[3] + sy;
    const float4 d = *reinterpret_cast<const float4*>(regression + (img_base + idx) * 4);
    const float w = ax2 - ax1, hgt = ay2 - ay1;
    float x1 = ax1 + (d.x * 0.2f + 0.0f) * w;
    float y1 = ay1 + (d.y * 0.2f + 0.0f) * hgt;
    float x2 = ax2 + (d.z * 0.2f + 0.0f) * w;
    float y2 = ay2 + (d.w * 0.2f + 0.0f) * hgt;
    x1 = fminf(fmaxf(x1, 0.f), cw); y1 = fminf(fmaxf(y1, 0.f), ch);
    x2 = fminf(fmaxf(x2, 0.f), cw); y2 = fminf(fmaxf(y2, 0.f), ch);
    return make_float4(x1, y1, x2, y2);
}

// tf.image.non_max_suppression's IoU (corner order normalised, zero for empty boxes)
__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
    const float ax1 = fminf(a.x, a.z), ax2 = fmaxf(a.x, a.z), ay1 = fminf(a.y, a.w), ay2 = fmaxf(a.y, a.w);
    const float bx1 = fminf(b.x, b.z), bx2 = fmaxf(b.x, b.z), by1 = fminf(b.y, b.w), by2 = fmaxf(b.y, b.w);
    const float area_a = (ay2 - ay1) * (ax2 - ax1);
    const float area_b = (by2 - by1) * (bx2 - bx1);
    if (area_a <= 0.f || area_b <= 0.f) return false;
    const float iy1 = fmaxf(ay1, by1), ix1 = fmaxf(ax1, bx1), iy2 = fminf(ay2, by2), ix2 = fminf(ax2, bx2);
    const float inter = fmaxf(iy2 - iy1, 0.f) * fmaxf(ix2 - ix1, 0.f);
    const float iou = inter / (area_a + area_b - inter);
    return iou > thr;
}

// grid = (ceil(N/256), B*K)
__global__ __launch_bounds__(256) void detect_candidates_kernel(int N, int K, const float* __restrict__ cls, float thr,
                                                                u64* __restrict__ keys, int* __restrict__ counts) {
    const int bk = blockIdx.y;
    const int b = bk / K, k = bk - b * K;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    float score = 0.f;
    bool pass = false;
    if (n < N) {
        score = cls[((long long)b * N + n) * K + k];
        pass = score > thr;
    }
    const u64 mask = __ballot(pass);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(&counts[bk], __popcll(mask));
    base = __shfl(base, 0, 64);
    if (pass) {
        const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        keys[(long long)bk * N + pos] = ((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - (unsigned)n);
    }
}

__device__ __forceinline__ void bitonic_sort_desc(u64* s, int P, int t, int nthreads) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u64 a = s[i], b = s[ixj];
                    const bool desc = ((i & k) == 0);
                    if (desc ? (a < b) : (a > b)) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(NMS_T) void nms_kernel(const DevAnchorCfg c, int K, const float* __restrict__ regression,
                                                    const u64* __restrict__ keys_all, const int* __restrict__ counts, float cw,
                                                    float ch, float iou_thr, int max_det, u64* __restrict__ sel_keys,
                                                    float4* __restrict__ sel_boxes, int* __restrict__ sel_count,
                                                    const float4* __restrict__ boxes_explicit, int n_explicit) {
    __shared__ u64 s_keys[CAP];
    __shared__ float4 s_box[CAP];
    __shared__ unsigned char s_removed[CAP];
    __shared__ float4 s_kbox[RTN_MAX_DET];
    __shared__ u64 s_kkey[RTN_MAX_DET];
    __shared__ unsigned s_hist[256];
    __shared__ int s_nb, s_kept, s_newbeg, s_need;
    __shared__ u64 s_prefix;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bk = blockIdx.x;
    const int b = bk / K;
    const int N = boxes_explicit ? n_explicit : c.total;
    const long long img_base = (long long)b * N;
    const u64* keys = keys_all + (long long)bk * N;
    int count = counts[bk];
    count = count < 0 ? 0 : (count > N ? N : count);

    if (t == 0) { s_kept = 0; s_newbeg = 0; }
    __syncthreads();

    bool upper_valid = false;
    u64 upper = 0;
    int remaining = count;
    while (true) {
        const int kept_now = s_kept;
        if (kept_now >= max_det || remaining <= 0) break;
        // ---- lower bound T of this batch: the CAP-th largest key below `upper`
        u64 T = 0;
        if (remaining > CAP) {
            if (t == 0) { s_prefix = 0; s_need = CAP; }
            __syncthreads();
            for (int d = 7; d >= 0; --d) {
                const int shift = 8 * d;
                for (int i = t; i < 256; i += NMS_T) s_hist[i] = 0;
                __syncthreads();
                const u64 prefix = s_prefix;
                for (int i = t; i < count; i += NMS_T) {
                    const u64 key = keys[i];
                    if (upper_valid && key >= upper) continue;
                    if (d < 7 && (key >> (shift + 8)) != (prefix >> (shift + 8))) continue;
                    atomicAdd(&s_hist[(unsigned)(key >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (t == 0) {
                    int need = s_need, acc = 0, digit = 0;
                    for (int bin = 255; bin >= 0; --bin) {
                        const int hcount = (int)s_hist[bin];
                        if (acc + hcount >= need) { digit = bin; break; }
                        acc += hcount;
                    }
                    s_need = need - acc;
                    s_prefix = prefix | ((u64)digit << shift);
                }
                __syncthreads();
            }
            T = s_prefix;
        }
        // ---- gather the batch
        if (t == 0) s_nb = 0;
        __syncthreads();
        for (int i = t; i < count; i += NMS_T) {
            const u64 key = keys[i];
            if (key >= T && (!upper_valid || key < upper)) {
                const int pos = atomicAdd(&s_nb, 1);
                if (pos < CAP) s_keys[pos] = key;
            }
        }
        __syncthreads();
        int nb = s_nb;
        nb = nb > CAP ? CAP : nb;
        if (nb <= 0) break;
        int P = 64;
        while (P < nb) P <<= 1;
        for (int i = nb + t; i < P; i += NMS_T) s_keys[i] = 0;
        __syncthreads();
        bitonic_sort_desc(s_keys, P, t, NMS_T);
        // ---- decode boxes; drop candidates overlapping boxes kept by earlier batches
        for (int i = t; i < nb; i += NMS_T) {
            const int idx = (int)(0xFFFFFFFFu - (unsigned)(s_keys[i] & 0xFFFFFFFFull));
            const float4 bx = boxes_explicit ? boxes_explicit[img_base + idx] : decode_box(c, regression, img_base, idx, cw, ch);
            s_box[i] = bx;
            unsigned char rem = 0;
            for (int j = 0; j < kept_now; ++j)
                if (iou_gt(bx, s_kbox[j], iou_thr)) { rem = 1; break; }
            s_removed[i] = rem;
        }
        __syncthreads();
        // ---- 64 candidates at a time
        const int nchunks = (nb + 63) >> 6;
        int kept_reg = kept_now;                     // s_kept as of the last barrier (wave 0 rewrites it below)
        for (int ci = 0; ci < nchunks; ++ci) {
            if (kept_reg >= max_det) break;          // uniform
            if (wave == 0) {
                const int i = ci * 64 + lane;
                const bool valid = (i < nb) && !s_removed[i];
                const float4 bx = valid ? s_box[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                const int kept0 = s_kept;
                bool alive = valid;
                int keptn = kept0;
                u64 keepmask = 0;
                u64 m = __ballot(alive);
                while (m != 0 && keptn < max_det) {
                    const int j = __ffsll((long long)m) - 1;
                    keepmask |= (1ull << j);
                    ++keptn;
                    const float4 bj = make_float4(__shfl(bx.x, j, 64), __shfl(bx.y, j, 64), __shfl(bx.z, j, 64), __shfl(bx.w, j, 64));
                    if (alive && lane > j && iou_gt(bx, bj, iou_thr)) alive = false;
                    const u64 below = (j == 63) ? ~0ull : ((2ull << j) - 1ull);
                    m = __ballot(alive) & ~below;
                }
                if ((keepmask >> lane) & 1ull) {
                    const int pos = kept0 + __popcll(keepmask & ((1ull << lane) - 1ull));
                    s_kbox[pos] = bx;
                    s_kkey[pos] = s_keys[i];
                }
                if (lane == 0) { s_newbeg = kept0; s_kept = keptn; }
            }
            __syncthreads();
            const int k0 = s_newbeg, k1 = s_kept;
            kept_reg = k1;
            if (k1 > k0) {
                for (int i = (ci + 1) * 64 + t; i < nb; i += NMS_T) {
                    if (s_removed[i]) continue;
                    const float4 bx = s_box[i];
                    for (int j = k0; j < k1; ++j)
                        if (iou_gt(bx, s_kbox[j], iou_thr)) { s_removed[i] = 1; break; }
                }
            }
            __syncthreads();
        }
        upper = T;
        upper_valid = true;
        remaining -= nb;
        if (T == 0) break;
    }
    __syncthreads();
    const int kept = s_kept < max_det ? s_kept : max_det;
    for (int j = t; j < kept; j += NMS_T) {
        sel_keys[(long long)bk * max_det + j] = s_kkey[j];
        sel_boxes[(long long)bk * max_det + j] = s_kbox[j];
    }
    if (t == 0) sel_count[bk] = kept;
}

// one workgroup per image: concat classes, top-k by (score desc, concat position asc), pad -1
__global__ __launch_bounds__(NMS_T) void merge_topk_kernel(int K, int max_det, const u64* __restrict__ sel_keys,
                                                           const float4* __restrict__ sel_boxes, const int* __restrict__ sel_count,
                                                           float* __restrict__ boxes, float* __restrict__ scores,
                                                           int* __restrict__ labels) {
    __shared__ u64 s_keys[MCAP];
    __shared__ unsigned short s_cls[MCAP];
    __shared__ unsigned short s_j[MCAP];
    __shared__ int s_total;
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    if (t == 0) {
        int total = 0;
        for (int k = 0; k < K; ++k) {
            const int cnt = sel_count[b * K + k];
            for (int j = 0; j < cnt && total < MCAP; ++j, ++total) { s_cls[total] = (unsigned short)k; s_j[total] = (unsigned short)j; }
        }
        s_total = total;
    }
    __syncthreads();
    const int total = s_total;
    int P = 64;
    while (P < total) P <<= 1;
    for (int i = t; i < P; i += NMS_T) {
        u64 key = 0;
        if (i < total) {
            const u64 sk = sel_keys[((long long)b * K + s_cls[i]) * max_det + s_j[i]];
            key = (sk & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - (unsigned)i);
        }
        s_keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_desc(s_keys, P, t, NMS_T);
    const int nout = total < max_det ? total : max_det;
    for (int r = t; r < max_det; r += NMS_T) {
        float4 bx = make_float4(-1.f, -1.f, -1.f, -1.f);
        float sc = -1.f;
        int lb = -1;
        if (r < nout) {
            const u64 key = s_keys[r];
            const int pos = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
            const int k = s_cls[pos], j = s_j[pos];
            bx = sel_boxes[((long long)b * K + k) * max_det + j];
            sc = __uint_as_float((unsigned)(key >> 32));
            lb = k;
        }
        reinterpret_cast<float4*>(boxes)[(long long)b * max_det + r] = bx;
        scores[(long long)b * max_det + r] = sc;
        labels[(long long)b * max_det + r] = lb;
    }
}

struct WsLayout { size_t keys, counts, sel_keys, sel_boxes, sel_count, total; };

WsLayout ws_layout(int B, long long N, int K) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WsLayout w;
    size_t o = 0;
    w.keys = o;      o += al((size_t)B * K * N * 8);
    w.counts = o;    o += al((size_t)B * K * 4 * 2);      // candidate counts then selected counts
    w.sel_count = w.counts + (size_t)B * K * 4;
    w.sel_keys = o;  o += al((size_t)B * K * RTN_MAX_DET * 8);
    w.sel_boxes = o; o += al((size_t)B * K * RTN_MAX_DET * 16);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t rtn_detect_workspace_bytes(int B, int64_t N, int num_classes) {
    if (B < 1 || N < 1 || num_classes < 1) return 0;
    return ws_layout(B, N, num_classes).total;
}

static int detect_launch(rtn_handle_t h, const DevAnchorCfg& d, int N, int B, int num_classes, const float* regression,
                         const float4* boxes_explicit, const float* classification, int canvas_h, int canvas_w, float score_threshold,
                         float nms_threshold, int max_detections, float* boxes, float* scores, int32_t* labels, void* workspace,
                         size_t workspace_bytes) {
    const WsLayout w = ws_layout(B, N, num_classes);
    if (workspace_bytes < w.total) return rtn_fail(h, RTN_ENOMEM, "detect: workspace %zu < %zu", workspace_bytes, w.total);
    char* ws = (char*)workspace;
    u64* keys = (u64*)(ws + w.keys);
    int* counts = (int*)(ws + w.counts);
    int* sel_count = (int*)(ws + w.sel_count);
    u64* sel_keys = (u64*)(ws + w.sel_keys);
    float4* sel_boxes = (float4*)(ws + w.sel_boxes);
    const int BK = B * num_classes;
    RTN_HIP(h, hipMemsetAsync(counts, 0, (size_t)BK * 4 * 2, h->stream));
    hipLaunchKernelGGL(detect_candidates_kernel, dim3((N + 255) / 256, BK), dim3(256), 0, h->stream, N, num_classes,
                       classification, score_threshold, keys, counts);
    RTN_CHECK_LAUNCH(h, "detect_candidates_kernel");
    hipLaunchKernelGGL(nms_kernel, dim3(BK), dim3(NMS_T), 0, h->stream, d, num_classes, regression, (const u64*)keys,
                       (const int*)counts, (float)canvas_w, (float)canvas_h, nms_threshold, max_detections, sel_keys, sel_boxes,
                       sel_count, boxes_explicit, N);
    RTN_CHECK_LAUNCH(h, "nms_kernel");
    hipLaunchKernelGGL(merge_topk_kernel, dim3(B), dim3(NMS_T), 0, h->stream, num_classes, max_detections, (const u64*)sel_keys,
                       (const float4*)sel_boxes, (const int*)sel_count, boxes, scores, labels);
    RTN_CHECK_LAUNCH(h, "merge_topk_kernel");
    return RTN_OK;
}

static int detect_check(rtn_handle_t h, int B, int num_classes, float score_threshold, int max_detections, const void* a, const void* b,
                        const void* boxes, const void* scores, const void* labels, const void* workspace) {
    if (B < 1 || num_classes < 1 || (long long)B * num_classes > 65535) return rtn_fail(h, RTN_EINVAL, "detect: B %d classes %d", B, num_classes);
    if (max_detections < 1 || max_detections > RTN_MAX_DET) return rtn_fail(h, RTN_EINVAL, "detect: max_detections %d not in [1,%d]", max_detections, RTN_MAX_DET);
    if ((long long)num_classes * max_detections > MCAP) return rtn_fail(h, RTN_EINVAL, "detect: classes*max_detections > %d", MCAP);
    if (!(score_threshold >= 0.f)) return rtn_fail(h, RTN_EINVAL, "detect: score_threshold must be >= 0");
    if (!a || !b || !boxes || !scores || !labels || !workspace) return rtn_fail(h, RTN_EINVAL, "detect: null pointer");
    if (((uintptr_t)a & 15) || ((uintptr_t)boxes & 15) || ((uintptr_t)workspace & 255))
        return rtn_fail(h, RTN_EINVAL, "detect: regression/boxes must be 16-byte and workspace 256-byte aligned");
    return RTN_OK;
}

extern "C" int rtn_decode_filter_nms(rtn_handle_t h, const rtn_anchor_cfg_t* cfg, int B, int num_classes, const float* regression,
                                     const float* classification, int canvas_h, int canvas_w, float score_threshold,
                                     float nms_threshold, int max_detections, float* boxes, float* scores, int32_t* labels,
                                     void* workspace, size_t workspace_bytes) {
    if (!h) return RTN_EINVAL;
    int rc = detect_check(h, B, num_classes, score_threshold, max_detections, regression, classification, boxes, scores, labels, workspace);
    if (rc) return rc;
    DevAnchorCfg d;
    rc = make_dev_cfg(h, cfg, &d);
    if (rc) return rc;
    return detect_launch(h, d, d.total, B, num_classes, regression, nullptr, classification, canvas_h, canvas_w, score_threshold,
                         nms_threshold, max_detections, boxes, scores, labels, workspace, workspace_bytes);
}

/* FilterDetections on explicit boxes (model/layers.py:177-264, 267-332) */
extern "C" int rtn_filter_detections(rtn_handle_t h, int B, int64_t N, int num_classes, const float* in_boxes, const float* classification,
                                     float score_threshold, float nms_threshold, int max_detections, float* boxes, float* scores,
                                     int32_t* labels, void* workspace, size_t workspace_bytes) {
    if (!h) return RTN_EINVAL;
    if (N < 1 || N > (1 << 30)) return rtn_fail(h, RTN_EINVAL, "filter_detections: N");
    const int rc = detect_check(h, B, num_classes, score_threshold, max_detections, in_boxes, classification, boxes, scores, labels, workspace);
    if (rc) return rc;
    DevAnchorCfg d;
    memset(&d, 0, sizeof(d));
    return detect_launch(h, d, (int)N, B, num_classes, nullptr, (const float4*)in_boxes, classification, 0, 0, score_threshold,
                         nms_threshold, max_detections, boxes, scores, labels, workspace, workspace_bytes);
}

namespace {
// RegressBoxes: model/utils.py:84-112 with explicit mean/std, float32 op by op
__global__ __launch_bounds__(256) void regress_boxes_kernel(const float4* __restrict__ anchors, const float4* __restrict__ deltas, long long n,
                                                            float m0, float m1, float m2, float m3, float s0, float s1, float s2, float s3,
                                                            float4* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = anchors[i], d = deltas[i];
        const float w = a.z - a.x, hh = a.w - a.y;
        out[i] = make_float4(a.x + (d.x * s0 + m0) * w, a.y + (d.y * s1 + m1) * hh, a.z + (d.z * s2 + m2) * w, a.w + (d.w * s3 + m3) * hh);
    }
}
// ClipBoxes: model/layers.py:157-171
__global__ __launch_bounds__(256) void clip_boxes_kernel(const float4* __restrict__ in, long long n, float width, float height,
                                                         float4* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float4 b = in[i];
        out[i] = make_float4(fminf(fmaxf(b.x, 0.f), width), fminf(fmaxf(b.y, 0.f), height), fminf(fmaxf(b.z, 0.f), width),
                             fminf(fmaxf(b.w, 0.f), height));
    }
}
}  // namespace

extern "C" int rtn_regress_boxes(rtn_handle_t h, const float* anchors, const float* deltas, int64_t n_boxes, const float* mean4,
                                 const float* std4, float* out) {
    if (!h) return RTN_EINVAL;
    if (!anchors || !deltas || !mean4 || !std4 || !out || n_boxes < 1) return rtn_fail(h, RTN_EINVAL, "regress_boxes: bad argument");
    if (((uintptr_t)anchors & 15) || ((uintptr_t)deltas & 15) || ((uintptr_t)out & 15)) return rtn_fail(h, RTN_EINVAL, "regress_boxes: alignment");
    long long g = (n_boxes + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(regress_boxes_kernel, dim3((unsigned)g), dim3(256), 0, h->stream, (const float4*)anchors, (const float4*)deltas,
                       (long long)n_boxes, mean4[0], mean4[1], mean4[2], mean4[3], std4[0], std4[1], std4[2], std4[3], (float4*)out);
    RTN_CHECK_LAUNCH(h, "regress_boxes_kernel");
    return RTN_OK;
}

extern "C" int rtn_clip_boxes(rtn_handle_t h, const float* in, int64_t n_boxes, float width, float height, float* out) {
    if (!h) return RTN_EINVAL;
    if (!in || !out || n_boxes < 1) return rtn_fail(h, RTN_EINVAL, "clip_boxes: bad argument");
    if (((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return rtn_fail(h, RTN_EINVAL, "clip_boxes: alignment");
    long long g = (n_boxes + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(clip_boxes_kernel, dim3((unsigned)g), dim3(256), 0, h->stream, (const float4*)in, (long long)n_boxes, width, height, (float4*)out);
    RTN_CHECK_LAUNCH(h, "clip_boxes_kernel");
    return RTN_OK;
}
