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
// Stage 2: candidates are taken in exact descending key order in batches of <= 4096 (radix select on the
//   keys when more remain).  First batch: (a) one workgroup per (image, class) selects, sorts (register /
//   shuffle / LDS bitonic network) and decodes it; (b) ALL CUs build its 4096 x 4096 suppression bit
//   matrix, one wavefront per 64 x 64 tile; (c) one wavefront per (image, class) walks the matrix greedily
//   (scalar bit operations on the diagonal tile, OR of the kept rows).  Later batches (rare: more than
//   4096 candidates and fewer than max_detections kept) run inside the scan workgroup: select, sort,
//   decode, drop those overlapping kept boxes, resolve 64 at a time (ballot / readlane), broadcast the
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

// tf.image.non_max_suppression's IoU (corner order normalised, zero for empty boxes), on pre-normalised boxes:
// NBox = {min x, min y, max x, max y} + area, built once per box instead of once per pair.
struct NBox { float x1, y1, x2, y2, area; };

__device__ __forceinline__ NBox nbox(const float4 a) {
    NBox n;
    n.x1 = fminf(a.x, a.z); n.x2 = fmaxf(a.x, a.z); n.y1 = fminf(a.y, a.w); n.y2 = fmaxf(a.y, a.w);
    n.area = (n.y2 - n.y1) * (n.x2 - n.x1);
    return n;
}

// iou(a, b) > thr with the float32 operation order of the TF kernel (inter / (area_a + area_b - inter)).  The quotient
// is first estimated with v_rcp_f32 (1 ulp): only when the estimate lies within 1e-5 (relative) of the threshold is the
// correctly rounded division evaluated, so the decision is the IEEE one for every pair at a fraction of the cost.
__device__ __forceinline__ bool iou_gt(const NBox& a, const NBox& b, float thr) {
    if (a.area <= 0.f || b.area <= 0.f) return false;
    const float iy1 = fmaxf(a.y1, b.y1), ix1 = fmaxf(a.x1, b.x1), iy2 = fminf(a.y2, b.y2), ix2 = fminf(a.x2, b.x2);
    const float inter = fmaxf(iy2 - iy1, 0.f) * fmaxf(ix2 - ix1, 0.f);
    if (inter <= 0.f && thr >= 0.f) return false;           // iou == 0: most pairs of a page
    const float uni = a.area + b.area - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    if (fabsf(q - thr) > 1e-5f * fabsf(thr) + 1e-30f && q == q && fabsf(q) < 3.0e38f) return q > thr;
    return inter / uni > thr;
}

// Same decision without early exits (the mask kernel's 64-column loop: straight-line code issues faster than three
// divergent branches per pair); the exact division only runs for lanes whose estimate is within the margin.
__device__ __forceinline__ bool iou_gt_flat(const NBox& a, const NBox& b, float thr) {
    const bool area_ok = a.area > 0.f && b.area > 0.f;
    const float iy1 = fmaxf(a.y1, b.y1), ix1 = fmaxf(a.x1, b.x1), iy2 = fminf(a.y2, b.y2), ix2 = fminf(a.x2, b.x2);
    const float inter = fmaxf(iy2 - iy1, 0.f) * fmaxf(ix2 - ix1, 0.f);
    const float uni = a.area + b.area - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    bool hit = q > thr;
    const bool sure = fabsf(q - thr) > 1e-5f * fabsf(thr) + 1e-30f && fabsf(q) < 3.0e38f;     // false for NaN
    if (!sure && area_ok) hit = inter / uni > thr;
    return hit && area_ok;
}

// grid = (ceil(N/CAND_PER_BLOCK), B*K), 1024 threads x 4 anchors.  One global atomic per workgroup reserves the
// block's slice of the (image, class) key list: a per-wave atomic made ~1500 same-address atomics per image at a
// 1 % candidate rate and cost 50 us; the list order is irrelevant (keys are sorted later).
constexpr int CAND_T = 1024, CAND_R = 4, CAND_PER_BLOCK = CAND_T * CAND_R;
__global__ __launch_bounds__(CAND_T) void detect_candidates_kernel(int N, int K, const float* __restrict__ cls, float thr,
                                                                   u64* __restrict__ keys, int* __restrict__ counts) {
    __shared__ int s_cnt[CAND_R * (CAND_T / 64)];
    const int bk = blockIdx.y;
    const int b = bk / K, k = bk - b * K;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float score[CAND_R];
    u64 mask[CAND_R];
    bool pass[CAND_R];
#pragma unroll
    for (int r = 0; r < CAND_R; ++r) {
        const int n = blockIdx.x * CAND_PER_BLOCK + r * CAND_T + t;
        score[r] = 0.f;
        pass[r] = false;
        if (n < N) {
            score[r] = cls[((long long)b * N + n) * K + k];
            pass[r] = score[r] > thr;
        }
        mask[r] = __ballot(pass[r]);
        if (lane == 0) s_cnt[r * (CAND_T / 64) + wave] = __popcll(mask[r]);
    }
    __syncthreads();
    if (wave == 0) {                                  // 64 (round, wave) counts -> exclusive offsets, one atomic
        const int v = s_cnt[lane];
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        int base = 0;
        if (lane == 0 && total > 0) base = atomicAdd(&counts[bk], total);
        base = __shfl(base, 0, 64);
        s_cnt[lane] = base + incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CAND_R; ++r) {
        if (pass[r]) {
            const int n = blockIdx.x * CAND_PER_BLOCK + r * CAND_T + t;
            const int pos = s_cnt[r * (CAND_T / 64) + wave] + __popcll(mask[r] & ((1ull << lane) - 1ull));
            keys[(long long)bk * N + pos] = ((u64)__float_as_uint(score[r]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)n);
        }
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

// ---- block-wide descending sort of P = 1024 | 2048 | 4096 keys, up to four per thread -------------------------------
// Element e = r * NMS_T + t lives in v[r] of thread t.  A bitonic network whose compare-exchange partner e ^ j is
//   * the same thread's other register      (j >= 1024): no communication,
//   * another lane of the same wavefront    (j <  64):   one xor-shuffle,
//   * a thread of another wavefront         (64..512):   one LDS round (double-buffered: one barrier per stage).
// 14..18 barriers for 2048..4096 keys; the all-LDS network needed 66..78 and an O(n^2) rank sort 80 us for 2048 keys.
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = (unsigned)__shfl_xor((int)lo, m, 64);
    hi = (unsigned)__shfl_xor((int)hi, m, 64);
    return ((u64)hi << 32) | (u64)lo;
}

__device__ __forceinline__ u64 cx_pick(u64 a, u64 b, int e, int j, int k) {
    const bool lower = (e & j) == 0, desc = (e & k) == 0;
    const u64 hi = a > b ? a : b, lo = a > b ? b : a;
    return (lower == desc) ? hi : lo;
}

constexpr int ROWS = 4;                              // CAP / NMS_T
__device__ __forceinline__ void bitonic_regs_desc(u64 (&v)[ROWS], int P, u64* buf0, u64* buf1, int t) {
    const int ER = P / NMS_T;                        // live registers: 1, 2 or 4
    int pb = 0;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= NMS_T) {
                if (j == NMS_T) {                    // pairs (0,1) (2,3)
                    const u64 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                    v[0] = cx_pick(a0, a1, t, j, k);             v[1] = cx_pick(a1, a0, NMS_T + t, j, k);
                    if (ER > 2) { v[2] = cx_pick(a2, a3, 2 * NMS_T + t, j, k); v[3] = cx_pick(a3, a2, 3 * NMS_T + t, j, k); }
                } else {                             // j == 2048: pairs (0,2) (1,3)
                    const u64 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                    v[0] = cx_pick(a0, a2, t, j, k);             v[2] = cx_pick(a2, a0, 2 * NMS_T + t, j, k);
                    v[1] = cx_pick(a1, a3, NMS_T + t, j, k);     v[3] = cx_pick(a3, a1, 3 * NMS_T + t, j, k);
                }
            } else if (j >= 64) {
                u64* buf = pb ? buf1 : buf0;
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    if (r < ER) buf[r * NMS_T + t] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    if (r < ER) {
                        const int e = r * NMS_T + t;
                        v[r] = cx_pick(v[r], buf[e ^ j], e, j, k);
                    }
                pb ^= 1;
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    if (r < ER) v[r] = cx_pick(v[r], shfl_xor_u64(v[r], j), r * NMS_T + t, j, k);
            }
        }
    }
}

struct SelShared {          // LDS scratch of the batch selection
    unsigned* hist;         // [256]
    int* nb;
    int* need;
    u64* prefix;
};

// Lower bound T of the next batch: the CAP-th largest key below `upper` (0 when everything left fits one batch).
__device__ __forceinline__ u64 select_threshold(const u64* __restrict__ keys, int count, int remaining, bool upper_valid, u64 upper,
                                                const SelShared& sh, int t) {
    if (remaining <= CAP) return 0;
    if (t == 0) { *sh.prefix = 0; *sh.need = CAP; }
    __syncthreads();
    for (int d = 7; d >= 0; --d) {
        const int shift = 8 * d;
        for (int i = t; i < 256; i += NMS_T) sh.hist[i] = 0;
        __syncthreads();
        const u64 prefix = *sh.prefix;
        for (int i = t; i < count; i += NMS_T) {
            const u64 key = keys[i];
            if (upper_valid && key >= upper) continue;
            if (d < 7 && (key >> (shift + 8)) != (prefix >> (shift + 8))) continue;
            atomicAdd(&sh.hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (t == 0) {
            int need = *sh.need, acc = 0, digit = 0;
            for (int bin = 255; bin >= 0; --bin) {
                const int hcount = (int)sh.hist[bin];
                if (acc + hcount >= need) { digit = bin; break; }
                acc += hcount;
            }
            *sh.need = need - acc;
            *sh.prefix = prefix | ((u64)digit << shift);
        }
        __syncthreads();
    }
    return *sh.prefix;
}

// Keys in [T, upper) -> s_keys[0..nb) in exact descending order (nb <= CAP).  Returns nb; s_tmp is CAP words of scratch.
__device__ __forceinline__ int gather_sorted(const u64* __restrict__ keys, int count, u64 T, bool upper_valid, u64 upper,
                                             u64* s_keys, u64* s_tmp, const SelShared& sh, int t) {
    const int lane = t & 63;
    if (t == 0) *sh.nb = 0;
    __syncthreads();
    for (int i0 = 0; i0 < count; i0 += NMS_T) {               // uniform trip count: the ballot sees whole waves
        const int i = i0 + t;
        u64 key = 0;
        bool take = false;
        if (i < count) {
            key = keys[i];
            take = key >= T && (!upper_valid || key < upper);
        }
        const u64 tm = __ballot(take);
        if (tm != 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(sh.nb, __popcll(tm));
            base = __shfl(base, 0, 64);
            const int pos = base + __popcll(tm & ((1ull << lane) - 1ull));
            if (take && pos < CAP) s_keys[pos] = key;
        }
    }
    __syncthreads();
    int nb = *sh.nb;
    nb = nb > CAP ? CAP : nb;
    if (nb <= 0) return 0;
    const int P = nb <= NMS_T ? NMS_T : (nb <= 2 * NMS_T ? 2 * NMS_T : CAP);
    u64 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int e = r * NMS_T + t;
        v[r] = e < nb ? s_keys[e] : 0ull;
    }
    __syncthreads();                                          // s_keys is one of the exchange buffers
    bitonic_regs_desc(v, P, s_keys, s_tmp, t);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int e = r * NMS_T + t;
        if (e < P) s_keys[e] = v[r];
    }
    __syncthreads();
    return nb;
}

__device__ __forceinline__ float4 candidate_box(const DevAnchorCfg& c, const float* __restrict__ regression,
                                                const float4* __restrict__ boxes_explicit, long long img_base, u64 key, float cw, float ch) {
    const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    return boxes_explicit ? boxes_explicit[img_base + idx] : decode_box(c, regression, img_base, idx, cw, ch);
}

// ---- stage 2a: first batch of every (image, class): select + sort + decode, written out for the mask kernel ----------
// meta[bk] = {nb, T low word, T high word, 0}
__global__ __launch_bounds__(NMS_T) void nms_sort_kernel(const DevAnchorCfg c, int K, const float* __restrict__ regression,
                                                         const u64* __restrict__ keys_all, const int* __restrict__ counts, float cw,
                                                         float ch, const float4* __restrict__ boxes_explicit, int n_explicit,
                                                         u64* __restrict__ sorted_keys, float4* __restrict__ sorted_boxes,
                                                         int* __restrict__ meta) {
    __shared__ u64 s_keys[CAP];
    __shared__ u64 s_tmp[CAP];
    __shared__ unsigned s_hist[256];
    __shared__ int s_nb, s_need;
    __shared__ u64 s_prefix;
    const SelShared sh{s_hist, &s_nb, &s_need, &s_prefix};
    const int t = threadIdx.x;
    const int bk = blockIdx.x;
    const int b = bk / K;
    const int N = boxes_explicit ? n_explicit : c.total;
    const long long img_base = (long long)b * N;
    const u64* keys = keys_all + (long long)bk * N;
    int count = counts[bk];
    count = count < 0 ? 0 : (count > N ? N : count);
    const u64 T = select_threshold(keys, count, count, false, 0, sh, t);
    const int nb = count > 0 ? gather_sorted(keys, count, T, false, 0, s_keys, s_tmp, sh, t) : 0;
    for (int e = t; e < nb; e += NMS_T) {
        const u64 key = s_keys[e];
        sorted_keys[(long long)bk * CAP + e] = key;
        sorted_boxes[(long long)bk * CAP + e] = candidate_box(c, regression, boxes_explicit, img_base, key, cw, ch);
    }
    if (t == 0) {
        meta[bk * 4 + 0] = nb;
        meta[bk * 4 + 1] = (int)(unsigned)(T & 0xFFFFFFFFull);
        meta[bk * 4 + 2] = (int)(unsigned)(T >> 32);
        meta[bk * 4 + 3] = 0;
    }
}

// ---- stage 2b: suppression bit matrix of the first batch, all CUs ----------------------------------------------------
// mask[bk][i][w] bit j: candidate w*64+j (later in score order than i) overlaps candidate i by more than the threshold.
// One wavefront per 64x64 tile of the upper triangle (lane = row i, the 64 column boxes broadcast with v_readlane).
constexpr int MASK_T = 256;
__global__ __launch_bounds__(MASK_T) void nms_mask_kernel(const float4* __restrict__ sorted_boxes, const int* __restrict__ meta,
                                                          float iou_thr, u64* __restrict__ mask) {
    const int bk = blockIdx.y;
    const int nb = meta[bk * 4];
    const int nc = (nb + 63) >> 6;
    const int ntiles = nc * (nc + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* B = sorted_boxes + (long long)bk * CAP;
    u64* M = mask + (long long)bk * CAP * (CAP / 64);
    for (int tile = blockIdx.x * (MASK_T / 64) + wave; tile < ntiles; tile += gridDim.x * (MASK_T / 64)) {
        int rc = 0, left = tile;                     // tile -> (row chunk rc, column chunk cc >= rc), row-major over the triangle
        while (left >= nc - rc) { left -= nc - rc; ++rc; }
        const int cc = rc + left;
        const int i = rc * 64 + lane, jx = cc * 64 + lane;
        const NBox rb = nbox(i < nb ? B[i] : make_float4(0.f, 0.f, 0.f, 0.f));
        const NBox cb = nbox(jx < nb ? B[jx] : make_float4(0.f, 0.f, 0.f, 0.f));
        u64 word = 0;
#pragma unroll 4
        for (int j = 0; j < 64; ++j) {
            NBox bj;
            bj.x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.x1), j));
            bj.y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.y1), j));
            bj.x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.x2), j));
            bj.y2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.y2), j));
            bj.area = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.area), j));
            const int col = cc * 64 + j;
            if (iou_gt_flat(rb, bj, iou_thr) && col < nb && col > i) word |= (1ull << j);
        }
        if (i < nb) M[(long long)i * (CAP / 64) + cc] = word;
    }
}

__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((u64)hi << 32) | (u64)lo;
}

// ---- stage 2c: greedy scan.  The first batch walks the bit matrix (one wavefront: lane w holds the removed-bits word w;
// a chunk of 64 candidates is resolved with scalar bit operations on its diagonal tile, then the rows of the newly kept
// candidates are OR-ed into the removed words).  Later batches (more than CAP candidates and fewer than max_det kept so
// far) run the in-workgroup path: select, sort, decode, test against the kept boxes, 64 candidates at a time.
__global__ __launch_bounds__(NMS_T) void nms_kernel(const DevAnchorCfg c, int K, const float* __restrict__ regression,
                                                    const u64* __restrict__ keys_all, const int* __restrict__ counts, float cw,
                                                    float ch, float iou_thr, int max_det, u64* __restrict__ sel_keys,
                                                    float4* __restrict__ sel_boxes, int* __restrict__ sel_count,
                                                    const float4* __restrict__ boxes_explicit, int n_explicit,
                                                    const u64* __restrict__ sorted_keys, const float4* __restrict__ sorted_boxes,
                                                    const u64* __restrict__ mask, const int* __restrict__ meta) {
    __shared__ u64 s_keys[CAP];
    __shared__ float4 s_box[CAP];
    __shared__ unsigned char s_removed[CAP];
    __shared__ float4 s_kbox[RTN_MAX_DET];          // kept boxes as decoded (output)
    __shared__ float4 s_kn[RTN_MAX_DET];            // ... normalised corners
    __shared__ float s_ka[RTN_MAX_DET];             // ... area
    __shared__ u64 s_kkey[RTN_MAX_DET];
    __shared__ unsigned s_hist[256];
    __shared__ int s_nb, s_kept, s_newbeg, s_need;
    __shared__ u64 s_prefix;
    const SelShared sh{s_hist, &s_nb, &s_need, &s_prefix};

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bk = blockIdx.x;
    const int b = bk / K;
    const int N = boxes_explicit ? n_explicit : c.total;
    const long long img_base = (long long)b * N;
    const u64* keys = keys_all + (long long)bk * N;
    int count = counts[bk];
    count = count < 0 ? 0 : (count > N ? N : count);

    // ---- first batch: scan of the bit matrix
    const int nb0 = meta[bk * 4];
    const u64 T0 = ((u64)(unsigned)meta[bk * 4 + 2] << 32) | (u64)(unsigned)meta[bk * 4 + 1];
    if (wave == 0) {
        const u64* M = mask + (long long)bk * CAP * (CAP / 64);
        const u64* SK = sorted_keys + (long long)bk * CAP;
        const float4* SB = sorted_boxes + (long long)bk * CAP;
        const int nc = (nb0 + 63) >> 6;
        u64 rem = 0;                                 // lane w: removed bits of candidates w*64 .. w*64+63
        int kept = 0;
        u64 d_n = 0, key_n = 0;
        float4 bx_n = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < nb0) { d_n = M[(long long)lane * (CAP / 64)]; key_n = SK[lane]; bx_n = SB[lane]; }
        for (int cidx = 0; cidx < nc && kept < max_det; ++cidx) {
            const u64 d = d_n, key = key_n;
            const float4 bx = bx_n;
            {   // prefetch the next chunk's diagonal words, keys and boxes: they do not depend on the scan state
                const int i = (cidx + 1) * 64 + lane;
                d_n = 0; key_n = 0;
                if (i < nb0) { d_n = M[(long long)i * (CAP / 64) + cidx + 1]; key_n = SK[i]; bx_n = SB[i]; }
            }
            const int left = nb0 - cidx * 64;
            const u64 validmask = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            u64 alive = ~readlane_u64(rem, cidx) & validmask;
            u64 keep = 0;
            while (alive != 0 && kept < max_det) {
                const int j = __ffsll((long long)alive) - 1;
                keep |= (1ull << j);
                ++kept;
                alive &= ~(readlane_u64(d, j) | (1ull << j));    // row j of the diagonal tile holds only bits above j
            }
            if (keep != 0) {
                if ((keep >> lane) & 1ull) {
                    const int pos = kept - __popcll(keep) + __popcll(keep & ((1ull << lane) - 1ull));
                    const NBox nbx = nbox(bx);
                    s_kbox[pos] = bx;
                    s_kn[pos] = make_float4(nbx.x1, nbx.y1, nbx.x2, nbx.y2);
                    s_ka[pos] = nbx.area;
                    s_kkey[pos] = key;
                }
                if (cidx + 1 < nc && kept < max_det) {           // rows of the newly kept candidates -> removed words
                    u64 km = keep, acc = 0;
                    const u64* Mc = M + (long long)cidx * 64 * (CAP / 64) + lane;
                    while (km != 0) {                            // eight independent row loads in flight (one L2 round trip per batch)
                        int jj[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            jj[q] = km != 0 ? __ffsll((long long)km) - 1 : -1;
                            if (km != 0) km &= km - 1;
                        }
                        u64 w[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) w[q] = jj[q] >= 0 ? Mc[(long long)jj[q] * (CAP / 64)] : 0ull;
                        acc |= ((w[0] | w[1]) | (w[2] | w[3])) | ((w[4] | w[5]) | (w[6] | w[7]));
                    }
                    rem |= acc;
                }
            }
        }
        if (lane == 0) { s_kept = kept; s_newbeg = kept; }
    }
    __syncthreads();

    // ---- later batches
    bool upper_valid = true;
    u64 upper = T0;
    int remaining = (T0 == 0) ? 0 : count - nb0;
    while (true) {
        const int kept_now = s_kept;
        if (kept_now >= max_det || remaining <= 0) break;
        const u64 T = select_threshold(keys, count, remaining, upper_valid, upper, sh, t);
        const int nb = gather_sorted(keys, count, T, upper_valid, upper, s_keys, reinterpret_cast<u64*>(s_box), sh, t);
        if (nb <= 0) break;
        // ---- decode boxes; drop candidates overlapping boxes kept by earlier batches.  A thread owns the rows
        // t, t+1024, ..: their normalised boxes stay in registers for the whole batch.
        NBox rb[ROWS];
        bool ralive[ROWS];
        float4 rbox[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = t + r * NMS_T;
            ralive[r] = false;
            rb[r] = NBox{0.f, 0.f, 0.f, 0.f, 0.f};
            rbox[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nb) {
                rbox[r] = candidate_box(c, regression, boxes_explicit, img_base, s_keys[i], cw, ch);
                rb[r] = nbox(rbox[r]);
                ralive[r] = true;
            }
        }
        __syncthreads();                             // gather_sorted used s_box as its exchange buffer
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = t + r * NMS_T;
            if (i < nb) s_box[i] = rbox[r];
        }
        for (int j = 0; j < kept_now; ++j) {
            const float4 kq = s_kn[j];
            const NBox kb{kq.x, kq.y, kq.z, kq.w, s_ka[j]};
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                if (ralive[r] && iou_gt(rb[r], kb, iou_thr)) ralive[r] = false;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = t + r * NMS_T;
            if (i < nb) s_removed[i] = ralive[r] ? 0 : 1;
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
                const NBox nb_ = nbox(bx);
                const int kept0 = s_kept;
                bool alive = valid;
                int keptn = kept0;
                u64 keepmask = 0;
                u64 m = __ballot(alive);
                while (m != 0 && keptn < max_det) {
                    const int j = __ffsll((long long)m) - 1;
                    keepmask |= (1ull << j);
                    ++keptn;
                    // j comes from the ballot (wave-uniform): v_readlane, not a bpermute round trip through LDS
                    NBox bj;
                    bj.x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nb_.x1), j));
                    bj.y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nb_.y1), j));
                    bj.x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nb_.x2), j));
                    bj.y2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nb_.y2), j));
                    bj.area = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nb_.area), j));
                    if (alive && lane > j && iou_gt(nb_, bj, iou_thr)) alive = false;
                    const u64 below = (j == 63) ? ~0ull : ((2ull << j) - 1ull);
                    m = __ballot(alive) & ~below;
                }
                if ((keepmask >> lane) & 1ull) {
                    const int pos = kept0 + __popcll(keepmask & ((1ull << lane) - 1ull));
                    s_kbox[pos] = bx;
                    s_kn[pos] = make_float4(nb_.x1, nb_.y1, nb_.x2, nb_.y2);
                    s_ka[pos] = nb_.area;
                    s_kkey[pos] = s_keys[i];
                }
                if (lane == 0) { s_newbeg = kept0; s_kept = keptn; }
            }
            __syncthreads();
            const int k0 = s_newbeg, k1 = s_kept;
            kept_reg = k1;
            if (k1 > k0 && ci + 1 < nchunks) {
                bool changed[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    changed[r] = false;
                    if (t + r * NMS_T < (ci + 1) * 64) ralive[r] = false;      // rows of finished chunks need no more tests
                }
                for (int j = k0; j < k1; ++j) {
                    const float4 kq = s_kn[j];
                    const NBox kb{kq.x, kq.y, kq.z, kq.w, s_ka[j]};
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
                        if (ralive[r] && iou_gt(rb[r], kb, iou_thr)) { ralive[r] = false; changed[r] = true; }
                }
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    if (changed[r]) s_removed[t + r * NMS_T] = 1;
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
    if (K > 1) bitonic_sort_desc(s_keys, P, t, NMS_T);       // one class: the NMS output already is in descending key order
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

struct WsLayout { size_t keys, counts, sel_keys, sel_boxes, sel_count, sorted_keys, sorted_boxes, mask, meta, total; };

WsLayout ws_layout(int B, long long N, int K) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WsLayout w;
    size_t o = 0;
    w.keys = o;      o += al((size_t)B * K * N * 8);
    w.counts = o;    o += al((size_t)B * K * 4 * 2);      // candidate counts then selected counts
    w.sel_count = w.counts + (size_t)B * K * 4;
    w.sel_keys = o;  o += al((size_t)B * K * RTN_MAX_DET * 8);
    w.sel_boxes = o; o += al((size_t)B * K * RTN_MAX_DET * 16);
    // first batch of every (image, class): sorted keys, decoded boxes, suppression bit matrix (CAP x CAP bits = 2 MiB)
    w.sorted_keys = o;  o += al((size_t)B * K * CAP * 8);
    w.sorted_boxes = o; o += al((size_t)B * K * CAP * 16);
    w.mask = o;         o += al((size_t)B * K * CAP * (CAP / 64) * 8);
    w.meta = o;         o += al((size_t)B * K * 16);
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
    hipLaunchKernelGGL(detect_candidates_kernel, dim3((N + CAND_PER_BLOCK - 1) / CAND_PER_BLOCK, BK), dim3(CAND_T), 0, h->stream, N, num_classes,
                       classification, score_threshold, keys, counts);
    RTN_CHECK_LAUNCH(h, "detect_candidates_kernel");
    u64* sorted_keys = (u64*)(ws + w.sorted_keys);
    float4* sorted_boxes = (float4*)(ws + w.sorted_boxes);
    u64* mask = (u64*)(ws + w.mask);
    int* meta = (int*)(ws + w.meta);
    hipLaunchKernelGGL(nms_sort_kernel, dim3(BK), dim3(NMS_T), 0, h->stream, d, num_classes, regression, (const u64*)keys,
                       (const int*)counts, (float)canvas_w, (float)canvas_h, boxes_explicit, N, sorted_keys, sorted_boxes, meta);
    RTN_CHECK_LAUNCH(h, "nms_sort_kernel");
    int mask_blocks = 1024 / BK;                     // workgroups per (image, class): ~4 per CU in total
    mask_blocks = mask_blocks < 1 ? 1 : (mask_blocks > 512 ? 512 : mask_blocks);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(mask_blocks, BK), dim3(MASK_T), 0, h->stream, (const float4*)sorted_boxes,
                       (const int*)meta, nms_threshold, mask);
    RTN_CHECK_LAUNCH(h, "nms_mask_kernel");
    hipLaunchKernelGGL(nms_kernel, dim3(BK), dim3(NMS_T), 0, h->stream, d, num_classes, regression, (const u64*)keys,
                       (const int*)counts, (float)canvas_w, (float)canvas_h, nms_threshold, max_detections, sel_keys, sel_boxes,
                       sel_count, boxes_explicit, N, (const u64*)sorted_keys, (const float4*)sorted_boxes, (const u64*)mask,
                       (const int*)meta);
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
