// rtn_loss.hip — focal + smooth-L1 forward sums and backward, fused over the ~200k anchors.
//
// Restates model/losses.py:13-44 (focal, alpha 0.25 gamma 2) and :58-90 (smooth_l1, sigma 3):
//   focal   : rows with anchor state -1 are dropped; alpha_t = alpha for label 1 else 1-alpha;
//             weight = alpha_t * (1-p)^gamma for label 1 else alpha_t * p^gamma;
//             times K.binary_crossentropy(label, p) on probabilities clipped to [1e-7, 1-1e-7];
//             summed, divided by max(1, #rows with state 1).
//   smoothL1: rows with state 1 only; d = |pred - target|; d < 1/sigma^2 -> 0.5 sigma^2 d^2
//             else d - 0.5/sigma^2; summed, divided by max(1, #positive rows).
// HBM-bound: one pass over (K+1) + 5 + K + 4 floats per anchor; per-thread f64 partials,
// wave shuffle reduction, fixed-order second stage => bitwise reproducible run to run.
// The division by the normaliser is the caller's (it is all-reduced under data parallelism).
#include "rtn_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int LOSS_BLOCK = 256;
constexpr int LOSS_MAX_BLOCKS = 2048;
constexpr float BCE_EPS = 1e-7f;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ __launch_bounds__(LOSS_BLOCK) void loss_fwd_kernel(long long rows, int K, const float* __restrict__ lab,
                                                               const float* __restrict__ regt, const float* __restrict__ cls,
                                                               const float* __restrict__ reg, float alpha, float gamma,
                                                               float sigma2, double* __restrict__ partial) {
    double s_cls = 0.0, s_reg = 0.0, n_pos = 0.0, n_pos_reg = 0.0;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const float* l = lab + r * (K + 1);
        const float state = l[K];
        if (state != -1.f) {
            for (int k = 0; k < K; ++k) {
                const float y = l[k];
                const float p = cls[r * K + k];
                const bool one = (y == 1.f);
                const float af = one ? alpha : 1.f - alpha;
                const float fw0 = one ? 1.f - p : p;
                const float fw = af * (gamma == 2.f ? fw0 * fw0 : powf(fw0, gamma));
                const float pc = fminf(fmaxf(p, BCE_EPS), 1.f - BCE_EPS);
                const float bce = -(y * logf(pc) + (1.f - y) * logf(1.f - pc));
                s_cls += (double)(fw * bce);
            }
        }
        if (state == 1.f) n_pos += 1.0;
        const float4 t4 = *reinterpret_cast<const float4*>(reg + r * 4);   // prediction
        const float* tg = regt + r * 5;
        if (tg[4] == 1.f) {
            n_pos_reg += 1.0;
            const float pr[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = fabsf(pr[j] - tg[j]);
                const float v = d < 1.0f / sigma2 ? 0.5f * sigma2 * (d * d) : d - 0.5f / sigma2;
                s_reg += (double)v;
            }
        }
    }
    __shared__ double sh[4][LOSS_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_cls = wave_sum(s_cls); s_reg = wave_sum(s_reg); n_pos = wave_sum(n_pos); n_pos_reg = wave_sum(n_pos_reg);
    if (lane == 0) { sh[0][wave] = s_cls; sh[1][wave] = s_reg; sh[2][wave] = n_pos; sh[3][wave] = n_pos_reg; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0.0;
        for (int w = 0; w < LOSS_BLOCK / 64; ++w) v += sh[threadIdx.x][w];
        partial[(long long)blockIdx.x * 4 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void loss_final_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ sums) {
    // fixed-order tree over the block partials: thread j owns partials j, j+256, ...
    __shared__ double sh[4][256];
    double v[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblocks; i += 256)
        for (int q = 0; q < 4; ++q) v[q] += partial[(long long)i * 4 + q];
    for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] = v[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) sums[threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(long long rows, int K, const float* __restrict__ lab,
                                                       const float* __restrict__ regt, const float* __restrict__ cls,
                                                       const float* __restrict__ reg, float alpha, float gamma, float sigma2,
                                                       float inv_cls, float inv_reg, int wrt_logits, float* __restrict__ d_cls,
                                                       float* __restrict__ d_reg, const double* __restrict__ sums) {
    if (sums) {     // normalisers from the (all-reduced) forward sums: K.maximum(1, count), model/losses.py:42,88
        inv_cls = 1.0f / fmaxf(1.0f, (float)sums[2]);
        inv_reg = 1.0f / fmaxf(1.0f, (float)sums[3]);
    }
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const float* l = lab + r * (K + 1);
        const float state = l[K];
        for (int k = 0; k < K; ++k) {
            float g = 0.f;
            if (state != -1.f) {
                const float y = l[k];
                const float p = cls[r * K + k];
                const bool one = (y == 1.f);
                const float pc = fminf(fmaxf(p, BCE_EPS), 1.f - BCE_EPS);
                const bool inside = (p >= BCE_EPS) && (p <= 1.f - BCE_EPS);   // clip passes gradient only inside
                // L = af * w(p) * bce(pc):  dL/dp = af * (w'(p) * bce + w(p) * dbce/dp)
                if (one) {
                    const float q = 1.f - p;
                    const float wv = (gamma == 2.f) ? q * q : powf(q, gamma);
                    const float dw = (gamma == 2.f) ? -2.f * q : -gamma * powf(q, gamma - 1.f);
                    const float bce = -logf(pc);
                    const float dbce = inside ? -1.f / pc : 0.f;
                    g = alpha * (dw * bce + wv * dbce);
                } else {
                    const float wv = (gamma == 2.f) ? p * p : powf(p, gamma);
                    const float dw = (gamma == 2.f) ? 2.f * p : gamma * powf(p, gamma - 1.f);
                    // general y in [0,1): bce = -(y log pc + (1-y) log(1-pc))
                    const float bce = -(y * logf(pc) + (1.f - y) * logf(1.f - pc));
                    const float dbce = inside ? -(y / pc) + (1.f - y) / (1.f - pc) : 0.f;
                    g = (1.f - alpha) * (dw * bce + wv * dbce);
                }
                g *= inv_cls;
                if (wrt_logits) g *= p * (1.f - p);
            }
            d_cls[r * K + k] = g;
        }
        const float* tg = regt + r * 5;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tg[4] == 1.f) {
            const float4 p4 = *reinterpret_cast<const float4*>(reg + r * 4);
            const float pr[4] = {p4.x, p4.y, p4.z, p4.w};
            float gg[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float df = pr[j] - tg[j];
                const float d = fabsf(df);
                const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                gg[j] = (d < 1.0f / sigma2 ? sigma2 * df : sgn) * inv_reg;
            }
            o = make_float4(gg[0], gg[1], gg[2], gg[3]);
        }
        *reinterpret_cast<float4*>(d_reg + r * 4) = o;
    }
}

inline int loss_blocks(long long rows) {
    long long g = (rows + LOSS_BLOCK - 1) / LOSS_BLOCK;
    if (g < 1) g = 1;
    if (g > LOSS_MAX_BLOCKS) g = LOSS_MAX_BLOCKS;
    return (int)g;
}

}  // namespace

extern "C" size_t rtn_retina_loss_workspace_bytes(int64_t rows) { return (size_t)loss_blocks(rows) * 4 * sizeof(double); }

extern "C" int rtn_retina_loss_fwd(rtn_handle_t h, int64_t rows, int num_classes, const float* labels_batch,
                                   const float* regression_batch, const float* classification, const float* regression,
                                   float alpha, float gamma, float sigma, double* sums, void* workspace, size_t workspace_bytes) {
    if (!h) return RTN_EINVAL;
    if (rows < 1 || num_classes < 1) return rtn_fail(h, RTN_EINVAL, "loss_fwd: rows %lld classes %d", (long long)rows, num_classes);
    if (!labels_batch || !regression_batch || !classification || !regression || !sums || !workspace)
        return rtn_fail(h, RTN_EINVAL, "loss_fwd: null pointer");
    if ((uintptr_t)regression & 15) return rtn_fail(h, RTN_EINVAL, "loss_fwd: regression not 16-byte aligned");
    if (workspace_bytes < rtn_retina_loss_workspace_bytes(rows)) return rtn_fail(h, RTN_ENOMEM, "loss_fwd: workspace %zu < %zu", workspace_bytes, rtn_retina_loss_workspace_bytes(rows));
    const int nb = loss_blocks(rows);
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(nb), dim3(LOSS_BLOCK), 0, h->stream, (long long)rows, num_classes, labels_batch,
                       regression_batch, classification, regression, alpha, gamma, sigma * sigma, (double*)workspace);
    RTN_CHECK_LAUNCH(h, "loss_fwd_kernel");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, h->stream, (const double*)workspace, nb, sums);
    RTN_CHECK_LAUNCH(h, "loss_final_kernel");
    return RTN_OK;
}

extern "C" int rtn_retina_loss_bwd(rtn_handle_t h, int64_t rows, int num_classes, const float* labels_batch,
                                   const float* regression_batch, const float* classification, const float* regression,
                                   float alpha, float gamma, float sigma, float inv_norm_cls, float inv_norm_reg, int wrt_logits,
                                   float* d_cls, float* d_reg) {
    if (!h) return RTN_EINVAL;
    if (rows < 1 || num_classes < 1) return rtn_fail(h, RTN_EINVAL, "loss_bwd: rows %lld classes %d", (long long)rows, num_classes);
    if (!labels_batch || !regression_batch || !classification || !regression || !d_cls || !d_reg)
        return rtn_fail(h, RTN_EINVAL, "loss_bwd: null pointer");
    if (((uintptr_t)regression & 15) || ((uintptr_t)d_reg & 15)) return rtn_fail(h, RTN_EINVAL, "loss_bwd: regression buffers not 16-byte aligned");
    const int nb = loss_blocks(rows);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(nb), dim3(256), 0, h->stream, (long long)rows, num_classes, labels_batch,
                       regression_batch, classification, regression, alpha, gamma, sigma * sigma, inv_norm_cls, inv_norm_reg,
                       wrt_logits, d_cls, d_reg, (const double*)nullptr);
    RTN_CHECK_LAUNCH(h, "loss_bwd_kernel");
    return RTN_OK;
}

extern "C" int rtn_retina_loss_bwd_dev(rtn_handle_t h, int64_t rows, int num_classes, const float* labels_batch,
                                       const float* regression_batch, const float* classification, const float* regression,
                                       float alpha, float gamma, float sigma, const double* sums, int wrt_logits, float* d_cls,
                                       float* d_reg) {
    if (!h) return RTN_EINVAL;
    if (rows < 1 || num_classes < 1) return rtn_fail(h, RTN_EINVAL, "loss_bwd_dev: rows %lld classes %d", (long long)rows, num_classes);
    if (!labels_batch || !regression_batch || !classification || !regression || !d_cls || !d_reg || !sums)
        return rtn_fail(h, RTN_EINVAL, "loss_bwd_dev: null pointer");
    if (((uintptr_t)regression & 15) || ((uintptr_t)d_reg & 15)) return rtn_fail(h, RTN_EINVAL, "loss_bwd_dev: regression buffers not 16-byte aligned");
    const int nb = loss_blocks(rows);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(nb), dim3(256), 0, h->stream, (long long)rows, num_classes, labels_batch,
                       regression_batch, classification, regression, alpha, gamma, sigma * sigma, 0.f, 0.f, wrt_logits, d_cls,
                       d_reg, sums);
    RTN_CHECK_LAUNCH(h, "loss_bwd_kernel");
    return RTN_OK;
}
