// rtn_preprocess.hip — page preprocessing on the device:
//   DetectTablesUtils.py:251-261  cvtColor(BGR2GRAY) -> adaptiveThreshold(255, GAUSSIAN_C, BINARY, 11, 2) ->
//                                 distanceTransform(L2 mask 5 | L1 | C) -> merge(b,g,r) -> imwrite (saturate to uint8)
//   model/utils.py:43-46,140-154  x/127.5 - 1, then cv2.resize(fx=fy=scale, INTER_CUBIC), written into the zero-padded batch
//                                 canvas of Generator.compute_inputs (csv_generator.py:320-336).
// OpenCV's algorithms are restated (opencv is neither vendored nor pinned by the reference; oracle/ref_preprocess.py is the
// CPU statement of the same and is checked against the reference's sample page pair).
//
// Distance transform = OpenCV's two-pass chamfer in 16.16 fixed point, kept EXACT and made parallel: within one raster row the
// recurrence D[x] = min(t[x], D[x-1] + a) is a prefix minimum of t[x] - a*x, so a 1024-thread workgroup sweeps the rows of one
// (page, metric) in order — previous rows live in an LDS ring — and scans each row in parallel.  24 workgroups serve a batch
// of 8 pages x 3 metrics concurrently.  FP contraction is off: the Gaussian mean and the bicubic taps round like the oracle.
#include "rtn_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int DT_MAXW = 4096;
// threads per row sweep (DT_T) x columns per thread (DT_E) are template parameters: the launcher picks the smallest
// workgroup that covers the page width (fewer waves = cheaper barriers and a shorter cross-wave prefix on the critical path)
constexpr int DT_INF = (0x7fffffff >> 2);
constexpr int DT_SHIFT = 16;

__constant__ float c_gauss11[11];

// pass A: gray (fixed point, OpenCV BGR2GRAY) + horizontal 11-tap Gaussian, replicate border
__global__ __launch_bounds__(256) void gray_hblur_kernel(const unsigned char* __restrict__ src, int channels, int B, int H, int W,
                                                         unsigned char* __restrict__ gray, float* __restrict__ tmp) {
    const long long total = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long row = i / W;                      // b*H + y
        float acc = 0.f;
        unsigned char centre = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            int xx = x + k - 5;
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const unsigned char* p = src + (row * W + xx) * channels;
            const int g = channels == 3 ? ((1868 * p[0] + 9617 * p[1] + 4899 * p[2] + 8192) >> 14) : p[0];
            if (k == 5) centre = (unsigned char)g;
            acc = acc + (float)g * c_gauss11[k];
        }
        gray[i] = centre;
        tmp[i] = acc;
    }
}

// pass B: vertical Gaussian, round half to even, threshold: 255 if gray - mean > -2 else 0
__global__ __launch_bounds__(256) void vblur_threshold_kernel(const unsigned char* __restrict__ gray, const float* __restrict__ tmp, int B,
                                                              int H, int W, int delta, unsigned char* __restrict__ binary) {
    const long long total = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long r = i / W;
        const int y = (int)(r % H);
        const long long b = r / H;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            int yy = y + k - 5;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            acc = acc + tmp[(b * H + yy) * W + x] * c_gauss11[k];
        }
        float m = rintf(acc);
        m = m < 0.f ? 0.f : (m > 255.f ? 255.f : m);
        binary[i] = ((int)gray[i] - (int)m > -delta) ? 255 : 0;
    }
}

__device__ __forceinline__ int wave_scan_min(int v, int lane) {      // inclusive prefix min over the 64 lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(v, o, 64);
        if (lane >= o) v = v < n ? v : n;
    }
    return v;
}

// One workgroup = one (page, metric).  dir = +1 forward raster pass, -1 backward pass (mirror image of the same code).
// ring[3][W]: rows y-2, y-1 (forward) or y+2, y+1 (backward) and the row being written.
template <bool BACKWARD, int DT_T, int DT_E>
__device__ void dt_pass(const unsigned char* __restrict__ bin, int* __restrict__ D, int H, int W, int a, int b, int c,
                        int* ring, int* wagg, unsigned char* __restrict__ out, int ch) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int x0 = t * DT_E;
    // ring rows start as "infinity"
    for (int i = t; i < 3 * DT_MAXW; i += DT_T) ring[i] = DT_INF;
    // The only global reads of a row step (the binary pixel, or the forward distance in the backward pass) do not depend on
    // the recurrence: they are requested one row ahead, so their latency is off the 2*H-step critical path.
    int pre[DT_E];
#pragma unroll
    for (int e = 0; e < DT_E; ++e) {
        const int x = x0 + e, y = BACKWARD ? H - 1 : 0;
        pre[e] = x < W ? (BACKWARD ? D[(long long)y * W + x] : (int)bin[(long long)y * W + x]) : 0;
    }
    __syncthreads();
    for (int step = 0; step < H; ++step) {
        const int y = BACKWARD ? H - 1 - step : step;
        int now[DT_E];
#pragma unroll
        for (int e = 0; e < DT_E; ++e) now[e] = pre[e];
        if (step + 1 < H) {
            const int yn = BACKWARD ? y - 1 : y + 1;
#pragma unroll
            for (int e = 0; e < DT_E; ++e) {
                const int x = x0 + e;
                pre[e] = x < W ? (BACKWARD ? D[(long long)yn * W + x] : (int)bin[(long long)yn * W + x]) : 0;
            }
        }
        const int* p1 = ring + ((step + 2) % 3) * DT_MAXW;      // previous row in sweep order
        const int* p2 = ring + ((step + 1) % 3) * DT_MAXW;      // the one before
        int* cur = ring + (step % 3) * DT_MAXW;
        int v[DT_E];
        // ---- candidates from the two previous rows (+ the forward value of this pixel in the backward pass)
#pragma unroll
        for (int e = 0; e < DT_E; ++e) {
            const int x = x0 + e;
            int tv = DT_INF;
            if (x < W) {
                auto at = [&](const int* r, int xx) { return (xx >= 0 && xx < W) ? r[xx] : DT_INF; };
                tv = min(min(at(p1, x - 1) + b, p1[x] + a), at(p1, x + 1) + b);
                if (c > 0) {
                    tv = min(tv, min(at(p2, x - 1), at(p2, x + 1)) + c);
                    tv = min(tv, min(at(p1, x - 2), at(p1, x + 2)) + c);
                }
                if (BACKWARD) tv = min(tv, now[e]);
                else if (now[e] == 0) tv = 0;
                tv = min(tv, DT_INF);
            }
            // min-plus scan along the sweep direction: forward uses t - a*x, backward t + a*x scanned from the right
            v[e] = BACKWARD ? tv + a * x : tv - a * x;
        }
        // ---- in-thread scan, then workgroup scan of the thread aggregates
        if (!BACKWARD) {
#pragma unroll
            for (int e = 1; e < DT_E; ++e) v[e] = min(v[e], v[e - 1]);
        } else {
#pragma unroll
            for (int e = DT_E - 2; e >= 0; --e) v[e] = min(v[e], v[e + 1]);
        }
        // thread order along the sweep: forward t ascending, backward t descending -> scan over rt
        const int agg = BACKWARD ? v[0] : v[DT_E - 1];
        // mirror lanes for the backward pass so that one inclusive scan routine serves both
        int s_in = agg;
        if (BACKWARD) s_in = __shfl(agg, 63 - lane, 64);
        int incl = wave_scan_min(s_in, lane);
        const int wtot = __shfl(incl, 63, 64);
        int excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0x7fffffff;
        if (lane == 0) wagg[BACKWARD ? (DT_T / 64 - 1 - wave) : wave] = wtot;
        __syncthreads();
        // prefix over waves (16 values): every thread folds the waves before its own
        int carry = 0x7fffffff;
        const int wpos = BACKWARD ? (DT_T / 64 - 1 - wave) : wave;
        for (int w = 0; w < wpos; ++w) carry = min(carry, wagg[w]);
        excl = min(excl, carry);
        if (BACKWARD) excl = __shfl(excl, 63 - lane, 64);          // un-mirror: lane L gets the prefix of everything to its right
#pragma unroll
        for (int e = 0; e < DT_E; ++e) {
            const int x = x0 + e;
            if (x < W) {
                const int s = min(v[e], excl);
                const int d = BACKWARD ? s - a * x : s + a * x;
                cur[x] = d;
                if (BACKWARD) {
                    // distanceTransform result (float), then imwrite's saturate_cast<uchar>
                    float f = rintf((float)d * (1.0f / (float)(1 << DT_SHIFT)));
                    f = f > 255.f ? 255.f : f;
                    out[((long long)y * W + x) * 3 + ch] = (unsigned char)f;
                } else {
                    D[(long long)y * W + x] = d;
                }
            }
        }
        __syncthreads();
    }
}

// grid = B * 3 ; metric 0: DIST_L2 mask 5 -> channel 0 (b), 1: DIST_L1 -> channel 1 (g), 2: DIST_C -> channel 2 (r)
template <int DT_T, int DT_E>
__global__ __launch_bounds__(DT_T) void dt3_kernel(const unsigned char* __restrict__ binary, int* __restrict__ scratch, int H, int W,
                                                   unsigned char* __restrict__ out) {
    __shared__ int ring[3 * DT_MAXW];
    __shared__ int wagg[DT_T / 64];
    const int img = blockIdx.x / 3, metric = blockIdx.x % 3;
    const int one = 1 << DT_SHIFT;
    const int a = one;
    const int b = metric == 0 ? 91750 /* round(1.4 * 65536) */ : (metric == 1 ? 2 * one : one);
    const int c = metric == 0 ? 143976 /* round(2.1969 * 65536) */ : 0;
    const unsigned char* bin = binary + (long long)img * H * W;
    int* D = scratch + ((long long)img * 3 + metric) * H * W;
    unsigned char* o = out + (long long)img * H * W * 3;
    dt_pass<false, DT_T, DT_E>(bin, D, H, W, a, b, c, ring, wagg, o, metric);
    __syncthreads();
    dt_pass<true, DT_T, DT_E>(bin, D, H, W, a, b, c, ring, wagg, o, metric);
}

// cv2.resize(INTER_CUBIC): A = -0.75, src = (dst + 0.5)/scale - 0.5, replicate border; optional fused x/127.5 - 1 for uint8 input
__device__ __forceinline__ void cubic_coeffs(float x, float (&k)[4]) {
    const float A = -0.75f;
    k[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    k[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    k[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
    k[3] = 1.f - k[0] - k[1] - k[2];
}

template <int SRC_U8, int DST_BF16>
__global__ __launch_bounds__(256) void resize_cubic_kernel(const void* __restrict__ src, int H, int W, int C, double inv_scale,
                                                           void* __restrict__ dst, int Ho, int Wo, long long dst_row_stride) {
    const long long total = (long long)Ho * Wo * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % C);
        const long long r = i / C;
        const int dx = (int)(r % Wo), dy = (int)(r / Wo);
        const double fx = ((double)dx + 0.5) * inv_scale - 0.5, fy = ((double)dy + 0.5) * inv_scale - 0.5;
        const int sx = (int)floor(fx), sy = (int)floor(fy);
        float kx[4], ky[4];
        cubic_coeffs((float)(fx - (double)sx), kx);
        cubic_coeffs((float)(fy - (double)sy), ky);
        float rows[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int yy = sy - 1 + j;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int xx = sx - 1 + k;
                xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                const long long si = ((long long)yy * W + xx) * C + ch;
                float v;
                if (SRC_U8) v = __fsub_rn(__fdiv_rn((float)((const unsigned char*)src)[si], 127.5f), 1.0f);
                else v = ((const float*)src)[si];
                acc = acc + v * kx[k];
            }
            rows[j] = acc;
        }
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o = o + rows[j] * ky[j];
        const long long di = (long long)dy * dst_row_stride + (long long)dx * C + ch;
        if (DST_BF16) {
            const __bf16 hb = (__bf16)o;
            ((unsigned short*)dst)[di] = __builtin_bit_cast(unsigned short, hb);
        } else {
            ((float*)dst)[di] = o;
        }
    }
}

// The generator's case (uint8 BGR page, C = 3): one thread per OUTPUT PIXEL instead of per element, so the source coordinate and
// the eight cubic weights are computed once for the three channels; x/127.5 - 1 comes from a 256-entry LDS table filled with the
// very same two roundings; an interior pixel fetches each of its four source rows as three (unaligned) dwords = 4 px * 3 ch.
// Per channel the sums run in the order of resize_cubic_kernel: bit-identical results.
template <int DST_BF16>
__global__ __launch_bounds__(256) void resize_cubic_u8c3_kernel(const unsigned char* __restrict__ src, int H, int W, double inv_scale,
                                                                void* __restrict__ dst, int Ho, int Wo, long long dst_row_stride) {
    __shared__ float lut[256];
    lut[threadIdx.x] = __fsub_rn(__fdiv_rn((float)threadIdx.x, 127.5f), 1.0f);
    __syncthreads();
    const long long total = (long long)Ho * Wo;
    const long long src_bytes = (long long)H * W * 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int dx = (int)(i % Wo), dy = (int)(i / Wo);
        const double fx = ((double)dx + 0.5) * inv_scale - 0.5, fy = ((double)dy + 0.5) * inv_scale - 0.5;
        const int sx = (int)floor(fx), sy = (int)floor(fy);
        float kx[4], ky[4];
        cubic_coeffs((float)(fx - (double)sx), kx);
        cubic_coeffs((float)(fy - (double)sy), ky);
        const bool inside_x = sx >= 1 && sx + 2 < W;
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int yy = sy - 1 + j;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const long long rowb = (long long)yy * W * 3;
            unsigned char t[12];
            const long long at = rowb + (long long)(sx - 1) * 3;
            if (inside_x && at + 12 <= src_bytes) {
                unsigned int w3[3];
                __builtin_memcpy(w3, src + at, 12);
                __builtin_memcpy(t, w3, 12);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int xx = sx - 1 + k;
                    xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
#pragma unroll
                    for (int c = 0; c < 3; ++c) t[k * 3 + c] = src[rowb + (long long)xx * 3 + c];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = acc + lut[t[k * 3 + c]] * kx[k];
                o[c] = o[c] + acc * ky[j];
            }
        }
        const long long di = (long long)dy * dst_row_stride + (long long)dx * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (DST_BF16) {
                const __bf16 hb = (__bf16)o[c];
                ((unsigned short*)dst)[di + c] = __builtin_bit_cast(unsigned short, hb);
            } else {
                ((float*)dst)[di + c] = o[c];
            }
        }
    }
}

inline unsigned grid_for(long long work, int cap = 4096) {
    long long g = (work + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

bool g_gauss_uploaded[64] = {false};

}  // namespace

static void dt3_launch(rtn_handle_t h, const unsigned char* binary, int* scratch, int B, int H, int W, unsigned char* dst) {
    // Measured on 2200x1712 pages (RTN_DT_CFG="T,E" is the A/B knob): the per-thread column loop, not the cross-wave prefix,
    // is the critical path of a row step: 1024 threads x 2 columns 7.2 ms, x 4 columns 8.5, 512 x 8 12.4, 256 x 16 19.7 (8 pages).
    // So: as many threads as columns allow, the fewest columns per thread.
    int T = W <= 256 ? 256 : (W <= 512 ? 512 : 1024);
    int E = (W + T - 1) / T;
    if (const char* e = getenv("RTN_DT_CFG")) { int t_ = 0, e_ = 0; if (sscanf(e, "%d,%d", &t_, &e_) == 2 && (long long)t_ * e_ >= W) { T = t_; E = e_; } }
    const dim3 g(B * 3);
#define DT_GO(TT, EE) if (T == TT && E == EE) { hipLaunchKernelGGL((dt3_kernel<TT, EE>), g, dim3(TT), 0, h->stream, binary, scratch, H, W, dst); return; }
    DT_GO(256, 1) DT_GO(512, 1) DT_GO(1024, 1) DT_GO(1024, 2) DT_GO(1024, 3) DT_GO(1024, 4) DT_GO(512, 4) DT_GO(512, 8) DT_GO(256, 8) DT_GO(256, 16)
#undef DT_GO
    hipLaunchKernelGGL((dt3_kernel<1024, 4>), g, dim3(1024), 0, h->stream, binary, scratch, H, W, dst);
}

extern "C" size_t rtn_preprocess_dt3_workspace_bytes(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    const size_t px = (size_t)B * H * W;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return al(px) /* gray */ + al(px * 4) /* row-blurred f32 */ + al(px) /* binary */ + al(px * 3 * 4) /* forward distances */;
}

extern "C" int rtn_preprocess_dt3(rtn_handle_t h, const uint8_t* src, int channels, int B, int H, int W, uint8_t* dst,
                                  uint8_t* binary_out, void* workspace, size_t workspace_bytes) {
    if (!h) return RTN_EINVAL;
    if (!src || !dst || !workspace || B < 1 || H < 1 || W < 1 || (channels != 1 && channels != 3)) return rtn_fail(h, RTN_EINVAL, "preprocess_dt3: bad argument");
    if (W > DT_MAXW) return rtn_fail(h, RTN_EINVAL, "preprocess_dt3: pages wider than %d px are not supported (got %d)", DT_MAXW, W);
    if ((long long)B * 3 > 65535) return rtn_fail(h, RTN_EINVAL, "preprocess_dt3: batch too large");
    if (workspace_bytes < rtn_preprocess_dt3_workspace_bytes(B, H, W)) return rtn_fail(h, RTN_ENOMEM, "preprocess_dt3: workspace %zu < %zu", workspace_bytes, rtn_preprocess_dt3_workspace_bytes(B, H, W));
    if ((uintptr_t)workspace & 255) return rtn_fail(h, RTN_EINVAL, "preprocess_dt3: workspace must be 256-byte aligned");
    if (h->device >= 0 && h->device < 64 && !g_gauss_uploaded[h->device]) {
        // getGaussianKernel(11, sigma <= 0): sigma = 0.3*((n-1)*0.5 - 1) + 0.8 = 2.0, double math, stored as float
        double k[11], sum = 0.0;
        const double sigma = ((11 - 1) * 0.5 - 1) * 0.3 + 0.8;
        for (int i = 0; i < 11; ++i) { const double x = i - 5.0; k[i] = exp(-0.5 / (sigma * sigma) * x * x); sum += k[i]; }
        float kf[11];
        for (int i = 0; i < 11; ++i) kf[i] = (float)(k[i] / sum);
        RTN_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(c_gauss11), kf, sizeof(kf)));
        g_gauss_uploaded[h->device] = true;
    }
    const size_t px = (size_t)B * H * W;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* ws = (char*)workspace;
    unsigned char* gray = (unsigned char*)ws;
    float* tmp = (float*)(ws + al(px));
    unsigned char* binary = (unsigned char*)(ws + al(px) + al(px * 4));
    int* scratch = (int*)(ws + al(px) + al(px * 4) + al(px));
    hipLaunchKernelGGL(gray_hblur_kernel, dim3(grid_for((long long)px)), dim3(256), 0, h->stream, src, channels, B, H, W, gray, tmp);
    RTN_CHECK_LAUNCH(h, "gray_hblur_kernel");
    hipLaunchKernelGGL(vblur_threshold_kernel, dim3(grid_for((long long)px)), dim3(256), 0, h->stream, (const unsigned char*)gray, (const float*)tmp, B, H, W, 2, binary);
    RTN_CHECK_LAUNCH(h, "vblur_threshold_kernel");
    if (binary_out) RTN_HIP(h, hipMemcpyAsync(binary_out, binary, px, hipMemcpyDeviceToDevice, h->stream));
    dt3_launch(h, (const unsigned char*)binary, scratch, B, H, W, dst);
    RTN_CHECK_LAUNCH(h, "dt3_kernel");
    return RTN_OK;
}

/* distance transforms of a caller-provided binary image (the distanceTransform x3 + merge + saturate part alone) */
extern "C" int rtn_distance_transform3(rtn_handle_t h, const uint8_t* binary, int B, int H, int W, uint8_t* dst, void* workspace,
                                       size_t workspace_bytes) {
    if (!h) return RTN_EINVAL;
    if (!binary || !dst || !workspace || B < 1 || H < 1 || W < 1) return rtn_fail(h, RTN_EINVAL, "distance_transform3: bad argument");
    if (W > DT_MAXW || (long long)B * 3 > 65535) return rtn_fail(h, RTN_EINVAL, "distance_transform3: width %d / batch %d unsupported", W, B);
    if (workspace_bytes < (size_t)B * H * W * 12) return rtn_fail(h, RTN_ENOMEM, "distance_transform3: workspace too small");
    dt3_launch(h, binary, (int*)workspace, B, H, W, dst);
    RTN_CHECK_LAUNCH(h, "dt3_kernel");
    return RTN_OK;
}

extern "C" int rtn_resize_cubic(rtn_handle_t h, const void* src, int src_dtype, int H, int W, int C, double scale, void* dst, int dst_dtype,
                                int Ho, int Wo, int64_t dst_row_stride) {
    if (!h) return RTN_EINVAL;
    if (!src || !dst || H < 1 || W < 1 || C < 1 || !(scale > 0.0)) return rtn_fail(h, RTN_EINVAL, "resize_cubic: bad argument");
    if (src_dtype != RTN_F32 && src_dtype != 2) return rtn_fail(h, RTN_EINVAL, "resize_cubic: src must be f32 or u8");
    if (dst_dtype != RTN_F32 && dst_dtype != RTN_BF16) return rtn_fail(h, RTN_EINVAL, "resize_cubic: dst must be f32 or bf16");
    // cv2.resize: dsize = saturate_cast<int>(ssize * scale) (round half to even)
    if (Ho != (int)nearbyint(H * scale) || Wo != (int)nearbyint(W * scale)) return rtn_fail(h, RTN_EINVAL, "resize_cubic: output %dx%d is not round(%dx%d * %.6f)", Ho, Wo, H, W, scale);
    if (dst_row_stride < (int64_t)Wo * C) return rtn_fail(h, RTN_EINVAL, "resize_cubic: dst_row_stride too small");
    const long long total = (long long)Ho * Wo * C;
    const double inv = 1.0 / scale;
    if (src_dtype == 2 && C == 3) {
        dim3 g3(grid_for((long long)Ho * Wo, 16384)), b3(256);
        if (dst_dtype == RTN_BF16)
            hipLaunchKernelGGL((resize_cubic_u8c3_kernel<1>), g3, b3, 0, h->stream, (const unsigned char*)src, H, W, inv, dst, Ho, Wo, (long long)dst_row_stride);
        else
            hipLaunchKernelGGL((resize_cubic_u8c3_kernel<0>), g3, b3, 0, h->stream, (const unsigned char*)src, H, W, inv, dst, Ho, Wo, (long long)dst_row_stride);
        RTN_CHECK_LAUNCH(h, "resize_cubic_u8c3_kernel");
        return RTN_OK;
    }
    dim3 g(grid_for(total, 8192)), b(256);
#define RS(S, D) hipLaunchKernelGGL((resize_cubic_kernel<S, D>), g, b, 0, h->stream, src, H, W, C, inv, dst, Ho, Wo, (long long)dst_row_stride)
    if (src_dtype == 2) { if (dst_dtype == RTN_BF16) RS(1, 1); else RS(1, 0); }
    else                { if (dst_dtype == RTN_BF16) RS(0, 1); else RS(0, 0); }
#undef RS
    RTN_CHECK_LAUNCH(h, "resize_cubic_kernel");
    return RTN_OK;
}

// ---- transform.apply_transform (model/transform.py:343-362): cv2.warpAffine of a uint8 HxWxC page ---------------------------
// The host passes the INVERSE map (destination -> source), inverted in double the way warpAffine does when WARP_INVERSE_MAP is
// not set.  Coordinates are OpenCV's fixed point: AB_BITS = 10 fractional bits for the products, INTER_BITS = 5 for the sampling
// position (1/32 pixel); the bilinear weights are the 15-bit table entries 32*(32-a)*(32-b) (short-saturated), the result
// (sum + 2^14) >> 15.  Integer work from the two products on: bit-exact against oracle/ref_generator.py.
struct WarpParams {
    double m[6];
    int H, W, C, interp, border, dword_rows;
    unsigned char cval[4];
};

__device__ __forceinline__ int border_index(int p, int len, int mode) {
    if ((unsigned)p < (unsigned)len) return p;
    if (mode == 1) return p < 0 ? 0 : len - 1;                       // BORDER_REPLICATE
    if (mode == 2) {                                                 // BORDER_REFLECT_101
        if (len == 1) return 0;
        do { p = p < 0 ? -p : 2 * len - 2 - p; } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (mode == 3) {                                                 // BORDER_WRAP
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1;                                                       // BORDER_CONSTANT
}

__global__ __launch_bounds__(256) void warp_affine_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, WarpParams p) {
    __shared__ unsigned int tile[4][64];                  // one destination row segment per wave: 64 px * C bytes
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lx, y = blockIdx.y * 4 + ly;
    const bool live = x < p.W && y < p.H;
    unsigned char o[4] = {0, 0, 0, 0};
    if (live) {
        const int round_delta = p.interp ? 16 : 512;
        const int X0 = __double2int_rn((p.m[1] * (double)y + p.m[2]) * 1024.0) + round_delta;
        const int Y0 = __double2int_rn((p.m[4] * (double)y + p.m[5]) * 1024.0) + round_delta;
        const int ax = __double2int_rn(p.m[0] * (double)x * 1024.0), bx = __double2int_rn(p.m[3] * (double)x * 1024.0);
        const int sh = p.interp ? 5 : 10;
        const int X = (X0 + ax) >> sh, Y = (Y0 + bx) >> sh;
        int sx, sy, fa = 0, fb = 0;
        if (p.interp) { sx = X >> 5; sy = Y >> 5; fa = X & 31; fb = Y & 31; }
        else { sx = X; sy = Y; }
        sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
        sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
        if (!p.interp) {
            const int ix = border_index(sx, p.W, p.border), iy = border_index(sy, p.H, p.border);
            for (int c = 0; c < p.C; ++c) o[c] = (ix < 0 || iy < 0) ? p.cval[c] : src[((long long)iy * p.W + ix) * p.C + c];
        } else if (p.border == 0 && (sx >= p.W || sx + 1 < 0 || sy >= p.H || sy + 1 < 0)) {
            for (int c = 0; c < p.C; ++c) o[c] = p.cval[c];
        } else {
            int w[4] = {32 * (32 - fa) * (32 - fb), 32 * fa * (32 - fb), 32 * (32 - fa) * fb, 32 * fa * fb};
            if (w[0] > 32767) w[0] = 32767;
            const long long a0 = ((long long)sy * p.W + sx) * 3, a1 = a0 + (long long)p.W * 3;
            if (p.C == 3 && sx >= 0 && sx + 1 < p.W && sy >= 0 && sy + 1 < p.H && a1 + 8 <= (long long)p.H * p.W * 3) {
                // all four taps on the page: each source row's two pixels are 6 contiguous bytes, fetched as one unaligned 8-byte load
                unsigned long long q0, q1;
                __builtin_memcpy(&q0, src + a0, 8);
                __builtin_memcpy(&q1, src + a1, 8);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int v00 = (int)((q0 >> (8 * c)) & 0xff), v01 = (int)((q0 >> (8 * (c + 3))) & 0xff);
                    const int v10 = (int)((q1 >> (8 * c)) & 0xff), v11 = (int)((q1 >> (8 * (c + 3))) & 0xff);
                    const int r = (v00 * w[0] + v01 * w[1] + v10 * w[2] + v11 * w[3] + (1 << 14)) >> 15;
                    o[c] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
                }
            } else {
                const int x0 = border_index(sx, p.W, p.border), x1 = border_index(sx + 1, p.W, p.border);
                const int y0 = border_index(sy, p.H, p.border), y1 = border_index(sy + 1, p.H, p.border);
                for (int c = 0; c < p.C; ++c) {
                    const int cv = p.cval[c];
                    const int v00 = (x0 < 0 || y0 < 0) ? cv : src[((long long)y0 * p.W + x0) * p.C + c];
                    const int v01 = (x1 < 0 || y0 < 0) ? cv : src[((long long)y0 * p.W + x1) * p.C + c];
                    const int v10 = (x0 < 0 || y1 < 0) ? cv : src[((long long)y1 * p.W + x0) * p.C + c];
                    const int v11 = (x1 < 0 || y1 < 0) ? cv : src[((long long)y1 * p.W + x1) * p.C + c];
                    const int r = (v00 * w[0] + v01 * w[1] + v10 * w[2] + v11 * w[3] + (1 << 14)) >> 15;
                    o[c] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
                }
            }
        }
    }
    // a full 64-pixel segment of a 3-channel page whose rows are dword aligned leaves as 48 coalesced dwords per row
    const bool packed = p.dword_rows && (int)blockIdx.x * 64 + 64 <= p.W;
    if (packed) {
        unsigned char* tb = (unsigned char*)tile[ly];
        tb[lx * 3 + 0] = o[0];
        tb[lx * 3 + 1] = o[1];
        tb[lx * 3 + 2] = o[2];
        __syncthreads();
        if (lx < 48 && y < p.H) ((unsigned int*)(dst + ((long long)y * p.W + (long long)blockIdx.x * 64) * 3))[lx] = tile[ly][lx];
    } else if (live) {
        unsigned char* d = dst + ((long long)y * p.W + x) * p.C;
        for (int c = 0; c < p.C; ++c) d[c] = o[c];
    }
}

extern "C" int rtn_warp_affine_u8(rtn_handle_t h, const uint8_t* src, int H, int W, int C, const double* inv_map6, int interpolation,
                                  int border_mode, const uint8_t* cval4, uint8_t* dst) {
    if (!h) return RTN_EINVAL;
    if (!src || !dst || !inv_map6 || H < 1 || W < 1 || C < 1 || C > 4) return rtn_fail(h, RTN_EINVAL, "warp_affine: bad argument");
    if (H > 32767 || W > 32767) return rtn_fail(h, RTN_EINVAL, "warp_affine: image sides must be < 32768");
    if (interpolation != 0 && interpolation != 1) return rtn_fail(h, RTN_EINVAL, "warp_affine: interpolation must be nearest (0) or linear (1)");
    if (border_mode < 0 || border_mode > 3) return rtn_fail(h, RTN_EINVAL, "warp_affine: border mode must be constant/replicate/reflect101/wrap (0..3)");
    if (src == dst) return rtn_fail(h, RTN_EINVAL, "warp_affine: in-place is not supported");
    WarpParams p;
    for (int i = 0; i < 6; ++i) {
        if (!(inv_map6[i] == inv_map6[i]) || inv_map6[i] > 1e300 || inv_map6[i] < -1e300) return rtn_fail(h, RTN_EINVAL, "warp_affine: matrix entry %d is not finite", i);
        p.m[i] = inv_map6[i];
    }
    p.H = H; p.W = W; p.C = C; p.interp = interpolation; p.border = border_mode;
    p.dword_rows = C == 3 && (W * 3) % 4 == 0 && ((uintptr_t)dst & 3) == 0;
    for (int c = 0; c < 4; ++c) p.cval[c] = cval4 ? cval4[c] : 0;
    hipLaunchKernelGGL(warp_affine_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, h->stream, src, dst, p);
    RTN_CHECK_LAUNCH(h, "warp_affine_kernel");
    return RTN_OK;
}
