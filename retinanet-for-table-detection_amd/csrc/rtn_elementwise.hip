// rtn_elementwise.hip — context management and the HBM-bound helpers of the conv stack:
// stem input packing, MaxPool 3x3/2 (TF 'same'), ReLU.
#include "rtn_internal.h"

extern "C" const char* rtn_version(void) { return "librtn 0.1 (gfx950)"; }

static char g_create_err[256] = "";
extern "C" const char* rtn_create_error(void) { return g_create_err; }

extern "C" int rtn_create(rtn_handle_t* out, int device) {
    if (!out) return RTN_EINVAL;
    *out = nullptr;
    g_create_err[0] = 0;
    rtn_ctx* h = new (std::nothrow) rtn_ctx();
    if (!h) return RTN_ENOMEM;
    memset(h, 0, sizeof(*h));
    h->device = device;
    hipDeviceProp_t prop;
    const char* what = "hipSetDevice";
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) { what = "hipMalloc(zero page)"; e = hipMalloc(&h->zero_page, 256); }
    if (e == hipSuccess) { what = "hipMemset(zero page)"; e = hipMemset(h->zero_page, 0, 256); }
    if (e == hipSuccess) { what = "hipGetDeviceProperties"; e = hipGetDeviceProperties(&prop, device); }
    if (e != hipSuccess) {
        snprintf(g_create_err, sizeof(g_create_err), "%s on device %d: %s", what, device, hipGetErrorString(e));
        (void)hipGetLastError();
        if (h->zero_page) (void)hipFree(h->zero_page);
        delete h;
        return RTN_EHIP;
    }
    h->num_cus = prop.multiProcessorCount;
    *out = h;
    return RTN_OK;
}

extern "C" int rtn_destroy(rtn_handle_t h) {
    if (!h) return RTN_EINVAL;
    if (h->zero_page) (void)hipFree(h->zero_page);
    delete h;
    return RTN_OK;
}

extern "C" int rtn_set_stream(rtn_handle_t h, void* stream) {
    if (!h) return RTN_EINVAL;
    h->stream = (hipStream_t)stream;
    return RTN_OK;
}

extern "C" const char* rtn_last_error(rtn_handle_t h) { return h ? h->err : "null handle"; }

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    const __bf16 hb = (__bf16)f;
    return __builtin_bit_cast(unsigned short, hb);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// src_dtype: 0 bf16, 1 f32, 2 u8 (normalised x/127.5 - 1, model/utils.py:43-46)
template <int SRC, int DST>
__global__ __launch_bounds__(256) void stem_pack_kernel(const void* __restrict__ src, void* __restrict__ dst, int B, int H,
                                                        int W, int Hp, int Wp) {
    const long long total = (long long)B * Hp * Wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wp);
        const long long r = i / Wp;
        const int y = (int)(r % Hp);
        const int b = (int)(r / Hp);
        float v[3] = {0.f, 0.f, 0.f};
        const int sy = y - 3, sx = x - 3;
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
            const long long si = (((long long)b * H + sy) * W + sx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if constexpr (SRC == 0) v[c] = bf16_bits_to_f32(((const unsigned short*)src)[si + c]);
                else if constexpr (SRC == 1) v[c] = ((const float*)src)[si + c];
                else v[c] = __fsub_rn(__fdiv_rn((float)((const unsigned char*)src)[si + c], 127.5f), 1.0f);
            }
        }
        if constexpr (DST == RTN_BF16) {
            uint2 o;
            o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16_bits(v[2]);
            ((uint2*)dst)[i] = o;
        } else {
            ((float4*)dst)[i] = make_float4(v[0], v[1], v[2], 0.f);
        }
    }
}

// one thread = 16 bytes of channels of one output pixel
template <int ES>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const char* __restrict__ in, char* __restrict__ out, int B, int Hin,
                                                           int Win, int C, int Hout, int Wout, int pad_t, int pad_l) {
    constexpr int CE = 16 / ES;
    const int cv = C / CE;
    const long long total = (long long)B * Hout * Wout * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv);
        long long r = i / cv;
        const int ox = (int)(r % Wout);
        r /= Wout;
        const int oy = (int)(r % Hout);
        const int b = (int)(r / Hout);
        float mx[CE];
#pragma unroll
        for (int j = 0; j < CE; ++j) mx[j] = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy * 2 - pad_t + kh;
            if ((unsigned)iy >= (unsigned)Hin) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = ox * 2 - pad_l + kw;
                if ((unsigned)ix >= (unsigned)Win) continue;
                const uint4 q = *reinterpret_cast<const uint4*>(in + ((((long long)b * Hin + iy) * Win + ix) * C + cc * CE) * ES);
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
                if constexpr (ES == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mx[2 * j] = fmaxf(mx[2 * j], __uint_as_float(w4[j] << 16));
                        mx[2 * j + 1] = fmaxf(mx[2 * j + 1], __uint_as_float(w4[j] & 0xffff0000u));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], __uint_as_float(w4[j]));
                }
            }
        }
        uint4 o;
        if constexpr (ES == 2) {
            // inputs are bf16 values, so the max is representable: truncation is exact
            o.x = (__float_as_uint(mx[0]) >> 16) | (__float_as_uint(mx[1]) & 0xffff0000u);
            o.y = (__float_as_uint(mx[2]) >> 16) | (__float_as_uint(mx[3]) & 0xffff0000u);
            o.z = (__float_as_uint(mx[4]) >> 16) | (__float_as_uint(mx[5]) & 0xffff0000u);
            o.w = (__float_as_uint(mx[6]) >> 16) | (__float_as_uint(mx[7]) & 0xffff0000u);
        } else {
            o.x = __float_as_uint(mx[0]); o.y = __float_as_uint(mx[1]);
            o.z = __float_as_uint(mx[2]); o.w = __float_as_uint(mx[3]);
        }
        *reinterpret_cast<uint4*>(out + i * 16) = o;
    }
}

template <int ES>
__global__ __launch_bounds__(256) void relu_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long long nvec) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        uint4 q = in[i];
        unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ES == 2) {
                // negative (sign bit set) halves -> +0
                const unsigned lo = (w4[j] & 0x8000u) ? 0u : (w4[j] & 0xffffu);
                const unsigned hi = (w4[j] & 0x80000000u) ? 0u : (w4[j] & 0xffff0000u);
                w4[j] = lo | hi;
            } else {
                w4[j] = (w4[j] & 0x80000000u) ? 0u : w4[j];
            }
        }
        out[i] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
}

inline unsigned grid_for(long long work, int block = 256, int cap = 256 * 8) {
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace

extern "C" int rtn_stem_pack(rtn_handle_t h, const void* src, int src_dtype, void* dst, int dst_dtype, int B, int H, int W,
                             int Hp, int Wp) {
    if (!h) return RTN_EINVAL;
    if (!src || !dst || B < 1 || H < 1 || W < 1) return rtn_fail(h, RTN_EINVAL, "stem_pack: bad argument");
    if (src_dtype < 0 || src_dtype > 2 || (dst_dtype != RTN_BF16 && dst_dtype != RTN_F32))
        return rtn_fail(h, RTN_EINVAL, "stem_pack: bad dtype");
    if (Hp < H + 6 || Wp < W + 6 || (Wp & 1)) return rtn_fail(h, RTN_EINVAL, "stem_pack: padded extent %dx%d too small for %dx%d", Hp, Wp, H, W);
    if ((uintptr_t)dst & 15) return rtn_fail(h, RTN_EINVAL, "stem_pack: dst not 16-byte aligned");
    const long long total = (long long)B * Hp * Wp;
    dim3 g(grid_for(total)), b(256);      // (a 2-D grid without the 64-bit divisions measured slower in the step: 36 us against 25)
#define LAUNCH(S, D) hipLaunchKernelGGL((stem_pack_kernel<S, D>), g, b, 0, h->stream, src, dst, B, H, W, Hp, Wp)
    if (dst_dtype == RTN_BF16) {
        if (src_dtype == 0) LAUNCH(0, RTN_BF16); else if (src_dtype == 1) LAUNCH(1, RTN_BF16); else LAUNCH(2, RTN_BF16);
    } else {
        if (src_dtype == 0) LAUNCH(0, RTN_F32); else if (src_dtype == 1) LAUNCH(1, RTN_F32); else LAUNCH(2, RTN_F32);
    }
#undef LAUNCH
    RTN_CHECK_LAUNCH(h, "stem_pack_kernel");
    return RTN_OK;
}

extern "C" int rtn_maxpool3x3s2_tfsame_fwd(rtn_handle_t h, const void* in, void* out, int dtype, int B, int Hin, int Win, int C) {
    if (!h) return RTN_EINVAL;
    if (!in || !out || B < 1 || Hin < 1 || Win < 1 || C < 1) return rtn_fail(h, RTN_EINVAL, "maxpool: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "maxpool: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((C * es) % 16) return rtn_fail(h, RTN_EINVAL, "maxpool: C=%d is not a whole number of 16-byte chunks", C);
    if (((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return rtn_fail(h, RTN_EINVAL, "maxpool: pointer not 16-byte aligned");
    const int Hout = (Hin + 1) / 2, Wout = (Win + 1) / 2;
    // TF 'same': pad_total = max((out-1)*s + k - in, 0), pad_before = floor(pad_total/2)
    int pth = (Hout - 1) * 2 + 3 - Hin; if (pth < 0) pth = 0;
    int ptw = (Wout - 1) * 2 + 3 - Win; if (ptw < 0) ptw = 0;
    const long long total = (long long)B * Hout * Wout * (C * es / 16);
    dim3 g(grid_for(total, 256, 256 * 16)), b(256);
    if (es == 2) hipLaunchKernelGGL((maxpool3x3s2_kernel<2>), g, b, 0, h->stream, (const char*)in, (char*)out, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2);
    else         hipLaunchKernelGGL((maxpool3x3s2_kernel<4>), g, b, 0, h->stream, (const char*)in, (char*)out, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2);
    RTN_CHECK_LAUNCH(h, "maxpool3x3s2_kernel");
    return RTN_OK;
}

extern "C" int rtn_relu(rtn_handle_t h, const void* in, void* out, int dtype, int64_t n) {
    if (!h) return RTN_EINVAL;
    if (!in || !out || n < 1) return rtn_fail(h, RTN_EINVAL, "relu: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "relu: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((n * es) % 16 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return rtn_fail(h, RTN_EINVAL, "relu: size/pointer not 16-byte aligned");
    const long long nvec = n * es / 16;
    dim3 g(grid_for(nvec)), b(256);
    if (es == 2) hipLaunchKernelGGL((relu_kernel<2>), g, b, 0, h->stream, (const uint4*)in, (uint4*)out, nvec);
    else         hipLaunchKernelGGL((relu_kernel<4>), g, b, 0, h->stream, (const uint4*)in, (uint4*)out, nvec);
    RTN_CHECK_LAUNCH(h, "relu_kernel");
    return RTN_OK;
}

namespace {
// activations entering an fp8 layer: e4m3(clamp(x * scale, +-448)), 8 elements per thread (16 B of bf16 / 32 B of f32 in, 8 B out)
template <int ES>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const void* __restrict__ src, uint2* __restrict__ dst, long long nvec, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float v[8];
        if (ES == 2) {
            const uint4 q = reinterpret_cast<const uint4*>(src)[i];
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = __uint_as_float(w4[j] << 16);
                v[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u);
            }
        } else {
            const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float q = v[j] * scale;
            v[j] = q > 448.f ? 448.f : (q < -448.f ? -448.f : q);
        }
        unsigned lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
        dst[i] = make_uint2(lo, hi);
    }
}
}  // namespace

extern "C" int rtn_quantize_fp8(rtn_handle_t h, const void* src, int src_dtype, void* dst, int64_t n, float scale) {
    if (!h) return RTN_EINVAL;
    if (!src || !dst || n < 1 || !(scale > 0.f)) return rtn_fail(h, RTN_EINVAL, "quantize_fp8: bad argument");
    if (src_dtype != RTN_BF16 && src_dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "quantize_fp8: source must be bf16 or f32");
    if (n % 8 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return rtn_fail(h, RTN_EINVAL, "quantize_fp8: n must be a multiple of 8, src 16-byte and dst 8-byte aligned");
    const long long nvec = n / 8;
    dim3 g(grid_for(nvec)), b(256);
    if (src_dtype == RTN_BF16) hipLaunchKernelGGL((quantize_fp8_kernel<2>), g, b, 0, h->stream, src, (uint2*)dst, nvec, scale);
    else                       hipLaunchKernelGGL((quantize_fp8_kernel<4>), g, b, 0, h->stream, src, (uint2*)dst, nvec, scale);
    RTN_CHECK_LAUNCH(h, "quantize_fp8_kernel");
    return RTN_OK;
}

namespace {
// UpsampleLike (model/layers.py:89-98): legacy TF nearest, src = min(floor(dst * in/out), in-1), ratio in float32
__global__ __launch_bounds__(256) void upsample_nearest_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B, int Hs, int Ws,
                                                               int Hd, int Wd, int cv, float rh, float rw) {
    const long long total = (long long)B * Hd * Wd * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int x = (int)(r % Wd);
        r /= Wd;
        const int y = (int)(r % Hd);
        const int b = (int)(r / Hd);
        int sy = (int)floorf((float)y * rh), sx = (int)floorf((float)x * rw);
        sy = sy < Hs - 1 ? sy : Hs - 1;
        sx = sx < Ws - 1 ? sx : Ws - 1;
        dst[i] = src[(((long long)b * Hs + sy) * Ws + sx) * cv + c];
    }
}

// utils.preprocess_image (model/utils.py:19-47): float32 cast, then mode 0 'tf' x/127.5-1, 1 'caffe' BGR mean subtraction,
// 2 'custom_tf' x/scale - sub (model/Parameters.py:23-24)
template <typename SRC>
__global__ __launch_bounds__(256) void preprocess_kernel(const SRC* __restrict__ src, float* __restrict__ dst, long long n, int mode,
                                                         float scale, float sub) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = (float)src[i];
        if (mode == 0) v = __fsub_rn(__fdiv_rn(v, 127.5f), 1.0f);
        else if (mode == 1) { const int c = (int)(i % 3); v = __fsub_rn(v, c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f)); }
        else if (mode == 2) v = __fsub_rn(__fdiv_rn(v, scale), sub);
        dst[i] = v;
    }
}
}  // namespace

extern "C" int rtn_upsample_nearest(rtn_handle_t h, const void* src, void* dst, int dtype, int B, int Hs, int Ws, int Hd, int Wd, int C) {
    if (!h) return RTN_EINVAL;
    if (!src || !dst || B < 1 || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || C < 1) return rtn_fail(h, RTN_EINVAL, "upsample_nearest: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "upsample_nearest: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((C * es) % 16 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return rtn_fail(h, RTN_EINVAL, "upsample_nearest: alignment");
    const int cv = C * es / 16;
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(grid_for((long long)B * Hd * Wd * cv)), dim3(256), 0, h->stream, (const uint4*)src,
                       (uint4*)dst, B, Hs, Ws, Hd, Wd, cv, (float)Hs / (float)Hd, (float)Ws / (float)Wd);
    RTN_CHECK_LAUNCH(h, "upsample_nearest_kernel");
    return RTN_OK;
}

extern "C" int rtn_preprocess_image(rtn_handle_t h, const void* src, int src_dtype, float* dst, int64_t n, int mode, float scale, float sub) {
    if (!h) return RTN_EINVAL;
    if (!src || !dst || n < 1 || mode < 0 || mode > 2) return rtn_fail(h, RTN_EINVAL, "preprocess_image: bad argument");
    if (src_dtype != RTN_F32 && src_dtype != 2) return rtn_fail(h, RTN_EINVAL, "preprocess_image: src dtype must be f32 or u8");
    if (mode == 1 && n % 3) return rtn_fail(h, RTN_EINVAL, "preprocess_image: caffe mode needs 3 interleaved channels");
    if (src_dtype == 2) hipLaunchKernelGGL((preprocess_kernel<unsigned char>), dim3(grid_for(n)), dim3(256), 0, h->stream, (const unsigned char*)src, dst, (long long)n, mode, scale, sub);
    else                hipLaunchKernelGGL((preprocess_kernel<float>), dim3(grid_for(n)), dim3(256), 0, h->stream, (const float*)src, dst, (long long)n, mode, scale, sub);
    RTN_CHECK_LAUNCH(h, "preprocess_kernel");
    return RTN_OK;
}
