// rtn_conv_ksplit.hip — the ordered finish of the K-sliced persistent kernels (rtn_conv_halo8.hip / rtn_conv_gemm8.hip): the small-M,
// long-K layers of stage 5 (keras_resnet res5 branch2a / branch2b, C5_reduced, model/defineModel.py:183) fill a quarter of the chip
// with whole-K tiles, so their K loop is cut into S slices whose f32 partial sums land in caller-owned slabs [S][M][ld].  This
// kernel adds the slices IN SLICE ORDER (bit-reproducible, unlike atomics), the bias, the ReLU, and stores bf16.
#include "rtn_internal.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(256) void ksplit_finish_kernel(const float* __restrict__ slab, int S, long long M, int N, int ld,
                                                            const float* __restrict__ bias, int relu, unsigned short* __restrict__ out,
                                                            int out_ld) {
    const int ncol = N / 8;
    const long long total = M * ncol, slice = M * (long long)ld;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / ncol;
        const int n = (int)(i - m * ncol) * 8;
        const float* sp = slab + m * ld + n;
        float4 v0 = *reinterpret_cast<const float4*>(sp), v1 = *reinterpret_cast<const float4*>(sp + 4);
        for (int s = 1; s < S; ++s) {                                  // fixed slice order
            const float4 a0 = *reinterpret_cast<const float4*>(sp + s * slice), a1 = *reinterpret_cast<const float4*>(sp + s * slice + 4);
            v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
            v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
        }
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + n), b1 = *reinterpret_cast<const float4*>(bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        uint4 o;
        o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
        *reinterpret_cast<uint4*>(out + m * out_ld + n) = o;
    }
}

}  // namespace

// slab: f32 [S][M][ld] (ld >= N, both multiples of 8); out: bf16 [M][out_ld].
int rtn_conv_ksplit_finish(rtn_handle_t h, const float* slab, int S, long long M, int N, int ld, const float* bias, int relu, void* out,
                           int out_ld) {
    if (!slab || !out || S < 1 || M < 1 || N < 8 || N % 8 || ld % 8 || ld < N || out_ld % 8)
        return rtn_fail(h, RTN_EINVAL, "ksplit finish: bad argument");
    long long g = (M * (N / 8) + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(ksplit_finish_kernel, dim3((unsigned)g), dim3(256), 0, h->stream, slab, S, M, N, ld, bias, relu, (unsigned short*)out, out_ld);
    RTN_CHECK_LAUNCH(h, "ksplit_finish_kernel");
    return RTN_OK;
}
