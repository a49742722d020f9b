// rtn_stem.hip — the whole stem in one kernel (inference): ZeroPadding2D(3) + conv1 7x7/2 (frozen BN folded) + ReLU +
// MaxPool 3x3/2 'same' (keras_resnet ResNet*: conv1, bn_conv1, conv1_relu, pool1; model/defineModel.py:357-389) on the
// packed [B][Hp][Wp][4] image rtn_stem_pack writes (zero border of 3 included, 8 bytes per pixel).
//
// Why: unfused, conv1 writes 34 MB per 800x1333 image that pool1 reads straight back (the largest tensor of the network),
// and the implicit-GEMM stem pulls every input pixel ~16x through L2 (windows of neighbouring outputs overlap 4x4).
// Here a workgroup owns a 4 x 16 tile of POOLED pixels: it stages the input patch (23 x 72 packed pixels, 16-byte
// range-checked buffer loads) once, reads MFMA A fragments straight out of it (the 64-byte run of a kernel row for output x starts at pixel 2x: always
// 16-byte aligned), multiplies the 9 x 33 conv outputs the tile's pooling windows need (halo recomputed: 1.16x), and pools
// them from LDS.  HBM traffic: the image in, the pooled tensor out.
#include "rtn_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int PT_H = 4, PT_W = 16;                  // pooled tile
constexpr int CT_H = 2 * PT_H + 1, CT_W = 2 * PT_W + 1;   // conv outputs it needs: 9 x 33
constexpr int CT_N = CT_H * CT_W;                   // 297
constexpr int NSUB = (CT_N + 15) / 16;              // 19 MFMA row blocks
constexpr int SUB_PER_WAVE = (NSUB + 3) / 4;        // 5
constexpr int PA_H = 2 * (CT_H - 1) + 7;            // 23 input rows
constexpr int PA_W = 72;                            // 2*32 + 7 = 71 input pixels, padded to 72
constexpr int PA_ROW = PA_W * 8;                    // bytes per patch row (4 ch bf16 per pixel)
constexpr int PA_BYTES = PA_H * PA_ROW;             // 13248
constexpr int W_ROW = 7 * 64 + 16;                  // 7 kernel rows x 64 B per output channel, +16 B: conflict-free b128 reads
constexpr int W_BYTES = 64 * W_ROW;                 // 29696
constexpr int CV_PX = 144;                          // conv tile: 64 ch bf16 per pixel + 16 B (conflict-free 8-byte writes)
constexpr int CV_BYTES = CT_N * CV_PX;              // aliases patch + weights
constexpr int LDS_BYTES = (PA_BYTES + W_BYTES) > CV_BYTES ? (PA_BYTES + W_BYTES) : CV_BYTES;

__device__ __forceinline__ unsigned short to_bf16(float f) {
    const __bf16 hb = (__bf16)f;
    return __builtin_bit_cast(unsigned short, hb);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void stem_fused_kernel(const char* __restrict__ xp /* [B][Hp][Wp][4] bf16 */, unsigned xp_bytes,
                                                         const char* __restrict__ wk /* [>=64][512 B] */,
                                                         const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                         int Hp, int Wp, int H1, int W1, int H2, int W2, int pool_pt, int pool_pl) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    char* patch = lds;
    char* wl = lds + PA_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.z;
    const int py0 = blockIdx.y * PT_H, px0 = blockIdx.x * PT_W;
    const int cy0 = 2 * py0 - pool_pt, cx0 = 2 * px0 - pool_pl;        // first conv row / column of the tile
    // conv output (cy, cx) reads packed rows 2cy .. 2cy+6 and packed pixels 2cx .. 2cx+7
    const long long pbase = (((long long)b * Hp + 2 * cy0) * Wp + 2 * cx0) * 8;   // may be negative (tile starts at conv row/col -1)

    // ---- stage the weights (64 channels x 7 kernel rows x 64 B) and the input patch (23 rows x 36 chunks of 16 B).
    // Out-of-buffer chunks read as zeros (range-checked buffer loads); chunks that wrap into a neighbouring row only feed
    // conv outputs outside the image, which the pooling ignores.
    for (int i = t; i < 64 * 28; i += 256) {
        const int n = i / 28, q = i - n * 28;
        *reinterpret_cast<uint4*>(wl + n * W_ROW + q * 16) = *reinterpret_cast<const uint4*>(wk + (long long)n * 512 + q * 16);
    }
    {
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xp), (short)0, (int)xp_bytes, 0x00020000);
        for (int i = t; i < PA_H * (PA_ROW / 16); i += 256) {
            const int r = i / (PA_ROW / 16), q = i - r * (PA_ROW / 16);
            const long long off = pbase + (long long)r * Wp * 8 + q * 16;
            const unsigned voff = (off >= 0 && off + 16 <= (long long)xp_bytes) ? (unsigned)off : 0xFFFFFF00u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, (int)voff, 0, 0);
            *reinterpret_cast<u32x4*>(patch + r * PA_ROW + q * 16) = v;
        }
    }
    __syncthreads();

    // ---- 7 x (A fragment straight from the patch) x (4 channel blocks).  Operands swapped (D = W . X^T): a lane ends up with
    // 4 consecutive channels of ONE conv pixel, which it can write as 8 bytes.
    const int m = lane & 15, kq = lane >> 4;
    int abase[SUB_PER_WAVE];
#pragma unroll
    for (int s = 0; s < SUB_PER_WAVE; ++s) {
        int i = (wave * SUB_PER_WAVE + s) * 16 + m;
        if (i >= CT_N) i = 0;                                           // rows past the tile: computed, never stored
        const int ly = i / CT_W, lx = i - ly * CT_W;
        abase[s] = (2 * ly) * PA_ROW + (2 * lx) * 8 + kq * 16;
    }
    f32x4 acc[SUB_PER_WAVE][4];
#pragma unroll
    for (int s = 0; s < SUB_PER_WAVE; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[s][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        uint4 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const uint4*>(wl + (j * 16 + m) * W_ROW + kh * 64 + kq * 16);
#pragma unroll
        for (int s = 0; s < SUB_PER_WAVE; ++s) {
            const uint4 af = *reinterpret_cast<const uint4*>(patch + abase[s] + kh * PA_ROW);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[s][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, af),
                                                                    acc[s][j], 0, 0, 0);
        }
    }
    __syncthreads();                                                    // patch and weights are dead: the conv tile takes their place

    // ---- bias + ReLU -> bf16 conv tile in LDS; positions outside the conv image become -inf for the pooling.
    // acc[s][j][r] = conv pixel (subtile s, row m), channel j*16 + kq*4 + r
    float bn[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bn[j][r] = bias ? bias[j * 16 + kq * 4 + r] : 0.f;
#pragma unroll
    for (int s = 0; s < SUB_PER_WAVE; ++s) {
        const int i = (wave * SUB_PER_WAVE + s) * 16 + m;
        if (i >= CT_N) continue;
        const int ly = i / CT_W, lx = i - ly * CT_W;
        const bool inside = (unsigned)(cy0 + ly) < (unsigned)H1 && (unsigned)(cx0 + lx) < (unsigned)W1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 o;
            if (inside) {
                o.x = (unsigned)to_bf16(fmaxf(acc[s][j][0] + bn[j][0], 0.f)) | ((unsigned)to_bf16(fmaxf(acc[s][j][1] + bn[j][1], 0.f)) << 16);
                o.y = (unsigned)to_bf16(fmaxf(acc[s][j][2] + bn[j][2], 0.f)) | ((unsigned)to_bf16(fmaxf(acc[s][j][3] + bn[j][3], 0.f)) << 16);
            } else {
                o.x = o.y = 0xFF80FF80u;
            }
            *reinterpret_cast<uint2*>(lds + i * CV_PX + j * 32 + kq * 8) = o;
        }
    }
    __syncthreads();

    // ---- 3x3/2 max over the tile: one thread = 8 channels of one pooled pixel
    for (int it = t; it < PT_H * PT_W * 8; it += 256) {
        const int cg = it & 7, pp = it >> 3;
        const int ppy = pp / PT_W, ppx = pp - ppy * PT_W;
        const int py = py0 + ppy, px = px0 + ppx;
        if (py >= H2 || px >= W2) continue;
        float mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int i = (2 * ppy + dy) * CT_W + 2 * ppx + dx;
                const uint4 q = *reinterpret_cast<const uint4*>(lds + i * CV_PX + cg * 16);
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mx[2 * e] = fmaxf(mx[2 * e], __uint_as_float(w4[e] << 16));
                    mx[2 * e + 1] = fmaxf(mx[2 * e + 1], __uint_as_float(w4[e] & 0xffff0000u));
                }
            }
        uint4 o;
        o.x = (__float_as_uint(mx[0]) >> 16) | (__float_as_uint(mx[1]) & 0xffff0000u);
        o.y = (__float_as_uint(mx[2]) >> 16) | (__float_as_uint(mx[3]) & 0xffff0000u);
        o.z = (__float_as_uint(mx[4]) >> 16) | (__float_as_uint(mx[5]) & 0xffff0000u);
        o.w = (__float_as_uint(mx[6]) >> 16) | (__float_as_uint(mx[7]) & 0xffff0000u);
        *reinterpret_cast<uint4*>(out + ((((long long)b * H2 + py) * W2 + px) * 64 + cg * 8)) = o;
    }
}

}  // namespace

extern "C" int rtn_stem_conv_pool(rtn_handle_t h, const void* packed, int Hp, int Wp, const void* w_packed, int w_rows,
                                  const float* bias, void* out, int B, int H, int W) {
    if (!h) return RTN_EINVAL;
    if (!packed || !w_packed || !out) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: null pointer");
    if (B < 1 || H < 1 || W < 1 || B > 65535) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: bad extent");
    if (w_rows < 64) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: needs the 64 packed stem filters");
    if (((uintptr_t)packed & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)out & 15) || ((uintptr_t)bias & 3) || (Wp & 1))
        return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: misaligned pointer or odd Wp");
    const int H1 = (H - 1) / 2 + 1, W1 = (W - 1) / 2 + 1;              // ZeroPadding2D(3) + 7x7/2 'valid'
    if (Hp < 2 * (H1 - 1) + 7 || Wp < 2 * (W1 - 1) + 8) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: packed image %dx%d too small for %dx%d", Hp, Wp, H, W);
    const long long bytes = (long long)B * Hp * Wp * 8;
    if (bytes >= 0xFFFFFF00ll) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: packed batch exceeds the 4 GiB buffer-descriptor range");
    const int H2 = (H1 + 1) / 2, W2 = (W1 + 1) / 2;                    // 3x3/2 'same'
    const int pth = (H2 - 1) * 2 + 3 - H1, ptw = (W2 - 1) * 2 + 3 - W1;
    const int pool_pt = (pth > 0 ? pth : 0) / 2, pool_pl = (ptw > 0 ? ptw : 0) / 2;
    dim3 grid((W2 + PT_W - 1) / PT_W, (H2 + PT_H - 1) / PT_H, B), block(256);
    if (grid.y > 65535) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: image too tall");
    hipLaunchKernelGGL(stem_fused_kernel, grid, block, 0, h->stream, (const char*)packed, (unsigned)bytes, (const char*)w_packed, bias,
                       (unsigned short*)out, Hp, Wp, H1, W1, H2, W2, pool_pt, pool_pl);
    RTN_CHECK_LAUNCH(h, "stem_fused_kernel");
    return RTN_OK;
}
