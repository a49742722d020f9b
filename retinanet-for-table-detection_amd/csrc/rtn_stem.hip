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
// Round 3: persistent workgroups (filters resident in LDS, next patch prefetched into registers) and, optionally, the first 1x1
// convolution of the network (res2a_branch2a + BN + ReLU) applied to the pooled pixels before they leave the registers.
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
constexpr int CV_BYTES = CT_N * CV_PX;              // aliases the patch (dead after the MFMAs); the filters stay resident
constexpr int LDS_BYTES = W_BYTES + (PA_BYTES > CV_BYTES ? PA_BYTES : CV_BYTES);      // 72,464 B: two workgroups per CU
constexpr int W2A_ROW = 128 + 16;                   // branch2a filters: 64 (permuted) rows x 128 B, +16 B: conflict-free b128 reads
constexpr int LDS_BYTES_A2 = LDS_BYTES + 64 * W2A_ROW;                                 // 81,680 B: still two per CU (163,360 of 163,840)

__device__ __forceinline__ unsigned short to_bf16(float f) {
    const __bf16 hb = (__bf16)f;
    return __builtin_bit_cast(unsigned short, hb);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {          // v_cvt_pk_bf16_f32
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned relu_pk(unsigned w) {                   // v_pk_max_i16 with 0: negative bf16 (and -0) -> +0
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}

struct StemParams {
    const char* xp;             // [B][Hp][Wp][4] bf16
    const char* wk;             // [>= 64][512 B]
    const float* bias;
    unsigned short* out;        // pool1 [B][H2][W2][64] bf16
    const char* w2a;            // A2: res2a_branch2a filters [>= 64][64] bf16 (BN folded), K-contiguous
    const float* b2a;           // A2: its folded BN shift [64]
    unsigned short* a_out;      // A2: [B][H2][W2][64] bf16
    unsigned char* idx;         // IDX: [B][H2][W2][64] u8, the winning tap kh * 3 + kw of every pooled element (training: the pool's backward)
    unsigned xp_bytes;
    int Hp, Wp, H1, W1, H2, W2, pool_pt, pool_pl;
    int tiles_x, tiles_y, ntiles;
};

// MFMA row (16 f + 4 q + r) of the branch2a product -> output channel 32 (f >> 1) + 8 q + 4 (f & 1) + r: two accumulator fragments
// of a lane are then 8 consecutive channels of its pixel (the permutation of rtn_bottleneck.hip)
__device__ __forceinline__ int stem_perm_row(int rho) {
    const int f = rho >> 4, q = (rho >> 2) & 3, r = rho & 3;
    return 32 * (f >> 1) + 8 * q + 4 * (f & 1) + r;
}

// PERSISTENT: two workgroups per CU walk the 4 x 16 pooled tiles of the batch; the filters are staged into LDS once per workgroup
// (they were 30 KB per tile: 250 MB of L2 reads per batch of 8), the next tile's input patch is requested into registers before
// this tile's MFMAs and written to LDS after this tile's pooling.  A2: the pooled pixels (8 channels per lane, which is exactly a
// B-operand fragment of the transposed product) also go through res2a_branch2a (1x1, 64 -> 64, BN, ReLU; keras_resnet bottleneck
// behind model/defineModel.py:376-380): the layer's launch and its read of pool1 disappear.
template <bool A2, bool IDX>
__global__ __launch_bounds__(256, 2) void stem_fused_kernel(const StemParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* wl = lds;                      // filters, resident
    char* patch = lds + W_BYTES;         // input patch of the tile; after the MFMAs the conv tile takes its place
    char* cv = patch;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, kq = lane >> 4;
    constexpr int PCH = PA_H * (PA_ROW / 16);             // 828 chunks of 16 B per patch
    constexpr int NPF = (PCH + 255) / 256;                // 4 per thread
    constexpr int KH_UNROLL = 7;

    for (int i = t; i < 64 * 28; i += 256) {
        const int n = i / 28, q = i - n * 28;
        *reinterpret_cast<uint4*>(wl + n * W_ROW + q * 16) = *reinterpret_cast<const uint4*>(p.wk + (long long)n * 512 + q * 16);
    }
    // the folded BN shifts live in the 16-byte pads of the filter rows: float4 #(4 j + kq) = channels 16 j + 4 kq .. + 4 in the pad of row
    // 4 j + kq (registers are what this kernel is short of)
    if (t < 16) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) v = (f32x4){p.bias[4 * t], p.bias[4 * t + 1], p.bias[4 * t + 2], p.bias[4 * t + 3]};
        *reinterpret_cast<f32x4*>(wl + t * W_ROW + 7 * 64) = v;
    }
    // A2: the branch2a filters in LDS behind the conv tile, rows already permuted: row rho holds filter perm(rho); the shifts of
    // this lane's accumulator rows in registers
    char* w2l = lds + LDS_BYTES;
    if (A2) {
        for (int i = t; i < 64 * 8; i += 256) {
            const int rho = i >> 3, q = i & 7;
            *reinterpret_cast<uint4*>(w2l + rho * W2A_ROW + q * 16) = *reinterpret_cast<const uint4*>(p.w2a + (stem_perm_row(rho) * 64 + q * 8) * 2);
        }
        if (t < 16) {   // shifts of accumulator fragment f, lane quarter kq (4 consecutive channels from perm(16 f + 4 kq)): pad of row 4 f + kq
            const int f = t >> 2, kq_ = t & 3;
            const float* bs = p.b2a + stem_perm_row(16 * f + 4 * kq_);
            *reinterpret_cast<f32x4*>(w2l + t * W2A_ROW + 128) = (f32x4){bs[0], bs[1], bs[2], bs[3]};
        }
    }

    // workgroup -> tiles: workgroup b runs on XCD b % 8; every XCD walks a contiguous range of tiles (neighbouring tiles share
    // their halo columns / rows through ONE L2)
    int tile, tile_end, tstep;
    {
        const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
        const int nx = nwg < 8 ? nwg : 8;                  // fewer than 8 workgroups: as many ranges as workgroups
        const int xcd = bid % nx, idx = bid / nx;
        const int wgs = nwg / nx + (xcd < nwg % nx ? 1 : 0);
        const int lo = (int)((long long)p.ntiles * xcd / nx), hi = (int)((long long)p.ntiles * (xcd + 1) / nx);
        tile = lo + idx; tile_end = hi; tstep = wgs;
    }
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.xp), (short)0, (int)p.xp_bytes, 0x00020000);
    u32x4 pf[NPF];
    auto tile_geo = [&](int T, int& b, int& py0, int& px0) {
        const int tx = T % p.tiles_x, r = T / p.tiles_x;
        b = r / p.tiles_y;
        py0 = (r - b * p.tiles_y) * PT_H;
        px0 = tx * PT_W;
    };
    auto request = [&](int T) {                           // the input patch of tile T -> registers (range-checked: zeros outside)
        int b, py0, px0;
        tile_geo(T < p.ntiles ? T : 0, b, py0, px0);
        const int cy0 = 2 * py0 - p.pool_pt, cx0 = 2 * px0 - p.pool_pl;
        const long long pbase = (((long long)b * p.Hp + 2 * cy0) * p.Wp + 2 * cx0) * 8;   // may be negative (conv row / column -1)
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int i = t + 256 * u;
            const int r = i / (PA_ROW / 16), q = i - r * (PA_ROW / 16);
            const long long off = pbase + (long long)r * p.Wp * 8 + q * 16;
            const bool ok = i < PCH && T < tile_end && off >= 0 && off + 16 <= (long long)p.xp_bytes;
            pf[u] = __builtin_amdgcn_raw_buffer_load_b128(srd, (int)(ok ? (unsigned)off : 0xFFFFFF00u), 0, 0);
        }
    };
    request(tile);
#pragma unroll 1
    for (; tile < tile_end; tile += tstep) {
        int b, py0, px0;
        tile_geo(tile, b, py0, px0);
        const int cy0 = 2 * py0 - p.pool_pt, cx0 = 2 * px0 - p.pool_pl;        // first conv row / column of the tile
        // conv output (cy, cx) reads packed rows 2cy .. 2cy+6 and packed pixels 2cx .. 2cx+7.  Chunks that wrap into a neighbouring
        // row only feed conv outputs outside the image, which the pooling ignores.
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int i = t + 256 * u;
            if (i < PCH) *reinterpret_cast<u32x4*>(patch + (i / (PA_ROW / 16)) * PA_ROW + (i % (PA_ROW / 16)) * 16) = pf[u];
        }
        __syncthreads();
        request(tile + tstep);                            // in flight under this tile's MFMAs and pooling

        // ---- 7 x (A fragment straight from the patch) x (4 channel blocks).  Operands swapped (D = W . X^T): a lane ends up with
        // 4 consecutive channels of ONE conv pixel, which it can write as 8 bytes.
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
            for (int j = 0; j < 4; ++j) acc[s][j] = *reinterpret_cast<const f32x4*>(wl + (4 * j + kq) * W_ROW + 7 * 64);     // bias-initialised, as every conv kernel of the library
#pragma unroll KH_UNROLL     // fully unrolled in every form since the BN shifts moved into the filter rows' LDS pads (round 4: the A2 form used to
        for (int kh = 0; kh < 7; ++kh) {        // spill and ran the loop rolled: 0.110 -> 0.098 ms)
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
        __syncthreads();                                                    // the patch is dead: the conv tile takes its place

        // ---- bias + ReLU -> bf16 conv tile in LDS; positions outside the conv image become -inf for the pooling.
        // acc[s][j][r] = conv pixel (subtile s, row m), channel j*16 + kq*4 + r
#pragma unroll
        for (int s = 0; s < SUB_PER_WAVE; ++s) {
            const int i = (wave * SUB_PER_WAVE + s) * 16 + m;
            if (i >= CT_N) continue;
            const int ly = i / CT_W, lx = i - ly * CT_W;
            const bool inside = (unsigned)(cy0 + ly) < (unsigned)p.H1 && (unsigned)(cx0 + lx) < (unsigned)p.W1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 o;
                if (inside) {       // round to bf16, then ReLU on the packed pair as a signed 16-bit max with 0 (== ReLU, then round)
                    o.x = relu_pk(pack_bf16(acc[s][j][0], acc[s][j][1]));
                    o.y = relu_pk(pack_bf16(acc[s][j][2], acc[s][j][3]));
                } else {            // outside the conv image: never the maximum.  Every pooling window holds a pixel inside and
                    o.x = o.y = IDX ? 0xFF80FF80u : 0u;    // ReLU outputs are >= 0, so 0 does what -inf does (IDX keeps -inf: it records WHICH tap won)
                }
                *reinterpret_cast<uint2*>(cv + i * CV_PX + j * 32 + kq * 8) = o;
            }
        }
        __syncthreads();

        // ---- 3x3/2 max over the tile.  Wave w owns pooled row w of the tile, lane (kq, m) pixel m and, for h = 0 / 1, the 8 channels
        // 32 h + 8 kq .. + 8: a B-operand fragment of the branch2a product as it stands.
        const int py = py0 + wave, px = px0 + m;
        const bool live = py < p.H2 && px < p.W2;
        uint4 pooled[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cg = 4 * h + kq;
            uint4 o;
            if (!IDX) {
                // non-negative bf16 values order like unsigned 16-bit integers: the 3x3 maximum is 9 x 4 v_pk_max_u16 on the packed tile
                u16x8 mxp = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int i = (2 * wave + dy) * CT_W + 2 * m + dx;
                        const uint4 q = *reinterpret_cast<const uint4*>(cv + i * CV_PX + cg * 16);
                        mxp = __builtin_elementwise_max(mxp, __builtin_bit_cast(u16x8, q));
                    }
                o = __builtin_bit_cast(uint4, mxp);
            } else {
                float mx[8];
                unsigned am[8];                            // first maximum in (kh, kw) scan order, as rtn_maxpool3x3s2_tfsame_fwd_idx records it
#pragma unroll
                for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; am[e] = 255u; }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int i = (2 * wave + dy) * CT_W + 2 * m + dx;
                        const uint4 q = *reinterpret_cast<const uint4*>(cv + i * CV_PX + cg * 16);
                        const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float lo = __uint_as_float(w4[e] << 16), hi = __uint_as_float(w4[e] & 0xffff0000u);
                            if (lo > mx[2 * e]) { mx[2 * e] = lo; am[2 * e] = (unsigned)(dy * 3 + dx); }
                            if (hi > mx[2 * e + 1]) { mx[2 * e + 1] = hi; am[2 * e + 1] = (unsigned)(dy * 3 + dx); }
                        }
                    }
                if (live) {
                    uint2 iq;
                    iq.x = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
                    iq.y = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
                    *reinterpret_cast<uint2*>(p.idx + ((((long long)b * p.H2 + py) * p.W2 + px) * 64 + cg * 8)) = iq;
                }
                o.x = (__float_as_uint(mx[0]) >> 16) | (__float_as_uint(mx[1]) & 0xffff0000u);
                o.y = (__float_as_uint(mx[2]) >> 16) | (__float_as_uint(mx[3]) & 0xffff0000u);
                o.z = (__float_as_uint(mx[4]) >> 16) | (__float_as_uint(mx[5]) & 0xffff0000u);
                o.w = (__float_as_uint(mx[6]) >> 16) | (__float_as_uint(mx[7]) & 0xffff0000u);
            }
            if (!live) o = make_uint4(0u, 0u, 0u, 0u);                     // a pixel past the image: finite operand, never stored
            pooled[h] = o;
            if (live) *reinterpret_cast<uint4*>(p.out + ((((long long)b * p.H2 + py) * p.W2 + px) * 64 + cg * 8)) = o;
        }
        if (A2) {       // a[px][perm rows] = relu(W2a . pooled[px] + b2a): D^T = W2a (A operand) x pooled^T (B operand)
            f32x4 a2[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                a2[f] = *reinterpret_cast<const f32x4*>(w2l + (4 * f + kq) * W2A_ROW + 128);
#pragma unroll
                for (int h = 0; h < 2; ++h) {       // A fragment f, k half h of lane (kq, m): row 16 f + m, input channels 32 h + 8 kq .. + 8
                    const uint4 wf = *reinterpret_cast<const uint4*>(w2l + (16 * f + m) * W2A_ROW + 64 * h + 16 * kq);
                    a2[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, pooled[h]),
                                                                    a2[f], 0, 0, 0);
                }
            }
            // lane (q = kq, column m): fragments 2 s and 2 s + 1 = channels 32 s + 8 q .. + 8 of pixel m
#pragma unroll
            for (int sgrp = 0; sgrp < 2; ++sgrp) {
                const f32x4 lo = a2[2 * sgrp], hi = a2[2 * sgrp + 1];
                uint4 o;
                o.x = (unsigned)to_bf16(fmaxf(lo[0], 0.f)) | ((unsigned)to_bf16(fmaxf(lo[1], 0.f)) << 16);
                o.y = (unsigned)to_bf16(fmaxf(lo[2], 0.f)) | ((unsigned)to_bf16(fmaxf(lo[3], 0.f)) << 16);
                o.z = (unsigned)to_bf16(fmaxf(hi[0], 0.f)) | ((unsigned)to_bf16(fmaxf(hi[1], 0.f)) << 16);
                o.w = (unsigned)to_bf16(fmaxf(hi[2], 0.f)) | ((unsigned)to_bf16(fmaxf(hi[3], 0.f)) << 16);
                if (live) *reinterpret_cast<uint4*>(p.a_out + ((((long long)b * p.H2 + py) * p.W2 + px) * 64 + 32 * sgrp + 8 * kq)) = o;
            }
        }
        __syncthreads();                                  // the conv tile is dead: the next patch may be written
    }
}

}  // namespace

static int stem_launch(rtn_handle_t h, const void* packed, int Hp, int Wp, const void* w_packed, int w_rows, const float* bias, void* out,
                       int B, int H, int W, const void* w2a, const float* b2a, void* a_out, uint8_t* idx) {
    if (!h) return RTN_EINVAL;
    rtn_env_sync();
    if (!packed || !w_packed || !out) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: null pointer");
    if (B < 1 || H < 1 || W < 1 || B > 65535) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: bad extent");
    if (w_rows < 64) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: needs the 64 packed stem filters");
    if (((uintptr_t)packed & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)out & 15) || ((uintptr_t)bias & 3) || (Wp & 1))
        return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: misaligned pointer or odd Wp");
    const bool a2 = a_out != nullptr;
    if (a2 && (!w2a || !b2a || ((uintptr_t)w2a & 15) || ((uintptr_t)a_out & 15) || ((uintptr_t)b2a & 3) || a_out == out))
        return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: a_out needs aligned branch2a filters / shifts and its own buffer");
    if (idx && ((uintptr_t)idx & 7)) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: pool_idx must be 8-byte aligned");
    const int H1 = (H - 1) / 2 + 1, W1 = (W - 1) / 2 + 1;              // ZeroPadding2D(3) + 7x7/2 'valid'
    if (Hp < 2 * (H1 - 1) + 7 || Wp < 2 * (W1 - 1) + 8) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: packed image %dx%d too small for %dx%d", Hp, Wp, H, W);
    const long long bytes = (long long)B * Hp * Wp * 8;
    if (bytes >= 0xFFFFFF00ll) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: packed batch exceeds the 4 GiB buffer-descriptor range");
    const int H2 = (H1 + 1) / 2, W2 = (W1 + 1) / 2;                    // 3x3/2 'same'
    const int pth = (H2 - 1) * 2 + 3 - H1, ptw = (W2 - 1) * 2 + 3 - W1;
    StemParams p;
    memset(&p, 0, sizeof(p));
    p.xp = (const char*)packed; p.wk = (const char*)w_packed; p.bias = bias; p.out = (unsigned short*)out;
    p.w2a = (const char*)w2a; p.b2a = b2a; p.a_out = (unsigned short*)a_out; p.idx = idx;
    p.xp_bytes = (unsigned)bytes;
    p.Hp = Hp; p.Wp = Wp; p.H1 = H1; p.W1 = W1; p.H2 = H2; p.W2 = W2;
    p.pool_pt = (pth > 0 ? pth : 0) / 2; p.pool_pl = (ptw > 0 ? ptw : 0) / 2;
    p.tiles_x = (W2 + PT_W - 1) / PT_W; p.tiles_y = (H2 + PT_H - 1) / PT_H;
    const long long ntiles = (long long)p.tiles_x * p.tiles_y * B;
    if (ntiles > 0x3fffffff) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool: too many tiles");
    p.ntiles = (int)ntiles;
    long long grid = 2ll * (h->num_cus > 0 ? h->num_cus : 256);        // two workgroups per CU (LDS_BYTES each)
    { const int gl = rtn_env_int("RTN_STEM_GRID", 0); if (gl > 0 && gl < grid) grid = gl; }     // tests: many tiles per workgroup on small images
    if (grid > ntiles) grid = ntiles;
#define RTN_STEM_LAUNCH(A2_, IDX_, LDSB)                                                                       \
    do {                                                                                                      \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                         \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                      \
            RTN_HIP(h, hipFuncSetAttribute((const void*)stem_fused_kernel<A2_, IDX_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                                  \
        }                                                                                                     \
        hipLaunchKernelGGL((stem_fused_kernel<A2_, IDX_>), dim3((unsigned)grid), dim3(256), LDSB, h->stream, p); \
    } while (0)
    if (a2) { if (idx) RTN_STEM_LAUNCH(true, true, LDS_BYTES_A2); else RTN_STEM_LAUNCH(true, false, LDS_BYTES_A2); }
    else    { if (idx) RTN_STEM_LAUNCH(false, true, LDS_BYTES); else RTN_STEM_LAUNCH(false, false, LDS_BYTES); }
#undef RTN_STEM_LAUNCH
    RTN_CHECK_LAUNCH(h, "stem_fused_kernel");
    return RTN_OK;
}

extern "C" int rtn_stem_conv_pool(rtn_handle_t h, const void* packed, int Hp, int Wp, const void* w_packed, int w_rows,
                                  const float* bias, void* out, int B, int H, int W) {
    return stem_launch(h, packed, Hp, Wp, w_packed, w_rows, bias, out, B, H, W, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int rtn_stem_conv_pool_branch2a(rtn_handle_t h, const void* packed, int Hp, int Wp, const void* w_packed, int w_rows,
                                           const float* bias, void* out, int B, int H, int W, const void* w2a, const float* b2a, void* a_out,
                                           uint8_t* pool_idx) {
    if (h && !a_out && !pool_idx) return rtn_fail(h, RTN_EINVAL, "stem_conv_pool_branch2a: neither a_out nor pool_idx (use rtn_stem_conv_pool)");
    return stem_launch(h, packed, Hp, Wp, w_packed, w_rows, bias, out, B, H, W, w2a, b2a, a_out, pool_idx);
}
