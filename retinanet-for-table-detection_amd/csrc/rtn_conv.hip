// rtn_conv.hip — implicit-GEMM convolution forward on gfx950 MFMA.
//
// Replaces the Conv2D ops TF executes for the reference graph
// (model/defineModel.py:101-117,155-163,183-203 and the keras_resnet backbone, :376-380).
//
// GEMM view: M = batch*Hout*Wout output pixels, N = Cout, K = KH*KW*Crun.
//   A[m][k]  gathered on the fly from the NHWC activation (zero page for padding taps)
//   B[n][k]  = w[n][k] (K-contiguous, prepared once by the host, BN-folded)
// Tile: 128 (M) x BN (N) x 128 bytes of K per step; 256 threads = 4 waves in 2x2, each wave
// owns 64 x BN/2 as 16x16 MFMA tiles (v_mfma_f32_16x16x32_bf16, or the exact-f32
// v_mfma_f32_16x16x4_f32 for the fp32 parity path — same byte layout in LDS).
// LDS image: [row][128 B], 16-byte chunk index XOR-swizzled with (row & 7): conflict-free
// for the ds_read_b128 fragment reads (checked against the lane-group table of
// MI355X_MICROARCH.md §LDS) and for the 8-lane ds_write_b128 groups.
// Pipeline: register-staged double buffer, one barrier per K step (loads for step k+1 are
// issued before the MFMAs of step k, written to the other LDS buffer after them).
// Epilogue: accumulators -> LDS (f32) -> whole-row 16-byte stores, with bias, residual
// (identity or legacy-TF nearest upsample gather), ReLU / sigmoid fused.
#include "rtn_internal.h"

namespace {

constexpr int BM = 128;
constexpr int NT = 256;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;   // beyond every buffer: the hardware range check returns zeros

__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct KGroup {
    const char* in;
    char* out;
    const char* res;
    unsigned in_bytes;          // size of `in` (buffer descriptor range)
    long long in_img_stride_b;  // bytes
    long long out_img_stride;   // elements
    long long out_off;          // elements
    long long res_img_stride;   // elements
    int in_row_stride_b;        // bytes
    int Hin, Win, Hout, Wout;
    int Hres, Wres, res_ld;
    float rs_h, rs_w;
    int M;
    int tile_begin;
};

struct KParams {
    KGroup g[RTN_MAX_GROUPS];
    const char* w;
    const float* bias;
    unsigned w_bytes;
    int ngroups, ntiles_n;
    int N, Kbytes, nkt;
    int cshift, crun_mask, kw_inv, KW;
    int pix_stride_b, sy, sx, pad_t, pad_l;
    int out_ld, flags, vec_ok;
};

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

template <int ES>
__device__ __forceinline__ void mma_step(f32x4& acc, const uint4& a, const uint4& b) {
    if constexpr (ES == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    } else {
        // exact f32: the lane holds k = 4*kq + e of a 16-wide k group; step e pairs equal k of A and B
        const f32x4 af = __builtin_bit_cast(f32x4, a);
        const f32x4 bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
    }
}

template <int ES, int BN>
__global__ __launch_bounds__(NT, 2) void conv_igemm_kernel(const KParams p) {
    constexpr int ESH = (ES == 2) ? 1 : 2;
    constexpr int WN = BN / 2;
    constexpr int NI = WN / 16;
    constexpr int MI = 4;
    constexpr int NB = BN / 32;                   // B rows staged per thread
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES);
    constexpr int SLD = BN + 4;
    constexpr int EPI_BYTES = BM * SLD * 4;
    constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    // ---- block -> (m tile, n tile), XCD-aware: blocks that share an XCD get a contiguous
    // range of tiles, so the A rows an m tile re-reads for its n tiles hit that XCD's L2.
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mtile_g = wg / p.ntiles_n;
    const int ntile = wg - mtile_g * p.ntiles_n;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < RTN_MAX_GROUPS; ++i)
        if (i < p.ngroups && mtile_g >= p.g[i].tile_begin) gi = i;
    const KGroup& G = p.g[gi];
    const int m0 = (mtile_g - G.tile_begin) * BM;
    const int n0 = ntile * BN;
    const int M = G.M;
    const int Wout = G.Wout;
    const int cells = G.Hout * Wout;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int c = t & 7, r0 = t >> 3;
    const int st_off = r0 * 128 + ((c ^ (r0 & 7)) << 4);   // LDS byte offset of this thread's chunk

    // ---- per-thread A rows (4 output pixels), fixed for the whole K loop.  Taps are fetched with
    // range-checked buffer loads: an out-of-image tap gets OOB_OFFSET and the hardware returns zeros
    // (no branch, no select on pointers — hipcc turns those into a branch + vmcnt(0) per load).
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)G.in, 0, (int)__builtin_amdgcn_readfirstlane((int)G.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w, 0, (int)__builtin_amdgcn_readfirstlane((int)p.w_bytes), 0x00020000);
    unsigned rowbase[4];
    int iy0[4], ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < M) {
            const int b = m / cells;
            const int rem = m - b * cells;
            const int oy = rem / Wout;
            const int ox = rem - oy * Wout;
            iy0[i] = oy * p.sy - p.pad_t;
            ix0[i] = ox * p.sx - p.pad_l;
            // modulo-2^32 arithmetic: only offsets of in-image taps (which fit, host-checked) are ever used
            rowbase[i] = (unsigned)((long long)b * G.in_img_stride_b + (long long)iy0[i] * G.in_row_stride_b +
                                    (long long)ix0[i] * p.pix_stride_b);
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = 0;
            rowbase[i] = 0;
        }
    }
    const unsigned wbase = (unsigned)(n0 + r0) * (unsigned)p.Kbytes + (unsigned)c * 16u;
    const unsigned wstep = 32u * (unsigned)p.Kbytes;
    const int Hin = G.Hin, Win = G.Win, in_row_stride_b = G.in_row_stride_b;

    uint4 ra[4], rb[NB];
#define RTN_LOAD_TILE(KT)                                                                               \
    {                                                                                                   \
        const int kb_ = (KT) * 128 + c * 16;                                                            \
        const int k0_ = kb_ >> ESH;                                                                     \
        const int kpos_ = k0_ >> p.cshift;                                                              \
        const int coff_ = k0_ & p.crun_mask;                                                            \
        const int kh_ = (kpos_ * p.kw_inv) >> 16;                                                       \
        const int kw_ = kpos_ - kh_ * p.KW;                                                             \
        const unsigned delta_ = (unsigned)(kh_ * in_row_stride_b + kw_ * p.pix_stride_b + coff_ * ES);  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                              \
            const int iy_ = iy0[i_] + kh_, ix_ = ix0[i_] + kw_;                                         \
            const bool ok_ = (unsigned)iy_ < (unsigned)Hin && (unsigned)ix_ < (unsigned)Win;            \
            ra[i_] = buffer_load16(in_rsrc, ok_ ? rowbase[i_] + delta_ : OOB_OFFSET);                   \
        }                                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_)                                               \
            rb[i_] = buffer_load16(w_rsrc, wbase + (unsigned)i_ * wstep + (unsigned)(KT) * 128u);       \
    }
#define RTN_STORE_TILE(BUF)                                                                             \
    {                                                                                                   \
        char* A_ = lds + (BUF) * A_BYTES;                                                               \
        char* B_ = lds + 2 * A_BYTES + (BUF) * B_BYTES;                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                \
            *reinterpret_cast<uint4*>(A_ + st_off + i_ * 32 * 128) = ra[i_];                            \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_)                                               \
            *reinterpret_cast<uint4*>(B_ + st_off + i_ * 32 * 128) = rb[i_];                            \
    }

    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, kq = lane >> 4;
    const int rd0 = ((kq ^ (lrow & 7)) << 4);          // chunk kq       (k step 0)
    const int rd1 = (((4 + kq) ^ (lrow & 7)) << 4);    // chunk 4 + kq   (k step 1)
    const int a_row_off = (wm * 64 + lrow) * 128;
    const int b_row_off = (wn * WN + lrow) * 128;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define RTN_COMPUTE_TILE(BUF)                                                                           \
    {                                                                                                   \
        const char* A_ = lds + (BUF) * A_BYTES + a_row_off;                                             \
        const char* B_ = lds + 2 * A_BYTES + (BUF) * B_BYTES + b_row_off;                               \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                           \
            const int rd_ = ks_ ? rd1 : rd0;                                                            \
            uint4 a_[MI], b_[NI];                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                           \
                a_[i_] = *reinterpret_cast<const uint4*>(A_ + i_ * 16 * 128 + rd_);                     \
            _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_)                                           \
                b_[j_] = *reinterpret_cast<const uint4*>(B_ + j_ * 16 * 128 + rd_);                     \
            _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                           \
                _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) mma_step<ES>(acc[i_][j_], a_[i_], b_[j_]); \
        }                                                                                               \
    }

    // ---- main loop
    const int nkt = p.nkt;
    RTN_LOAD_TILE(0);
    RTN_STORE_TILE(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        // even step: compute buffer 0 while the loads of step kt+1 fly, then stage them into buffer 1
        if (kt + 1 < nkt) RTN_LOAD_TILE(kt + 1);
        RTN_COMPUTE_TILE(0);
        if (kt + 1 < nkt) RTN_STORE_TILE(1);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        if (kt + 2 < nkt) RTN_LOAD_TILE(kt + 2);
        RTN_COMPUTE_TILE(1);
        if (kt + 2 < nkt) RTN_STORE_TILE(0);
        __syncthreads();
    }
#undef RTN_LOAD_TILE
#undef RTN_STORE_TILE
#undef RTN_COMPUTE_TILE

    // ---- epilogue: accumulators -> LDS (f32), then whole-row stores
    float* S = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                S[(wm * 64 + i * 16 + kq * 4 + r) * SLD + wn * WN + j * 16 + lrow] = acc[i][j][r];
    __syncthreads();

    constexpr int TPR = BN / 8, RPP = NT / TPR;
    const int ecol = (t % TPR) * 8;
    const int n = n0 + ecol;
    if (n >= p.N) return;
    const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
    const int flags = p.flags;
    const bool out_f32 = (ES == 4) || (flags & RTN_CONV_OUT_F32);
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = 0.f;
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    for (int row = t / TPR; row < BM; row += RPP) {
        const int m = m0 + row;
        if (m >= M) break;
        const int b = m / cells;
        const int cell = m - b * cells;
        const float* s = S + row * SLD + ecol;
        const float4 v0 = *reinterpret_cast<const float4*>(s);
        const float4 v1 = *reinterpret_cast<const float4*>(s + 4);
        float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3],
                      v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
        if (flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE)) {
            long long rpix;
            if (flags & RTN_CONV_RES_UPSAMPLE) {
                // tf.image.resize_images(NEAREST, align_corners=False): src = min(floor(dst*in/out), in-1), f32
                const int oy = cell / Wout, ox = cell - oy * Wout;
                int sy_ = (int)floorf((float)oy * G.rs_h);
                int sx_ = (int)floorf((float)ox * G.rs_w);
                sy_ = sy_ < G.Hres - 1 ? sy_ : G.Hres - 1;
                sx_ = sx_ < G.Wres - 1 ? sx_ : G.Wres - 1;
                rpix = (long long)sy_ * G.Wres + sx_;
            } else {
                rpix = cell;
            }
            const char* rp = G.res + ((long long)b * G.res_img_stride + rpix * G.res_ld + n) * ES;
            if (p.vec_ok) {
                if constexpr (ES == 2) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(rp);
                    const unsigned w4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[2 * j] += __uint_as_float(w4[j] << 16);
                        v[2 * j + 1] += __uint_as_float(w4[j] & 0xffff0000u);
                    }
                } else {
                    const float4 r0v = *reinterpret_cast<const float4*>(rp);
                    const float4 r1v = *reinterpret_cast<const float4*>(rp + 16);
                    v[0] += r0v.x; v[1] += r0v.y; v[2] += r0v.z; v[3] += r0v.w;
                    v[4] += r1v.x; v[5] += r1v.y; v[6] += r1v.z; v[7] += r1v.w;
                }
            } else {
                for (int j = 0; j < nvalid; ++j) {
                    if constexpr (ES == 2)
                        v[j] += __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(rp)[j]) << 16);
                    else
                        v[j] += reinterpret_cast<const float*>(rp)[j];
                }
            }
        }
        if (flags & RTN_CONV_RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        if (flags & RTN_CONV_SIGMOID) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 1.0f / (1.0f + expf(-v[j]));
        }
        const long long oidx = (long long)b * G.out_img_stride + G.out_off + (long long)cell * p.out_ld + n;
        if (out_f32) {
            float* op = reinterpret_cast<float*>(G.out) + oidx;
            if (p.vec_ok) {
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                for (int j = 0; j < nvalid; ++j) op[j] = v[j];
            }
        } else {
            unsigned short* op = reinterpret_cast<unsigned short*>(G.out) + oidx;
            if (p.vec_ok) {
                uint4 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                o.z = pack_bf16x2(v[4], v[5]);
                o.w = pack_bf16x2(v[6], v[7]);
                *reinterpret_cast<uint4*>(op) = o;
            } else {
                for (int j = 0; j < nvalid; ++j) {
                    const __bf16 hb = (__bf16)v[j];
                    op[j] = __builtin_bit_cast(unsigned short, hb);
                }
            }
        }
    }
}

int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace

extern "C" int rtn_conv2d_fwd(rtn_handle_t h, const rtn_conv_desc_t* d) {
    if (!h) return RTN_EINVAL;
    if (!d) return rtn_fail(h, RTN_EINVAL, "conv: null descriptor");
    if (d->dtype != RTN_BF16 && d->dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "conv: bad dtype %d", d->dtype);
    const int es = rtn_dtype_size(d->dtype);
    const bool out_f32 = (d->dtype == RTN_F32) || (d->flags & RTN_CONV_OUT_F32);
    if (d->ngroups < 1 || d->ngroups > RTN_MAX_GROUPS) return rtn_fail(h, RTN_EINVAL, "conv: ngroups %d", d->ngroups);
    if (d->batch < 1 || d->N < 1 || d->KH < 1 || d->KW < 1 || d->sy < 1 || d->sx < 1)
        return rtn_fail(h, RTN_EINVAL, "conv: non-positive dimension");
    const int cshift = ilog2_exact(d->Crun);
    if (cshift < 0 || (d->Crun * es) % 16) return rtn_fail(h, RTN_EINVAL, "conv: Crun %d must be a power of two spanning whole 16-byte chunks", d->Crun);
    const long long Ktot = (long long)d->KH * d->KW * d->Crun;
    if ((Ktot * es) % 128) return rtn_fail(h, RTN_EINVAL, "conv: K=%lld elements is not a multiple of 128 bytes", Ktot);
    if (Ktot * es > (1ll << 30)) return rtn_fail(h, RTN_EINVAL, "conv: K too large");
    if (d->KH * d->KW > 4096) return rtn_fail(h, RTN_EINVAL, "conv: kernel window too large");
    if (d->w_rows % 128 || d->w_rows < d->N) return rtn_fail(h, RTN_EINVAL, "conv: w_rows %d must be a multiple of 128 and >= N %d", d->w_rows, d->N);
    if (!d->w) return rtn_fail(h, RTN_EINVAL, "conv: null weights");
    if (((uintptr_t)d->w & 15) || ((uintptr_t)d->bias & 15)) return rtn_fail(h, RTN_EINVAL, "conv: weights/bias not 16-byte aligned");
    const long long pix_b = (long long)d->pix_stride * es;
    if (pix_b % 16) {
        // narrow pixels (packed stem): every tap start must still be 16-byte aligned
        if (d->KW != 1 || (d->sx * pix_b) % 16 || (d->pad_l * pix_b) % 16)
            return rtn_fail(h, RTN_EINVAL, "conv: pix_stride %d gives unaligned taps", d->pix_stride);
    }
    if ((d->flags & RTN_CONV_RES_SAME) && (d->flags & RTN_CONV_RES_UPSAMPLE))
        return rtn_fail(h, RTN_EINVAL, "conv: both residual modes set");
    const bool has_res = d->flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE);

    KParams p;
    memset(&p, 0, sizeof(p));
    bool vec_ok = (d->N % 8 == 0) && (d->out_ld % 8 == 0);
    long long mtiles = 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        KGroup& g = p.g[i];
        if (!s.in || !s.out) return rtn_fail(h, RTN_EINVAL, "conv: group %d null in/out", i);
        if (((uintptr_t)s.in & 15) || ((uintptr_t)s.out & 15) || ((uintptr_t)s.res & 15))
            return rtn_fail(h, RTN_EINVAL, "conv: group %d pointer not 16-byte aligned", i);
        if (s.Hin < 1 || s.Win < 1 || s.Hout < 1 || s.Wout < 1) return rtn_fail(h, RTN_EINVAL, "conv: group %d empty extent", i);
        if ((s.in_img_stride * es) % 16 || ((long long)s.in_row_stride * es) % 16)
            return rtn_fail(h, RTN_EINVAL, "conv: group %d input strides not 16-byte multiples", i);
        // farthest element a valid tap can read
        const long long in_max = (long long)(d->batch - 1) * s.in_img_stride + (long long)(s.Hin - 1) * s.in_row_stride +
                                 (long long)(s.Win - 1) * d->pix_stride + d->Crun;
        if (in_max > s.in_elems) return rtn_fail(h, RTN_EBOUNDS, "conv: group %d taps reach element %lld of a %lld-element input", i, in_max, (long long)s.in_elems);
        if (s.in_elems * es >= (long long)OOB_OFFSET) return rtn_fail(h, RTN_EINVAL, "conv: group %d input of %lld bytes exceeds the 4 GiB buffer-descriptor range", i, (long long)s.in_elems * es);
        const long long cells = (long long)s.Hout * s.Wout;
        const long long M = cells * d->batch;
        if (M > (1ll << 30)) return rtn_fail(h, RTN_EINVAL, "conv: M too large");
        const long long out_max = (long long)(d->batch - 1) * s.out_img_stride + s.out_off + (cells - 1) * d->out_ld + d->N;
        if (s.out_off < 0 || out_max > s.out_elems) return rtn_fail(h, RTN_EBOUNDS, "conv: group %d writes reach element %lld of a %lld-element output", i, out_max, (long long)s.out_elems);
        if (s.out_img_stride % 8 || s.out_off % 8) vec_ok = false;
        if (has_res) {
            if (!s.res) return rtn_fail(h, RTN_EINVAL, "conv: group %d residual flag without res", i);
            const long long rcells = (d->flags & RTN_CONV_RES_UPSAMPLE) ? (long long)s.Hres * s.Wres : cells;
            if ((d->flags & RTN_CONV_RES_UPSAMPLE) && (s.Hres < 1 || s.Wres < 1)) return rtn_fail(h, RTN_EINVAL, "conv: group %d bad residual extent", i);
            const long long res_max = (long long)(d->batch - 1) * s.res_img_stride + (rcells - 1) * s.res_ld + d->N;
            if (res_max > s.res_elems) return rtn_fail(h, RTN_EBOUNDS, "conv: group %d residual reads reach %lld of %lld", i, res_max, (long long)s.res_elems);
            if (s.res_ld % 8 || s.res_img_stride % 8) vec_ok = false;
        }
        g.in = (const char*)s.in;
        g.out = (char*)s.out;
        g.res = (const char*)s.res;
        g.in_bytes = (unsigned)(s.in_elems * es);
        g.in_img_stride_b = s.in_img_stride * es;
        g.out_img_stride = s.out_img_stride;
        g.out_off = s.out_off;
        g.res_img_stride = s.res_img_stride;
        g.in_row_stride_b = (int)((long long)s.in_row_stride * es);
        g.Hin = s.Hin; g.Win = s.Win; g.Hout = s.Hout; g.Wout = s.Wout;
        g.Hres = s.Hres; g.Wres = s.Wres; g.res_ld = s.res_ld;
        // legacy TF nearest: float32 ratio in/out (SURVEY §8a notes)
        g.rs_h = s.Hout > 0 ? (float)s.Hres / (float)s.Hout : 0.f;
        g.rs_w = s.Wout > 0 ? (float)s.Wres / (float)s.Wout : 0.f;
        g.M = (int)M;
        g.tile_begin = (int)mtiles;
        mtiles += (M + BM - 1) / BM;
    }
    (void)out_f32;
    const int BN = d->N <= 64 ? 64 : 128;
    p.w = (const char*)d->w;
    p.bias = d->bias;
    if ((long long)d->w_rows * Ktot * es >= (long long)OOB_OFFSET) return rtn_fail(h, RTN_EINVAL, "conv: weights exceed the 4 GiB buffer-descriptor range");
    p.w_bytes = (unsigned)((long long)d->w_rows * Ktot * es);
    p.ngroups = d->ngroups;
    p.ntiles_n = (d->N + BN - 1) / BN;
    p.N = d->N;
    p.Kbytes = (int)(Ktot * es);
    p.nkt = p.Kbytes / 128;
    p.cshift = cshift;
    p.crun_mask = d->Crun - 1;
    p.KW = d->KW;
    p.kw_inv = (65536 + d->KW - 1) / d->KW;
    {   // the reciprocal must reproduce kpos / KW for every tap index
        for (int kp = 0; kp < d->KH * d->KW; ++kp)
            if (((kp * p.kw_inv) >> 16) != kp / d->KW) return rtn_fail(h, RTN_EINVAL, "conv: KW %d unsupported", d->KW);
    }
    p.pix_stride_b = (int)pix_b;
    p.sy = d->sy; p.sx = d->sx; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.out_ld = d->out_ld;
    p.flags = d->flags;
    p.vec_ok = vec_ok ? 1 : 0;
    const long long grid = mtiles * p.ntiles_n;
    if (grid < 1 || grid > 0x7fffffffll) return rtn_fail(h, RTN_EINVAL, "conv: grid %lld", grid);

    dim3 gdim((unsigned)grid), bdim(NT);
    if (es == 2) {
        if (BN == 64) hipLaunchKernelGGL((conv_igemm_kernel<2, 64>), gdim, bdim, 0, h->stream, p);
        else          hipLaunchKernelGGL((conv_igemm_kernel<2, 128>), gdim, bdim, 0, h->stream, p);
    } else {
        if (BN == 64) hipLaunchKernelGGL((conv_igemm_kernel<4, 64>), gdim, bdim, 0, h->stream, p);
        else          hipLaunchKernelGGL((conv_igemm_kernel<4, 128>), gdim, bdim, 0, h->stream, p);
    }
    RTN_CHECK_LAUNCH(h, "conv_igemm_kernel");
    return RTN_OK;
}
