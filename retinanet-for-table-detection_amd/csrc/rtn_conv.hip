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
#include <cstdlib>

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

typedef __attribute__((ext_vector_type(4))) int i32x4;

// Buffer descriptor words in SGPRs (raw buffer, stride 0, range = `bytes`); every input is made wave-uniform.
__device__ __forceinline__ i32x4 make_srd(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// LDS-DMA: 64 lanes x 16 B land at LDS byte address `lds_addr` (wave-uniform) + lane*16; lanes whose `voff` is outside
// the descriptor range write zeros.  Issued from asm so that hipcc's waitcnt insertion does not see it (it would
// drain it with vmcnt(0) before the next ds_read and lose the overlap with the MFMAs); the kernel waits for it itself
// (dma_wait_all) before the barrier that publishes the buffer.  M0 is saved/restored inside the statement.
__device__ __forceinline__ void dma16_to_lds(const i32x4& srd, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Wait until at most `stages` x PIECES of this wave's newest vector-memory operations are still in flight (vmcnt
// retires in order): the older stages' LDS-DMA pieces have landed.  `stages` is wave-uniform.
template <int PIECES>
__device__ __forceinline__ void dma_wait_keep(int stages) {
    static_assert(3 * PIECES < 64, "vmcnt is a 6-bit counter");
    if (stages <= 0)      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (stages == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
    else if (stages == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PIECES) : "memory");
    else                  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PIECES) : "memory");
}

struct KGroup {
    const char* in;
    char* out;
    const char* res;
    const char* mask;
    long long mask_img_stride;  // elements
    int mask_ld, out_step, out_pix_w;
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
    int cshift, crun_mask, kw_inv, KW, KH;
    int pix_stride_b, sy, sx, pad_t, pad_l;
    int out_ld, flags, vec_ok;
    int ksplit, kt_per_split;   // split-K (128-row kernel, one group): gridDim.y slices of the K loop add into `scratch`
    float* scratch;             // f32 [ksplit][M][scratch_ld] partial-sum slabs (plain stores; summed in slice order by the
                                // finish kernel: bitwise reproducible, unlike float atomics)
    int scratch_ld;
    int dense_out;   // out/res/mask are plain [M][ld] matrices: the epilogue needs no (image, pixel) split
    int dense_in;    // 1x1, stride 1, no padding on a contiguous NHWC input: row m starts at m * pix_stride
    int nstages;     // 256-row kernel: depth of the LDS staging ring (1..4), sized by the launcher
    int taps_uniform;  // every 128-byte K row lies inside one tap (Crun * element size is a multiple of 128)
    // second input of a K-concatenated 1x1 layer (rtn_conv1x1_dual_fwd): K steps >= nkt1 read `in2` (group 0 only)
    const char* in2;
    unsigned in2_bytes;
    int nkt1;
    long long in2_img_stride_b;
    int in2_row_stride_b, in2_pix_stride_b, in2_step;
    // tail split (256-row kernels, one group): the last `tail_tiles` tiles of a grid that is a few tiles over a whole number
    // of rounds are cut into `tail_slices` K slices each (extra workgroups of the SAME launch write f32 partial sums in
    // register layout to `scratch`); a second, tiny launch of the same kernel (tail_mode 2) sums the slices in order and runs
    // the epilogue.  tail_mode: 0 off, 1 main launch, 2 finish launch.
    int tail_mode, tail_main, tail_tiles, tail_slices, tail_per;   // tail_per: K steps (v2) / (kh,chunk) groups (v3) per slice
    // fp8 (OCP e4m3) operands, halo kernel only: y = acc * acc_scale + bias; stored as bf16, or as fp8 of sat(y * out_scale)
    float acc_scale, out_scale;
    int out_fp8;
    int korder_chunks, korder_kw;   // 256-row kernel: K-step visiting order (see KOrder in the kernel); {nkt, 1} = in order
};

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// fp8 e4m3 x fp8 e4m3, K = 128 per instruction (twice the bf16 rate): a lane supplies 32 bytes of its row per operand.  Which 32 of
// the row's 128 K positions a lane holds is free as long as A and B agree (the instruction pairs equal (lane group, byte) slots),
// so the two 16-byte fragments are the ones the bf16 loop reads for ks = 0 and 1.  Block scales are 2^0 (E8M0 127).
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma_step_fp8(f32x4& acc, const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    // a plain concatenation: built element by element the compiler shuffled dwords through VALU moves (+ s_nop 6) in front of the MFMAs
    typedef int i32x4c __attribute__((ext_vector_type(4)));
    const i32x8 A = __builtin_shufflevector(__builtin_bit_cast(i32x4c, a0), __builtin_bit_cast(i32x4c, a1), 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 B = __builtin_shufflevector(__builtin_bit_cast(i32x4c, b0), __builtin_bit_cast(i32x4c, b1), 0, 1, 2, 3, 4, 5, 6, 7);
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return w;
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


// ---- shared epilogue pieces.  A thread finishes 8 consecutive output channels of NIT output pixels; all residual
// loads of those pixels are issued back-to-back BEFORE any of them is consumed (a load -> wait -> store chain per
// row costs one HBM round trip per row: 25 % of the HBM-bound 1x1 layers' time when it was written that way).
template <int ES>
struct ResVec {
    uint4 q[ES == 2 ? 1 : 2];     // residual
    uint4 k[ES == 2 ? 1 : 2];     // ReLU mask source
};

struct RowRef {
    int b, cell;       // image, dense output pixel index (oy*Wout + ox)
    long long opix;    // pixel index inside the `out` / `res` / `mask` images (differs from cell when out_step > 1)
    bool valid;
};

__device__ __forceinline__ RowRef row_ref(const KParams& p, const KGroup& G, int m, int M, int cells, int Wout) {
    RowRef r;
    r.valid = m < M;
    const int mc = r.valid ? m : M - 1;            // clamped: address math and loads stay unconditional
    if (p.dense_out) {                             // image stride == cells * ld everywhere: index by m alone, no division
        r.b = 0; r.cell = mc; r.opix = mc;
        return r;
    }
    r.b = mc / cells;
    r.cell = mc - r.b * cells;
    if (G.out_step > 1) {
        const int oy = r.cell / Wout, ox = r.cell - oy * Wout;
        r.opix = (long long)oy * G.out_step * G.out_pix_w + (long long)ox * G.out_step;
    } else {
        r.opix = r.cell;
    }
    return r;
}

// residual source pixel: identity, or tf.image.resize_images(NEAREST, align_corners=False):
// src = min(floor(dst * in/out), in-1) with the ratio in float32 (model/layers.py:89-98)
template <int ES>
__device__ __forceinline__ const char* res_ptr(const KParams& p, const KGroup& G, const RowRef& r, int Wout, int n) {
    long long rpix;
    if (p.flags & RTN_CONV_RES_UPSAMPLE) {
        const int oy = r.cell / Wout, ox = r.cell - oy * Wout;
        int sy_ = (int)floorf((float)oy * G.rs_h);
        int sx_ = (int)floorf((float)ox * G.rs_w);
        sy_ = sy_ < G.Hres - 1 ? sy_ : G.Hres - 1;
        sx_ = sx_ < G.Wres - 1 ? sx_ : G.Wres - 1;
        rpix = (long long)sy_ * G.Wres + sx_;
    } else {
        rpix = r.opix;
    }
    return G.res + ((long long)r.b * G.res_img_stride + rpix * G.res_ld + n) * ES;
}

template <int ES>
__device__ __forceinline__ void res_prefetch(const KParams& p, const KGroup& G, const RowRef& r, int Wout, int n, ResVec<ES>& out) {
    const char* rp = res_ptr<ES>(p, G, r, Wout, n);
    out.q[0] = *reinterpret_cast<const uint4*>(rp);
    if constexpr (ES == 4) out.q[1] = *reinterpret_cast<const uint4*>(rp + 16);
}

template <int ES>
__device__ __forceinline__ void mask_prefetch(const KGroup& G, const RowRef& r, int n, ResVec<ES>& out) {
    const char* mp = G.mask + ((long long)r.b * G.mask_img_stride + r.opix * G.mask_ld + n) * ES;
    out.k[0] = *reinterpret_cast<const uint4*>(mp);
    if constexpr (ES == 4) out.k[1] = *reinterpret_cast<const uint4*>(mp + 16);
}

__device__ __forceinline__ void load_bias8(const KParams& p, int n, float (&bv)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = 0.f;
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
}

// zero v[j] where the prefetched mask element is not > 0
template <int ES>
__device__ __forceinline__ void apply_relu_mask(const ResVec<ES>& pre, float (&v)[8]) {
    if constexpr (ES == 2) {
        const unsigned w4[4] = {pre.k[0].x, pre.k[0].y, pre.k[0].z, pre.k[0].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(__uint_as_float(w4[j] << 16) > 0.f)) v[2 * j] = 0.f;
            if (!(__uint_as_float(w4[j] & 0xffff0000u) > 0.f)) v[2 * j + 1] = 0.f;
        }
    } else {
        const unsigned w8[8] = {pre.k[0].x, pre.k[0].y, pre.k[0].z, pre.k[0].w, pre.k[ES == 4 ? 1 : 0].x,
                                pre.k[ES == 4 ? 1 : 0].y, pre.k[ES == 4 ? 1 : 0].z, pre.k[ES == 4 ? 1 : 0].w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!(__uint_as_float(w8[j]) > 0.f)) v[j] = 0.f;
    }
}

// bias -> residual -> ReLU / sigmoid -> bf16 or f32 store at the group's offsets.  `pre` is the prefetched residual
// (used when res_vec), otherwise an unaligned / partial residual is read element-wise here.
template <int ES>
__device__ __forceinline__ void epilogue_finish8(const KParams& p, const KGroup& G, const RowRef& r, int Wout, int n,
                                                 float (&v)[8], const float (&bv)[8], bool res_vec, const ResVec<ES>& pre) {
    if (!r.valid) return;
    const int flags = p.flags;
    const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += bv[j];
    if ((flags & RTN_CONV_RELU_MASK) && (flags & RTN_CONV_MASK_PRE)) apply_relu_mask<ES>(pre, v);
    if (flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE)) {
        if (res_vec) {
            if constexpr (ES == 2) {
                const unsigned w4[4] = {pre.q[0].x, pre.q[0].y, pre.q[0].z, pre.q[0].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] += __uint_as_float(w4[j] << 16);
                    v[2 * j + 1] += __uint_as_float(w4[j] & 0xffff0000u);
                }
            } else {
                v[0] += __uint_as_float(pre.q[0].x); v[1] += __uint_as_float(pre.q[0].y);
                v[2] += __uint_as_float(pre.q[0].z); v[3] += __uint_as_float(pre.q[0].w);
                v[4] += __uint_as_float(pre.q[ES == 4 ? 1 : 0].x); v[5] += __uint_as_float(pre.q[ES == 4 ? 1 : 0].y);
                v[6] += __uint_as_float(pre.q[ES == 4 ? 1 : 0].z); v[7] += __uint_as_float(pre.q[ES == 4 ? 1 : 0].w);
            }
        } else {
            const char* rp = res_ptr<ES>(p, G, r, Wout, n);
            for (int j = 0; j < nvalid; ++j) {
                if constexpr (ES == 2)
                    v[j] += __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(rp)[j]) << 16);
                else
                    v[j] += reinterpret_cast<const float*>(rp)[j];
            }
        }
    }
    if ((flags & RTN_CONV_RELU_MASK) && !(flags & RTN_CONV_MASK_PRE)) apply_relu_mask<ES>(pre, v);   // vec_ok guaranteed by the host
    if (flags & RTN_CONV_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    if (flags & RTN_CONV_SIGMOID) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 1.0f / (1.0f + expf(-v[j]));
    }
    const long long oidx = (long long)r.b * G.out_img_stride + G.out_off + r.opix * p.out_ld + n;
    if constexpr (ES == 2) {
        if (p.out_fp8) {                               // bf16 layer feeding an fp8 layer: e4m3(clamp(y * out_scale)), vec_ok guaranteed
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float q = v[j] * p.out_scale;
                v[j] = q > 448.f ? 448.f : (q < -448.f ? -448.f : q);
            }
            uint2 o;
            o.x = pack_fp8x4(v[0], v[1], v[2], v[3]);
            o.y = pack_fp8x4(v[4], v[5], v[6], v[7]);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(G.out) + oidx) = o;
            return;
        }
    }
    const bool out_f32 = (ES == 4) || (flags & RTN_CONV_OUT_F32);
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

// fp8 layers: bias (+ ReLU) only; the accumulator carries (activation scale x weight scale)^-1, undone by acc_scale.
__device__ __forceinline__ void epilogue_fp8(const KParams& p, const KGroup& G, const RowRef& r, int n, float (&v)[8], const float (&bv)[8]) {
    if (!r.valid) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = v[j] * p.acc_scale + bv[j];
        if (p.flags & RTN_CONV_RELU) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    const long long oidx = (long long)r.b * G.out_img_stride + G.out_off + r.opix * p.out_ld + n;
    if (p.out_fp8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float q = v[j] * p.out_scale;
            v[j] = q > 448.f ? 448.f : (q < -448.f ? -448.f : q);
        }
        uint2 o;
        o.x = pack_fp8x4(v[0], v[1], v[2], v[3]);
        o.y = pack_fp8x4(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(G.out) + oidx) = o;
    } else {
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]);
        o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(G.out) + oidx) = o;
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
        if (m < M && p.dense_in) {
            iy0[i] = 0;
            ix0[i] = 0;
            rowbase[i] = (unsigned)((long long)m * p.pix_stride_b);
        } else if (m < M) {
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
    int kt_lo = 0, kt_hi = nkt;
    if (p.ksplit > 1) {
        kt_lo = blockIdx.y * p.kt_per_split;
        kt_hi = kt_lo + p.kt_per_split < nkt ? kt_lo + p.kt_per_split : nkt;
        if (kt_lo >= kt_hi) return;
    }
    RTN_LOAD_TILE(kt_lo);
    RTN_STORE_TILE(0);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        // compute buffer `cur` while the loads of step kt+1 fly, then stage them into the other buffer
        const bool more = kt + 1 < kt_hi;
        if (more) RTN_LOAD_TILE(kt + 1);
        RTN_COMPUTE_TILE(cur);
        if (more) RTN_STORE_TILE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef RTN_LOAD_TILE
#undef RTN_STORE_TILE
#undef RTN_COMPUTE_TILE

    if (p.ksplit > 1) {      // partial sums of this K slice -> its f32 slab; conv_splitk_finish_kernel reduces + does the epilogue
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nn = n0 + wn * WN + j * 16 + lrow;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = m0 + wm * 64 + i * 16 + kq * 4 + r;
                    if (mm < M && nn < p.scratch_ld)
                        p.scratch[((long long)blockIdx.y * M + mm) * p.scratch_ld + nn] = acc[i][j][r];
                }
            }
        return;
    }
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

    constexpr int TPR = BN / 8, RPP = NT / TPR, NIT = BM / RPP;
    const int ecol = (t % TPR) * 8;
    const int n = n0 + ecol;
    if (n >= p.N) return;
    const bool res_vec = (p.flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE)) && p.vec_ok;
    float bv[8];
    load_bias8(p, n, bv);
    RowRef rows[NIT];
    ResVec<ES> pre[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        rows[it] = row_ref(p, G, m0 + t / TPR + it * RPP, M, cells, Wout);
        if (res_vec) res_prefetch<ES>(p, G, rows[it], Wout, n, pre[it]);
        if (p.flags & RTN_CONV_RELU_MASK) mask_prefetch<ES>(G, rows[it], n, pre[it]);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const float* sp = S + (t / TPR + it * RPP) * SLD + ecol;
        const float4 v0 = *reinterpret_cast<const float4*>(sp);
        const float4 v1 = *reinterpret_cast<const float4*>(sp + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        epilogue_finish8<ES>(p, G, rows[it], Wout, n, v, bv, res_vec, pre[it]);
    }
}

// split-K finish: sum of the K-slice slabs in slice order -> standard epilogue -> out
template <int ES>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const KParams p) {
    const KGroup& G = p.g[0];
    const int M = G.M, Wout = G.Wout, cells = G.Hout * Wout;
    const int ncol = p.scratch_ld / 8;
    const long long total = (long long)M * ncol;
    const bool res_vec = (p.flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE)) && p.vec_ok;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ncol), n = (int)(i - (long long)m * ncol) * 8;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        for (int ks = 0; ks < p.ksplit; ++ks) {                    // fixed slice order
            const float* sp = p.scratch + ((long long)ks * M + m) * p.scratch_ld + n;
            const float4 a0 = *reinterpret_cast<const float4*>(sp);
            const float4 a1 = *reinterpret_cast<const float4*>(sp + 4);
            v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
            v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
        }
        if (n >= p.N) continue;
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        float bv[8];
        load_bias8(p, n, bv);
        const RowRef r = row_ref(p, G, m, M, cells, Wout);
        ResVec<ES> pre;
        if (res_vec) res_prefetch<ES>(p, G, r, Wout, n, pre);
        if (p.flags & RTN_CONV_RELU_MASK) mask_prefetch<ES>(G, r, n, pre);
        epilogue_finish8<ES>(p, G, r, Wout, n, v, bv, res_vec, pre);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Second generation, for the compute-bound layers: 256 (M) x BN (64|128|256) tile, 512 threads = 8 waves in 4x2, wave
// tile 64 x BN/2.  Both operands are staged by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB = 8 rows x 128 B per
// wave-instruction): no staging VGPRs, no ds_write pass.  The LDS image is lane-linear, so the XOR swizzle sits on the
// per-lane SOURCE chunk (same 128-byte line: coalescing unchanged) and on the fragment reads (cdna guide, rule 21).
// Out-of-image taps still get OOB_OFFSET: the range check makes the DMA write zeros.  One barrier per K step: stage
// step k+1 into the other buffer, MFMA step k, vmcnt(0)+barrier.  Epilogue is wave-private (no block barrier): each wave
// moves its 64 x BN/2 accumulators through its own LDS slice, 32 rows at a time, and stores whole rows.
constexpr int BM2 = 256;
constexpr int NT2 = 512;

template <int ES, int BN, bool IL, bool DUAL = false>
__global__ __launch_bounds__(NT2, 2) void conv_igemm2_kernel(const KParams p) {
    constexpr int ESH = (ES == 1) ? 0 : ((ES == 2) ? 1 : 2);
    constexpr int WN = BN / 2;
    constexpr int NI = WN / 16;
    constexpr int MI = 4;
    constexpr int NBI = BN / 64;                 // B row-blocks (8 rows) staged per wave
    constexpr int A_BYTES = BM2 * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + B_BYTES);
    constexpr int SLDW = WN + 4;
    constexpr int EPI_WAVE_BYTES = 32 * SLDW * 4;
    constexpr int EPI_BYTES = 8 * EPI_WAVE_BYTES;
    constexpr int SB = A_BYTES + B_BYTES;        // one stage: A image then B image
    // dynamic LDS: conv2_lds_bytes() stages (1 when the K loop has a single step, else 2), never less than EPI_BYTES
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(STAGE_BYTES == 2 * SB && EPI_BYTES <= 160 * 1024 && STAGE_BYTES <= 160 * 1024, "LDS budget");

    int wg, slice = -1;                               // slice >= 0: this workgroup computes one K slice of a tail tile
    {
        const int nwg = p.tail_mode == 1 ? p.tail_main : gridDim.x, bid = blockIdx.x;
        if (p.tail_mode == 2) {
            wg = p.tail_main + bid;
        } else if (p.tail_mode == 1 && bid >= p.tail_main) {
            const int idx = bid - p.tail_main;
            wg = p.tail_main + idx / p.tail_slices;
            slice = idx - (idx / p.tail_slices) * p.tail_slices;
        } else {
            const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
            wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
    }
    const int mtile_g = wg / p.ntiles_n;
    const int ntile = wg - mtile_g * p.ntiles_n;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < RTN_MAX_GROUPS; ++i)
        if (i < p.ngroups && mtile_g >= p.g[i].tile_begin) gi = i;
    const KGroup& G = p.g[gi];
    const int m0 = (mtile_g - G.tile_begin) * BM2;
    const int n0 = ntile * BN;
    const int M = G.M;
    const int Wout = G.Wout;
    const int cells = G.Hout * Wout;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lr = lane >> 3;                         // row inside an 8-row block
    const int c = (lane & 7) ^ lr;                    // global 16-byte chunk this lane fetches (swizzle on the source)

    const i32x4 in_srd = make_srd(G.in, G.in_bytes);
    const i32x4 w_srd = make_srd(p.w, p.w_bytes);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned rowbase[4];
    int iy0[4], ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (i * 8 + wave) * 8 + lr;
        if (m < M && p.dense_in) {
            iy0[i] = 0;
            ix0[i] = 0;
            rowbase[i] = (unsigned)((long long)m * p.pix_stride_b);
        } else if (m < M) {
            const int b = m / cells;
            const int rem = m - b * cells;
            const int oy = rem / Wout;
            const int ox = rem - oy * Wout;
            iy0[i] = oy * p.sy - p.pad_t;
            ix0[i] = ox * p.sx - p.pad_l;
            rowbase[i] = (unsigned)((long long)b * G.in_img_stride_b + (long long)iy0[i] * G.in_row_stride_b +
                                    (long long)ix0[i] * p.pix_stride_b);
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = 0;
            rowbase[i] = 0;
        }
    }
    // second source (DUAL): pixel (b, oy*step, ox*step) of `in2`, channels walked by the K steps past nkt1
    unsigned rowbase2[DUAL ? 4 : 1];
    i32x4 in2_srd = in_srd;
    if constexpr (DUAL) {
        in2_srd = make_srd(p.in2, p.in2_bytes);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + (i * 8 + wave) * 8 + lr;
            rowbase2[i] = 0;
            if (m < M) {
                const int b = m / cells;
                const int rem = m - b * cells;
                const int oy = rem / Wout;
                const int ox = rem - oy * Wout;
                rowbase2[i] = (unsigned)((long long)b * p.in2_img_stride_b + (long long)oy * p.in2_step * p.in2_row_stride_b +
                                         (long long)ox * p.in2_step * p.in2_pix_stride_b);
            }
        }
    }
    // B rows beyond w_rows fall outside the weight descriptor and read as zeros
    const unsigned wbase = (unsigned)(n0 + wave * 8 + lr) * (unsigned)p.Kbytes + (unsigned)c * 16u;
    const unsigned wstep = 64u * (unsigned)p.Kbytes;
    const int Hin = G.Hin, Win = G.Win, in_row_stride_b = G.in_row_stride_b;

    // One staging piece (1 KiB per wave): D < 4 -> A row-block D of this wave, else B row-block D-4.
#define RTN_TAPS(KT)                                                                                                \
    const bool src2_ = DUAL && (KT) >= p.nkt1;                                                                      \
    const unsigned delta2_ = (unsigned)(((KT) - p.nkt1) * 128 + c * 16);                                            \
    /* KT = position of the step in K (see KOrder).  When a tap spans whole 128-byte chunks (every layer but the packed   \
       stem) the tap of a step is the same for all lanes: decoded from the uniform KT on the scalar unit, the lane's     \
       16-byte piece added at the end; otherwise (two taps per row) the decode stays per lane. */                        \
    const int kb_ = (KT) * 128 + (p.taps_uniform ? 0 : c * 16);                                                     \
    const int k0_ = kb_ >> ESH;                                                                                     \
    const int kpos_ = k0_ >> p.cshift;                                                                              \
    const int coff_ = k0_ & p.crun_mask;                                                                            \
    const int kh_ = (kpos_ * p.kw_inv) >> 16;                                                                       \
    const int kw_ = kpos_ - kh_ * p.KW;                                                                             \
    const unsigned delta_ = (unsigned)(kh_ * in_row_stride_b + kw_ * p.pix_stride_b + coff_ * ES) +                 \
                            (p.taps_uniform ? (unsigned)c * 16u : 0u);                                              \
    const unsigned wk_ = wbase + (unsigned)(KT) * 128u;
#define RTN_DMA(BUF, D)                                                                                             \
    {                                                                                                               \
        if ((D) < 4) {                                                                                              \
            const int iy_ = iy0[(D) & 3] + kh_, ix_ = ix0[(D) & 3] + kw_;                                           \
            const bool ok_ = (unsigned)iy_ < (unsigned)Hin && (unsigned)ix_ < (unsigned)Win;                        \
            if (DUAL && src2_)                                                                                      \
                dma16_to_lds(in2_srd, iy0[(D) & 3] > -(1 << 27) ? rowbase2[DUAL ? ((D) & 3) : 0] + delta2_ : OOB_OFFSET,    \
                             lds_base + (unsigned)((BUF) * SB + wave * 1024 + ((D) & 3) * 8192));                   \
            else                                                                                                    \
            dma16_to_lds(in_srd, ok_ ? rowbase[(D) & 3] + delta_ : OOB_OFFSET,                                      \
                         lds_base + (unsigned)((BUF) * SB + wave * 1024 + ((D) & 3) * 8192));                       \
        } else {                                                                                                    \
            dma16_to_lds(w_srd, wk_ + (unsigned)((D) - 4) * wstep,                                                  \
                         lds_base + (unsigned)((BUF) * SB + A_BYTES + wave * 1024 + ((D) - 4) * 8192));             \
        }                                                                                                           \
    }

    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, kq = lane >> 4;
    const int rd0 = ((kq ^ (lrow & 7)) << 4);
    const int rd1 = (((4 + kq) ^ (lrow & 7)) << 4);
    const int a_row_off = (wm * 64 + lrow) * 128;
    const int b_row_off = (wn * WN + lrow) * 128;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Order in which the K steps are visited.  K is laid out (kh, kw, c); with `p.korder_chunks` = 128-byte chunks per
    // tap the loop walks a kernel row as (chunk, kw) instead of (kw, chunk): the three kw taps of one channel chunk
    // run back to back, and they read the same input lines shifted by one pixel, so two of every three A stagings
    // hit the L1/L2 lines the previous step just pulled.  f32 accumulation order changes with it (deterministic).
    int ko_row = 0, ko_cc = 0, ko_kw = 0;
    const int ko_nchunk = p.korder_chunks, ko_kw_n = p.korder_kw;
#define RTN_KO_POS() ((ko_row * ko_kw_n + ko_kw) * ko_nchunk + ko_cc)
#define RTN_KO_NEXT()                                                                                               \
    do {                                                                                                            \
        if (++ko_kw == ko_kw_n) { ko_kw = 0; if (++ko_cc == ko_nchunk) { ko_cc = 0; ++ko_row; } }                   \
    } while (0)

    constexpr int NJ = NI < 4 ? NI : 4;           // B fragments held at once
    constexpr int ND = 4 + NBI;                   // staging pieces per wave per K step
    constexpr int NS = 2 * (NI / NJ) * 2;         // MFMA groups per K step = slots the pieces are spread over

    // Staging ring of `nst` stages: steps kt+1 .. kt+nst-1 are in flight while step kt is multiplied, so a step
    // costs max(MFMA time, DMA latency / (nst-1)) instead of their sum on the narrow tiles.
    int klo = 0, khi = p.nkt;                         // K steps (in visiting order) of this workgroup
    if (slice >= 0) {
        klo = slice * p.tail_per;
        khi = klo + p.tail_per < p.nkt ? klo + p.tail_per : p.nkt;
    } else if (p.tail_mode == 2) {
        khi = 0;                                      // finish launch: the sums come from the slabs
    }
    const int nkt = khi - klo;
    if (klo > 0) {                                    // position of the iterator at visiting index klo
        const int per_row = ko_kw_n * ko_nchunk;
        ko_row = klo / per_row;
        const int rem_ = klo - ko_row * per_row;
        ko_cc = rem_ / ko_kw_n;
        ko_kw = rem_ - ko_cc * ko_kw_n;
    }
    const int nst = p.nstages;
    for (int s_ = 0; s_ < nst - 1 && s_ < nkt; ++s_) {
        RTN_TAPS(RTN_KO_POS());
#pragma unroll
        for (int d = 0; d < ND; ++d) RTN_DMA(s_, d);
        RTN_KO_NEXT();
    }
    dma_wait_keep<ND>((nst - 1 < nkt ? nst - 1 : nkt) - 1);
    __syncthreads();
    int cur = 0;                     // kt % nst
    int nxt = nst - 1;               // (kt + nst - 1) % nst: the buffer step kt-1 just released
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + nst - 1 < nkt;
        const char* A_ = lds + cur * SB + a_row_off;
        const char* B_ = lds + cur * SB + A_BYTES + b_row_off;
        // first fragments of this step requested before the tap arithmetic and DMA issue of a later step (as in the halo kernel)
        uint4 pa_[MI], pb_[NJ];
#pragma unroll
        for (int i_ = 0; i_ < MI; ++i_) pa_[i_] = *reinterpret_cast<const uint4*>(A_ + i_ * 16 * 128 + rd0);
#pragma unroll
        for (int j_ = 0; j_ < NJ; ++j_) pb_[j_] = *reinterpret_cast<const uint4*>(B_ + j_ * 16 * 128 + rd0);
        RTN_TAPS(RTN_KO_POS());
        RTN_KO_NEXT();
        if (!IL && more) {
#pragma unroll
            for (int d = 0; d < ND; ++d) RTN_DMA(nxt, d);
        }
        if constexpr (ES == 1) {                      // fp8: both 16-byte halves of a row feed ONE K = 128 instruction (see mma_step_fp8)
            uint4 a1_[MI];
#pragma unroll
            for (int i_ = 0; i_ < MI; ++i_) a1_[i_] = *reinterpret_cast<const uint4*>(A_ + i_ * 16 * 128 + rd1);
#pragma unroll
            for (int jh = 0; jh < NI; jh += NJ) {
                uint4 b0_[NJ], b1_[NJ];
#pragma unroll
                for (int j_ = 0; j_ < NJ; ++j_) {
                    b0_[j_] = jh == 0 ? pb_[j_] : *reinterpret_cast<const uint4*>(B_ + (jh + j_) * 16 * 128 + rd0);
                    b1_[j_] = *reinterpret_cast<const uint4*>(B_ + (jh + j_) * 16 * 128 + rd1);
                }
#pragma unroll
                for (int i_ = 0; i_ < MI; ++i_)
#pragma unroll
                    for (int j_ = 0; j_ < NJ; ++j_) mma_step_fp8(acc[i_][jh + j_], pa_[i_], a1_[i_], b0_[j_], b1_[j_]);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rd = ks ? rd1 : rd0;
            uint4 a_[MI];
#pragma unroll
            for (int i_ = 0; i_ < MI; ++i_) a_[i_] = ks == 0 ? pa_[i_] : *reinterpret_cast<const uint4*>(A_ + i_ * 16 * 128 + rd);
#pragma unroll
            for (int jh = 0; jh < NI; jh += NJ) {
                uint4 b_[NJ];
#pragma unroll
                for (int j_ = 0; j_ < NJ; ++j_)
                    b_[j_] = (ks == 0 && jh == 0) ? pb_[j_] : *reinterpret_cast<const uint4*>(B_ + (jh + j_) * 16 * 128 + rd);
#pragma unroll
                for (int ih = 0; ih < 2; ++ih) {
#pragma unroll
                    for (int i_ = 2 * ih; i_ < 2 * ih + 2; ++i_)
#pragma unroll
                        for (int j_ = 0; j_ < NJ; ++j_) mma_step<ES>(acc[i_][jh + j_], a_[i_], b_[j_]);
                    const int slot = (ks * (NI / NJ) + jh / NJ) * 2 + ih;
                    if (IL && more) {     // A/B variant: DMA issues spread between the MFMA groups (measured slower)
#pragma unroll
                        for (int d = slot * ND / NS; d < (slot + 1) * ND / NS; ++d) RTN_DMA(nxt, d);
                    }
                }
            }
        }
        // step kt+1 must have landed; the stages after it (up to the last one issued) may stay in flight
        const int last_issued = kt + nst - 1 < nkt - 1 ? kt + nst - 1 : nkt - 1;
        dma_wait_keep<ND>(last_issued - (kt + 1));
        __syncthreads();                                    // everyone's pieces of kt+1 landed; buffer `cur` is free
        cur = cur + 1 == nst ? 0 : cur + 1;
        nxt = nxt + 1 == nst ? 0 : nxt + 1;
    }
#undef RTN_TAPS
#undef RTN_DMA
#undef RTN_KO_POS
#undef RTN_KO_NEXT

    if (p.tail_mode != 0 && wg >= p.tail_main) {      // tail tile: partial sums travel through f32 slabs in register layout
        f32x4* slab = reinterpret_cast<f32x4*>(p.scratch);
        const long long tl = (long long)(wg - p.tail_main) * p.tail_slices;
        if (slice >= 0) {
            f32x4* dst = slab + (((tl + slice) * 8 + wave) * (MI * NI)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) dst[(i * NI + j) * 64] = acc[i][j];
            return;
        }
        for (int sl = 0; sl < p.tail_slices; ++sl) {  // finish launch: slices in order (bitwise reproducible)
            const f32x4* src = slab + (((tl + sl) * 8 + wave) * (MI * NI)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const f32x4 v = src[(i * NI + j) * 64];
                    acc[i][j][0] += v[0]; acc[i][j][1] += v[1]; acc[i][j][2] += v[2]; acc[i][j][3] += v[3];
                }
        }
    }

    // ---- wave-private epilogue: 2 halves of 32 rows through this wave's LDS slice
    float* S = reinterpret_cast<float*>(lds + wave * EPI_WAVE_BYTES);
    constexpr int TPRW = WN / 8, RPPW = 64 / TPRW, NITW = 32 / RPPW;
    const int ecol = (lane % TPRW) * 8;
    const int n = n0 + wn * WN + ecol;
    const bool ncol_ok = n < p.N;
    const bool res_vec = (p.flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE)) && p.vec_ok;
    float bv[8];
    load_bias8(p, ncol_ok ? n : 0, bv);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        RowRef rows[NITW];
        ResVec<ES> pre[NITW];
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            rows[it] = row_ref(p, G, m0 + wm * 64 + hh * 32 + lane / TPRW + it * RPPW, M, cells, Wout);
            rows[it].valid = rows[it].valid && ncol_ok;
            if (res_vec) res_prefetch<ES>(p, G, rows[it], Wout, ncol_ok ? n : 0, pre[it]);
            if (p.flags & RTN_CONV_RELU_MASK) mask_prefetch<ES>(G, rows[it], ncol_ok ? n : 0, pre[it]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(i * 16 + kq * 4 + r) * SLDW + j * 16 + lrow] = acc[hh * 2 + i][j][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's LDS writes have landed
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            const float* sp = S + (lane / TPRW + it * RPPW) * SLDW + ecol;
            const float4 v0 = *reinterpret_cast<const float4*>(sp);
            const float4 v1 = *reinterpret_cast<const float4*>(sp + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (ES == 1) epilogue_fp8(p, G, rows[it], n, v, bv);
            else epilogue_finish8<ES>(p, G, rows[it], Wout, n, v, bv, res_vec, pre[it]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);      // reads done before the next half overwrites the slice
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Third generation, for stride-1 'same' KHxKW convolutions (every 3x3 layer of the network): the 256-row kernel with the
// input taps of one kernel row sharing ONE staged image.  For a fixed (kh, channel chunk) the KW taps read the same input
// lines shifted by one pixel, so the kernel stages a HALO of 256 consecutive input pixels (flat NHWC index) once and the
// tap kw multiplies LDS rows [kw, kw+256): A traffic L2->LDS drops by KW (the 64-wide tiles were bound by exactly that
// traffic: 40 KB per K step, now 18.7 KB).  The tile advances by TM = 256-(KW-1) output pixels so the halo is exactly 32
// staging pieces (its last KW-1 MFMA rows are computed and dropped).
//   * rows whose vertical tap leaves the image are zero-filled at staging time (range-checked DMA, as before); the test
//     is on the halo pixel's own image row iy0: 0 <= iy0 + kh - pad_t < H, which is the consumer's test because every
//     consumer of that halo row that is not edge-masked sits on the same image row;
//   * the horizontal edge (ox + kw - pad_l outside [0, W)) depends on the consumer: the A fragment of such a row is
//     zeroed in registers (4 v_cndmask per fragment on the kw != centre steps).
// K is walked (kh, chunk, kw) - the weights keep their (kh, kw, c) layout, the B tile of a step is column block
// ((kh*KW + kw)*chunks + chunk).
template <int ES, int BN>
__global__ __launch_bounds__(NT2, 2) void conv_igemm3_kernel(const KParams p) {
    constexpr int WN = BN / 2;
    constexpr int NI = WN / 16;
    constexpr int MI = 4;
    constexpr int NBI = BN / 64;                 // B row-blocks (8 rows) staged per wave
    constexpr int A_BYTES = BM2 * 128, B_BYTES = BN * 128;
    constexpr int B_BASE = 2 * A_BYTES;
    constexpr int SLDW = WN + 4;
    constexpr int EPI_WAVE_BYTES = 32 * SLDW * 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 A halos, 2 B tiles (>= the epilogue slices)

    int wg, slice = -1;                               // slice >= 0: this workgroup computes one K slice of a tail tile
    {
        const int nwg = p.tail_mode == 1 ? p.tail_main : gridDim.x, bid = blockIdx.x;
        if (p.tail_mode == 2) {
            wg = p.tail_main + bid;
        } else if (p.tail_mode == 1 && bid >= p.tail_main) {
            const int idx = bid - p.tail_main;
            wg = p.tail_main + idx / p.tail_slices;
            slice = idx - (idx / p.tail_slices) * p.tail_slices;
        } else {
            const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
            wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
    }
    const int mtile_g = wg / p.ntiles_n;
    const int ntile = wg - mtile_g * p.ntiles_n;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < RTN_MAX_GROUPS; ++i)
        if (i < p.ngroups && mtile_g >= p.g[i].tile_begin) gi = i;
    const KGroup& G = p.g[gi];
    const int KWn = p.KW;
    const int TM = BM2 - (KWn - 1);
    const int m0 = (mtile_g - G.tile_begin) * TM;
    const int n0 = ntile * BN;
    const int M = G.M;
    const int Wout = G.Wout;
    const int cells = G.Hout * Wout;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lr = lane >> 3;
    const int c = (lane & 7) ^ lr;                    // source chunk of this lane (swizzle on the source)

    const i32x4 in_srd = make_srd(G.in, G.in_bytes);
    const i32x4 w_srd = make_srd(p.w, p.w_bytes);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- halo rows staged by this lane: piece i of the wave is row-block (i*8 + wave), row h = block*8 + lr
    unsigned hbase[4];      // byte offset of the halo pixel (kh = pad_t row), + chunk
    int hiy[4];             // its image row, or far negative when the pixel does not exist
    const int Hin = G.Hin;
    const long long total_px = (long long)(M);        // B * cells (Hout == Hin, Wout == Win)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = (i * 8 + wave) * 8 + lr;
        const long long f = (long long)m0 + h - p.pad_l;
        if (f >= 0 && f < total_px) {
            const int fi = (int)f;
            const int b = fi / cells;
            const int rem = fi - b * cells;
            hiy[i] = rem / Wout;
            hbase[i] = (unsigned)(f * p.pix_stride_b) + (unsigned)c * 16u;
        } else {
            hiy[i] = -(1 << 28);
            hbase[i] = 0;
        }
    }
    const unsigned wbase = (unsigned)(n0 + wave * 8 + lr) * (unsigned)p.Kbytes + (unsigned)c * 16u;
    const unsigned wstep = 64u * (unsigned)p.Kbytes;
    const int in_row_stride_b = G.in_row_stride_b;
    const int nchunk = p.korder_chunks;               // 128-byte chunks per tap

    // ---- fragment rows of this lane and their horizontal edge masks: bit (i*4 + kw) set = tap kw leaves the image
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, kq = lane >> 4;
    unsigned emask = 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * 64 + i * 16 + lrow;
        const int mc = m < M ? m : M - 1;
        const int rem = mc % cells;
        const int ox = rem % Wout;
        for (int kw = 0; kw < KWn; ++kw)
            if ((unsigned)(ox + kw - p.pad_l) >= (unsigned)Wout) emask |= 1u << (i * 4 + kw);
    }
    const int a_row = wm * 64 + lrow;
    const int b_row_off = (wn * WN + lrow) * 128;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NJ = NI < 4 ? NI : 4;

    // group = (kh, chunk); step = (group, kw)
#define RTN_A_STAGE(ABUF, KH_, CC_)                                                                                 \
    {                                                                                                               \
        const int dy_ = (KH_) - p.pad_t;                                                                            \
        const unsigned delta_ = (unsigned)(dy_ * in_row_stride_b + (CC_) * 128);                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
            const bool ok_ = (unsigned)(hiy[i_] + dy_) < (unsigned)Hin;                                             \
            dma16_to_lds(in_srd, ok_ ? hbase[i_] + delta_ : OOB_OFFSET,                                             \
                         lds_base + (unsigned)((ABUF) * A_BYTES + wave * 1024 + i_ * 8192));                        \
        }                                                                                                           \
    }
#define RTN_B_STAGE(BBUF, KH_, CC_, KW_)                                                                            \
    {                                                                                                               \
        const unsigned wk_ = wbase + (unsigned)((((KH_) * KWn + (KW_)) * nchunk + (CC_)) * 128);                    \
        _Pragma("unroll") for (int d_ = 0; d_ < NBI; ++d_)                                                          \
            dma16_to_lds(w_srd, wk_ + (unsigned)d_ * wstep,                                                         \
                         lds_base + (unsigned)(B_BASE + (BBUF) * B_BYTES + wave * 1024 + d_ * 8192));               \
    }

    // B ring of `nbst` stages (2 or 3): the B tile of step kt + nbst - 1 is staged while step kt is multiplied.
    // (kh, chunk) groups of this workgroup: all of them, or one slice of a tail tile, or none (finish launch)
    int g_lo = 0, g_hi = p.KH * nchunk;
    if (slice >= 0) {
        g_lo = slice * p.tail_per;
        g_hi = g_lo + p.tail_per < g_hi ? g_lo + p.tail_per : g_hi;
    } else if (p.tail_mode == 2) {
        g_hi = 0;
    }
    const int nkt = (g_hi - g_lo) * KWn;
    const int nbst = p.nstages;
    int kh = g_lo / nchunk, cc = g_lo - (g_lo / nchunk) * nchunk, kw = 0;   // the step being multiplied
    int nkh = kh, ncc = cc, nkw = 0;                  // the step whose B tile is staged next
    int gcur = g_lo;                                  // group of the step being multiplied
    int abuf = 0, bcur = 0, bnxt = 0;
    if (nkt > 0) RTN_A_STAGE(0, kh, cc);
    for (int s_ = 0; s_ < nbst - 1 && s_ < nkt; ++s_) {
        RTN_B_STAGE(bnxt, nkh, ncc, nkw);
        if (++nkw == KWn) { nkw = 0; if (++ncc == nchunk) { ncc = 0; ++nkh; } }
        bnxt = bnxt + 1 == nbst ? 0 : bnxt + 1;
    }
    if (nbst == 3 && nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NBI) : "memory");     // halo + B(0) landed
    else                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool a_prev = false;
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + nbst - 1 < nkt;
        bool a_now = false;
        // the step's first fragments are requested BEFORE the staging DMA of later steps is issued: their LDS latency then hides
        // behind the staging address arithmetic instead of standing in front of the first MFMA
        uint4 pa_[MI], pb_[NJ];
        if constexpr (ES == 2) {                      // (the fp8 instance spills when it holds them: measured 0.139 -> 0.211 ms)
            const int rr0 = a_row + kw;
            const char* A0 = lds + abuf * A_BYTES + rr0 * 128;
            const char* B0 = lds + B_BASE + bcur * B_BYTES + b_row_off;
            const int rdA = (kq ^ (rr0 & 7)) << 4, rdB = (kq ^ (lrow & 7)) << 4;
#pragma unroll
            for (int i_ = 0; i_ < MI; ++i_) pa_[i_] = *reinterpret_cast<const uint4*>(A0 + i_ * 16 * 128 + rdA);
#pragma unroll
            for (int j_ = 0; j_ < NJ; ++j_) pb_[j_] = *reinterpret_cast<const uint4*>(B0 + j_ * 16 * 128 + rdB);
        }
        if (more) {
            RTN_B_STAGE(bnxt, nkh, ncc, nkw);
            if (++nkw == KWn) { nkw = 0; if (++ncc == nchunk) { ncc = 0; ++nkh; } }
            bnxt = bnxt + 1 == nbst ? 0 : bnxt + 1;
        }
        if (kw == 0) {                                // first step of a group: stage the NEXT group's halo behind the B tile
            int gkh = kh, gcc = cc + 1;
            if (gcc == nchunk) { gcc = 0; ++gkh; }
            if (gcur + 1 < g_hi) { RTN_A_STAGE(abuf ^ 1, gkh, gcc); a_now = true; }
        }
        const int rr = a_row + kw;                    // LDS row of this lane's first fragment row for tap kw
        const int swz = rr & 7;
        const char* A_ = lds + abuf * A_BYTES + rr * 128;
        const char* B_ = lds + B_BASE + bcur * B_BYTES + b_row_off;
        const unsigned em = emask >> kw;
        if constexpr (ES == 1) {                      // fp8: the row's two fragment halves feed ONE K = 128 instruction
            const int rdA0 = (kq ^ swz) << 4, rdA1 = ((4 + kq) ^ swz) << 4;
            const int rdB0 = (kq ^ (lrow & 7)) << 4, rdB1 = ((4 + kq) ^ (lrow & 7)) << 4;
            uint4 a0_[MI], a1_[MI];
#pragma unroll
            for (int i_ = 0; i_ < MI; ++i_) {
                a0_[i_] = *reinterpret_cast<const uint4*>(A_ + i_ * 16 * 128 + rdA0);
                a1_[i_] = *reinterpret_cast<const uint4*>(A_ + i_ * 16 * 128 + rdA1);
                if ((em >> (i_ * 4)) & 1u) { a0_[i_] = make_uint4(0u, 0u, 0u, 0u); a1_[i_] = a0_[i_]; }
            }
#pragma unroll
            for (int jh = 0; jh < NI; jh += NJ) {
                uint4 b0_[NJ], b1_[NJ];
#pragma unroll
                for (int j_ = 0; j_ < NJ; ++j_) {
                    b0_[j_] = *reinterpret_cast<const uint4*>(B_ + (jh + j_) * 16 * 128 + rdB0);
                    b1_[j_] = *reinterpret_cast<const uint4*>(B_ + (jh + j_) * 16 * 128 + rdB1);
                }
#pragma unroll
                for (int i_ = 0; i_ < MI; ++i_)
#pragma unroll
                    for (int j_ = 0; j_ < NJ; ++j_) mma_step_fp8(acc[i_][jh + j_], a0_[i_], a1_[i_], b0_[j_], b1_[j_]);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rdA = ((ks * 4 + kq) ^ swz) << 4;
            const int rdB = ((ks * 4 + kq) ^ (lrow & 7)) << 4;
            uint4 a_[MI];
#pragma unroll
            for (int i_ = 0; i_ < MI; ++i_) {
                a_[i_] = (ES == 2 && ks == 0) ? pa_[i_] : *reinterpret_cast<const uint4*>(A_ + i_ * 16 * 128 + rdA);
                if ((em >> (i_ * 4)) & 1u) a_[i_] = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int jh = 0; jh < NI; jh += NJ) {
                uint4 b_[NJ];
#pragma unroll
                for (int j_ = 0; j_ < NJ; ++j_)
                    b_[j_] = (ES == 2 && ks == 0 && jh == 0) ? pb_[j_] : *reinterpret_cast<const uint4*>(B_ + (jh + j_) * 16 * 128 + rdB);
#pragma unroll
                for (int i_ = 0; i_ < MI; ++i_)
#pragma unroll
                    for (int j_ = 0; j_ < NJ; ++j_) mma_step<ES>(acc[i_][jh + j_], a_[i_], b_[j_]);
            }
        }
        }
        // The B tile of step kt+1 must have landed.  Operations issued after it may stay in flight (vmcnt retires in
        // order): with a 3-stage ring the B tile staged this step, plus a halo (4 pieces) staged this or the previous step.
        // A halo is complete one step before its group starts (KW >= 3 with the 3-stage ring: the host guarantees it).
        if (nbst == 3 && more) {
            if (a_now || a_prev) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NBI + 4) : "memory");
            else                 asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NBI) : "memory");
        } else if (nbst == 2 && a_now) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        a_prev = a_now;
        __syncthreads();
        bcur = bcur + 1 == nbst ? 0 : bcur + 1;
        if (++kw == KWn) { kw = 0; abuf ^= 1; ++gcur; if (++cc == nchunk) { cc = 0; ++kh; } }
    }
#undef RTN_A_STAGE
#undef RTN_B_STAGE

    if (p.tail_mode != 0 && wg >= p.tail_main) {      // tail tile: partial sums travel through f32 slabs in register layout
        f32x4* slab = reinterpret_cast<f32x4*>(p.scratch);
        const long long tl = (long long)(wg - p.tail_main) * p.tail_slices;
        if (slice >= 0) {
            f32x4* dst = slab + (((tl + slice) * 8 + wave) * (MI * NI)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) dst[(i * NI + j) * 64] = acc[i][j];
            return;
        }
        for (int sl = 0; sl < p.tail_slices; ++sl) {  // finish launch: slices in order (bitwise reproducible)
            const f32x4* src = slab + (((tl + sl) * 8 + wave) * (MI * NI)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const f32x4 v = src[(i * NI + j) * 64];
                    acc[i][j][0] += v[0]; acc[i][j][1] += v[1]; acc[i][j][2] += v[2]; acc[i][j][3] += v[3];
                }
        }
    }

    // ---- wave-private epilogue (rows beyond TM belong to the next tile)
    float* S = reinterpret_cast<float*>(lds + wave * EPI_WAVE_BYTES);
    constexpr int TPRW = WN / 8, RPPW = 64 / TPRW, NITW = 32 / RPPW;
    const int ecol = (lane % TPRW) * 8;
    const int n = n0 + wn * WN + ecol;
    const bool ncol_ok = n < p.N;
    const bool res_vec = (p.flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE)) && p.vec_ok;
    float bv[8];
    load_bias8(p, ncol_ok ? n : 0, bv);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        RowRef rows[NITW];
        ResVec<ES> pre[NITW];
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            const int rloc = wm * 64 + hh * 32 + lane / TPRW + it * RPPW;
            rows[it] = row_ref(p, G, m0 + rloc, M, cells, Wout);
            rows[it].valid = rows[it].valid && ncol_ok && rloc < TM;
            if (res_vec) res_prefetch<ES>(p, G, rows[it], Wout, ncol_ok ? n : 0, pre[it]);
            if (p.flags & RTN_CONV_RELU_MASK) mask_prefetch<ES>(G, rows[it], ncol_ok ? n : 0, pre[it]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(i * 16 + kq * 4 + r) * SLDW + j * 16 + lrow] = acc[hh * 2 + i][j][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int it = 0; it < NITW; ++it) {
            const float* sp = S + (lane / TPRW + it * RPPW) * SLDW + ecol;
            const float4 v0 = *reinterpret_cast<const float4*>(sp);
            const float4 v1 = *reinterpret_cast<const float4*>(sp + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (ES == 1) epilogue_fp8(p, G, rows[it], n, v, bv);
            else epilogue_finish8<ES>(p, G, rows[it], Wout, n, v, bv, res_vec, pre[it]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
}

// Tuning knobs, read on every call so one process can A/B them: RTN_CONV_IMPL=1|2 forces a kernel generation
// (unset/0 = heuristic).
int rtn_conv_impl_override() {
    const int v = rtn_env_int("RTN_CONV_IMPL", 0);
    return (v >= 1 && v <= 6) ? v : 0;                 // 4 / 5 / 6 = the persistent kernels (rtn_conv_halo8.hip / rtn_conv_gemm8.hip / rtn_conv_halon.hip) where they apply
}

int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace

// Integer tuning knobs from the environment.  They stay switchable inside one process (tests and tools/ab_*.py flip them between
// two launches), but a launch no longer pays ~30 getenv() string scans: the parsed values are cached per thread, and the cache is
// dropped when the process environment has changed — detected by a checksum over the `environ` pointer array (setenv / putenv
// replace the entry's pointer), one pass over ~50 pointers per rtn_env_sync() call at the top of an entry point.
extern char** environ;
namespace {
struct EnvKnob { const char* name; int dflt; int value; };
thread_local EnvKnob g_env_knobs[96];
thread_local int g_env_nknobs = 0;
thread_local unsigned long long g_env_stamp = 0;
}
void rtn_env_sync() {
    unsigned long long st = 0x9e3779b97f4a7c15ull;
    if (environ)
        for (char** e = environ; *e; ++e) st = (st ^ (unsigned long long)(uintptr_t)*e) * 0x100000001b3ull;
    if (st != g_env_stamp) { g_env_stamp = st; g_env_nknobs = 0; }
}
int rtn_env_int(const char* name, int dflt) {
    for (int i = 0; i < g_env_nknobs; ++i)
        if (g_env_knobs[i].name == name && g_env_knobs[i].dflt == dflt) return g_env_knobs[i].value;     // string literals: pointer identity
    const char* e = getenv(name);
    const int v = (e && *e) ? atoi(e) : dflt;
    if (g_env_nknobs < 96) g_env_knobs[g_env_nknobs++] = EnvKnob{name, dflt, v};
    return v;
}


// `query` != nullptr: no launch; *query = bytes of d->workspace this launch can use (the K-split paths below).
static constexpr long long kMaxConvWorkspace = 256ll << 20;
static int conv_launch(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_src2_t* s2 = nullptr, const rtn_conv_fp8_t* q8 = nullptr,
                       float out8_scale = 0.f, size_t* query = nullptr) {
    if (query) *query = 0;
    if (!h) return RTN_EINVAL;
    if (!d) return rtn_fail(h, RTN_EINVAL, "conv: null descriptor");
    rtn_env_sync();
    if (d->dtype != RTN_BF16 && d->dtype != RTN_F32 && !(d->dtype == RTN_FP8 && q8)) return rtn_fail(h, RTN_EINVAL, "conv: bad dtype %d", d->dtype);
    if (q8) {
        if (d->dtype != RTN_FP8) return rtn_fail(h, RTN_EINVAL, "conv fp8: the descriptor's dtype must be RTN_FP8");
        if (s2) return rtn_fail(h, RTN_EINVAL, "conv fp8: no second source");
        if (d->flags & ~RTN_CONV_RELU) return rtn_fail(h, RTN_EINVAL, "conv fp8: only bias and ReLU epilogues (flags 0x%x)", d->flags);
        if (q8->out_dtype != RTN_FP8 && q8->out_dtype != RTN_BF16) return rtn_fail(h, RTN_EINVAL, "conv fp8: output must be fp8 or bf16");
        if (!(q8->acc_scale > 0.f) || !(q8->out_scale > 0.f)) return rtn_fail(h, RTN_EINVAL, "conv fp8: scales must be positive");
        if (d->N % 8 || d->out_ld % 8) return rtn_fail(h, RTN_EINVAL, "conv fp8: N and out_ld must be multiples of 8");
    }
    const int es = rtn_dtype_size(d->dtype);
    const bool out_f32 = (d->dtype == RTN_F32) || (d->flags & RTN_CONV_OUT_F32);
    if (d->ngroups < 1 || d->ngroups > RTN_MAX_GROUPS) return rtn_fail(h, RTN_EINVAL, "conv: ngroups %d", d->ngroups);
    if (d->batch < 1 || d->N < 1 || d->KH < 1 || d->KW < 1 || d->sy < 1 || d->sx < 1)
        return rtn_fail(h, RTN_EINVAL, "conv: non-positive dimension");
    const int cshift = ilog2_exact(d->Crun);
    if (cshift < 0 || (d->Crun * es) % 16) return rtn_fail(h, RTN_EINVAL, "conv: Crun %d must be a power of two spanning whole 16-byte chunks", d->Crun);
    long long Ktot = (long long)d->KH * d->KW * d->Crun;
    const long long K1 = Ktot;
    if (s2) {       // K-concatenated second source: both are 1x1 taps, the second one possibly strided
        if (d->KH != 1 || d->KW != 1 || d->sy != 1 || d->sx != 1 || d->pad_t != 0 || d->pad_l != 0 || d->ngroups != 1)
            return rtn_fail(h, RTN_EINVAL, "conv dual: the first source must be a single-group 1x1 stride-1 layer");
        if (!s2->in || ((uintptr_t)s2->in & 15)) return rtn_fail(h, RTN_EINVAL, "conv dual: null / misaligned second source");
        if (s2->C < 1 || (s2->C * es) % 128 || (K1 * es) % 128) return rtn_fail(h, RTN_EINVAL, "conv dual: both channel counts must span whole 128-byte chunks");
        if (s2->step < 1 || s2->Hin < 1 || s2->Win < 1 || (s2->in_img_stride * es) % 16 || ((long long)s2->in_row_stride * es) % 16 || ((long long)s2->pix_stride * es) % 16)
            return rtn_fail(h, RTN_EINVAL, "conv dual: bad second-source geometry");
        const rtn_conv_group_t& g0 = d->g[0];
        if ((long long)(g0.Hout - 1) * s2->step >= s2->Hin || (long long)(g0.Wout - 1) * s2->step >= s2->Win)
            return rtn_fail(h, RTN_EBOUNDS, "conv dual: output %dx%d at step %d leaves the %dx%d second source", g0.Hout, g0.Wout, s2->step, s2->Hin, s2->Win);
        const long long in2_max = (long long)(d->batch - 1) * s2->in_img_stride + (long long)(g0.Hout - 1) * s2->step * s2->in_row_stride +
                                  (long long)(g0.Wout - 1) * s2->step * s2->pix_stride + s2->C;
        if (in2_max > s2->in_elems) return rtn_fail(h, RTN_EBOUNDS, "conv dual: second source reads reach %lld of %lld", in2_max, (long long)s2->in_elems);
        if (s2->in_elems * es >= (long long)OOB_OFFSET) return rtn_fail(h, RTN_EINVAL, "conv dual: second source exceeds the 4 GiB descriptor range");
        Ktot = K1 + s2->C;
    }
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
    // caller-owned scratch of the K-split paths (never allocated here).  Its first RTN_CONV_SYNC_BYTES are the sync block of the
    // in-launch reductions (flags; zero before and after every launch, see rtn_conv_workspace_init): every slab starts behind it.
    const bool ws_ok = !query && d->workspace && !((uintptr_t)d->workspace & 15) && d->workspace_bytes > RTN_CONV_SYNC_BYTES;
    unsigned* const ws_sync = ws_ok ? (unsigned*)d->workspace : nullptr;
    float* const ws_ptr = ws_ok ? (float*)((char*)d->workspace + RTN_CONV_SYNC_BYTES) : nullptr;
    const long long ws_cap = query ? kMaxConvWorkspace : (ws_ptr ? (long long)d->workspace_bytes - RTN_CONV_SYNC_BYTES : 0);
    if (!query) { h->last_conv_streamk = 0; h->last_conv_tile = 0; }

    KParams p;
    memset(&p, 0, sizeof(p));
    bool vec_ok = (d->N % 8 == 0) && (d->out_ld % 8 == 0);
    long long mtiles = 0;
    // ---- kernel generation: 2 = 256-row LDS-DMA kernel (compute-bound layers), 1 = 128-row register-staged kernel.
    long long Mtot = 0;
    for (int i = 0; i < d->ngroups; ++i) Mtot += (long long)d->g[i].Hout * d->g[i].Wout * d->batch;
    int impl = rtn_conv_impl_override();
    int bn2 = d->N <= 64 ? 64 : (d->N <= 128 ? 128 : 256);   // widest 256-row tile that N fills
    {
        // Tile choice, fitted to the same-process sweep in profiles/r1_conv_tile_sweep.txt (tools/ab_conv.py):
        //  * the 256-row LDS-DMA kernel runs one workgroup per CU, so its time is (grid rounded up to whole
        //    rounds of CUs) x (time of one tile); one 256x{64,128,256} tile costs 1 : 2.02 : 3.44 on a long K loop.
        //    Wide tiles therefore only pay when the grid stays many rounds deep (the head towers);
        //  * on short K loops the epilogue and HBM dominate and the narrow tile (more workgroups in flight) wins;
        //  * the 128-row register-staged kernel (3 workgroups/CU) keeps the <=64-channel 1x1 layers with K >= 256
        //    (res2*_branch2a: pure streaming, 0.074 vs 0.080 ms); the grouped head outputs moved to the 256-row kernel
        //    once it had dynamic LDS (0.113 -> 0.087 ms).
        const long long mt2 = (Mtot + BM2 - 1) / BM2;
        const int cus = h->num_cus > 0 ? h->num_cus : 256;
        if (Ktot * es >= 2048 && d->N > 64) {
            const int cand[3] = {64, 128, 256};
            const double wgt[3] = {1.0, 2.02, 3.44};
            const int widest = bn2;
            double best = 0;
            for (int c = 0; c < 3 && cand[c] <= widest; ++c) {
                const long long grid = mt2 * ((d->N + cand[c] - 1) / cand[c]);
                const double cost = (double)((grid + cus - 1) / cus) * wgt[c];
                if (c == 0 || cost < best * 0.97) { best = cost; bn2 = cand[c]; }   // ties go to the narrower tile
            }
        } else if (d->N > 64) {
            bn2 = 64;
        }
        if (impl == 0) {
            const bool stream64 = d->N <= 64 && d->KH * d->KW == 1 && Ktot * es >= 512;
            // tiny-M, long-K layers (P6, P7): only the 128-row kernel has the split-K path
            const long long grid1 = ((Mtot + BM - 1) / BM) * ((d->N + 127) / 128);
            const bool splitk = d->ngroups == 1 && grid1 < 128 && Ktot * es / 128 >= 16;
            impl = (stream64 || splitk) ? 1 : 2;
        }
    }
    // halo-sharing kernel: stride-1 'same' KHxKW conv over dense NHWC inputs whose taps span whole 128-byte chunks
    bool halo_ok = d->KW >= 2 && d->KW <= 4 && d->sy == 1 && d->sx == 1 && (d->Crun * es) % 128 == 0 && d->pix_stride == d->Crun &&
                   d->pad_l >= 0 && d->pad_l < d->KW && d->pad_t >= 0 && d->pad_t < d->KH;
    for (int i = 0; i < d->ngroups && halo_ok; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (s.Hin != s.Hout || s.Win != s.Wout || s.in_row_stride != (long long)s.Win * d->pix_stride ||
            s.in_img_stride != (long long)s.Hin * s.in_row_stride) halo_ok = false;
    }
    if (s2) impl = 2;                                  // only the 256-row per-tap kernel walks a second source
    if (q8) {                                          // fp8 layers: the shapes the halo kernel accepts (either kernel runs them)
        if (!halo_ok) return rtn_fail(h, RTN_EINVAL, "conv fp8: only stride-1 'same' KHxKW (KW 2..4) layers over dense NHWC inputs with whole 128-byte channel chunks");
        impl = bn2 == 256 ? 3 : 2;                     // as for bf16: halo kernel with 256-wide tiles, per-tap kernel below
    }
    if (impl == 3 && !halo_ok) impl = 2;
    // measured (tools/ab_conv.py): with 256-wide tiles (the grouped head layers) the halo kernel and the per-tap kernel are level
    // (0.2113 vs 0.2116 ms) and the halo kernel stays; with the 64-wide tiles the cost model gives the single-group layers
    // (res4/res5 3x3, P3, P4) the per-tap kernel is 1-10 % faster since both request their first fragments ahead of the staging
    // issue (res5 3x3 0.0749 vs 0.0821 ms, res4 0.0609 vs 0.0634), so those went back to it.  RTN_CONV_HALO=2 forces the halo
    // kernel wherever it applies, 0 disables it.
    const int halo_env = rtn_env_int("RTN_CONV_HALO", 1);
    if (impl == 2 && halo_ok && rtn_conv_impl_override() == 0 && (halo_env == 2 || (halo_env == 1 && d->N >= 256 && bn2 == 256))) impl = 3;
    // Fourth generation (rtn_conv_halo8.hip): every stride-1 3x3 bf16 layer with 129..256 output channels and a bias/ReLU epilogue
    // (the grouped head towers, P3-P5, res4 branch2b), persistent and on the staggered 8-phase schedule.  Measured against
    // generation 3 / 2 in one process (tools/ab_conv.py): head layer 0.225 -> 0.150 ms, P3 0.201 -> 0.150, res4 3x3 0.063 -> 0.049,
    // P4 0.062 -> 0.047.  RTN_CONV_H8=0 turns it off (A/B), RTN_CONV_IMPL=4 forces it like any other generation;
    // RTN_CONV_H8_GRID limits the workgroup count (tests: several tiles per workgroup on small layers), RTN_CONV_H8_STAGGER=0 runs
    // the two wave groups in lockstep (A/B: 0.181 ms on the head layer).
    if (!s2 && out8_scale == 0.f && (q8 ? d->dtype == RTN_FP8 && !query : d->dtype == RTN_BF16)) {
        const int h8 = rtn_env_int("RTN_CONV_H8", 1);
        const int forced = rtn_conv_impl_override();
        if (forced == 4 || (forced == 0 && h8 != 0)) {
            const int rc = rtn_conv_halo8_try(h, d, rtn_env_int("RTN_CONV_H8_GRID", 0), rtn_env_int("RTN_CONV_H8_STAGGER", 1) != 0, forced == 4,
                                              rtn_env_int("RTN_CONV_H8_MI", 0), ws_ptr, ws_cap, query, rtn_env_int("RTN_CONV_H8_KSPLIT", 0), q8);
            if (rc == RTN_OK && !query) h->last_conv_impl = 4;
            if (rc <= 0) return rc;                    // launched (or failed): done; 1 = not eligible, fall through
        }
    }
    // Sixth generation (rtn_conv_halon.hip): the head OUTPUT convolutions (3x3, <= 48 channels, f32 result in the concatenated
    // tensor).  RTN_CONV_HN=0 turns it off, RTN_CONV_IMPL=6 forces it.
    if (!query && !s2 && !q8 && out8_scale == 0.f && d->dtype == RTN_BF16 && d->N <= 48 && (d->flags & RTN_CONV_OUT_F32)) {
        const int hn = rtn_env_int("RTN_CONV_HN", 1);
        const int forced = rtn_conv_impl_override();
        if (forced == 6 || (forced == 0 && hn != 0)) {
            const int rc = rtn_conv_halon_try(h, d, rtn_env_int("RTN_CONV_H8_GRID", 0), forced == 6);
            if (rc == RTN_OK) h->last_conv_impl = 6;
            if (rc <= 0) return rc;
        }
    }
    // Fifth generation (rtn_conv_gemm8.hip): the 1x1 layers with N % 256 == 0 and a bias / ReLU epilogue, one or two sources.
    // RTN_CONV_G8=0 turns it off, RTN_CONV_IMPL=5 forces it; RTN_CONV_G8_MI pins the tile height (2 | 3 row fragments per wave).
    // RTN_CONV_G8_SK: its stream-K form (-1 = cost model, 0 = never, 1 = wherever the shape allows).
    if (!q8 && out8_scale == 0.f && d->dtype == RTN_BF16 && d->KH == 1 && d->KW == 1) {
        const int g8 = rtn_env_int("RTN_CONV_G8", 1);
        const int forced = rtn_conv_impl_override();
        if (forced == 5 || (forced == 0 && g8 != 0)) {
            const int rc = rtn_conv_gemm8_try(h, d, s2, rtn_env_int("RTN_CONV_H8_GRID", 0), rtn_env_int("RTN_CONV_H8_STAGGER", 1) != 0,
                                              forced == 5, rtn_env_int("RTN_CONV_G8_MI", 0), ws_sync, ws_ptr, ws_cap, query,
                                              rtn_env_int("RTN_CONV_G8_SK", -1));
            if (rc == RTN_OK && !query) h->last_conv_impl = 5;
            if (rc <= 0) return rc;
        }
    }
    if (impl == 4 || impl == 7) impl = halo_ok ? 3 : 2;
    if (impl == 5 || impl == 6) impl = 2;
    const int TM = impl == 3 ? BM2 - (d->KW - 1) : (impl == 2 ? BM2 : BM);
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
        const int ostep = s.out_step > 1 ? s.out_step : 1;
        if (ostep > 1 && s.out_pix_w < (s.Wout - 1) * ostep + 1) return rtn_fail(h, RTN_EINVAL, "conv: group %d out_pix_w %d too small for out_step %d", i, s.out_pix_w, ostep);
        const long long last_pix = ostep > 1 ? (long long)(s.Hout - 1) * ostep * s.out_pix_w + (long long)(s.Wout - 1) * ostep : cells - 1;
        const long long out_max = (long long)(d->batch - 1) * s.out_img_stride + s.out_off + last_pix * d->out_ld + d->N;
        if (s.out_off < 0 || out_max > s.out_elems) return rtn_fail(h, RTN_EBOUNDS, "conv: group %d writes reach element %lld of a %lld-element output", i, out_max, (long long)s.out_elems);
        if (s.out_img_stride % 8 || s.out_off % 8) vec_ok = false;
        if (has_res) {
            if (!s.res) return rtn_fail(h, RTN_EINVAL, "conv: group %d residual flag without res", i);
            const long long rcells = (d->flags & RTN_CONV_RES_UPSAMPLE) ? (long long)s.Hres * s.Wres : last_pix + 1;
            if ((d->flags & RTN_CONV_RES_UPSAMPLE) && (s.Hres < 1 || s.Wres < 1)) return rtn_fail(h, RTN_EINVAL, "conv: group %d bad residual extent", i);
            const long long res_max = (long long)(d->batch - 1) * s.res_img_stride + (rcells - 1) * s.res_ld + d->N;
            if (res_max > s.res_elems) return rtn_fail(h, RTN_EBOUNDS, "conv: group %d residual reads reach %lld of %lld", i, res_max, (long long)s.res_elems);
            if (s.res_ld % 8 || s.res_img_stride % 8) vec_ok = false;
        }
        if (d->flags & RTN_CONV_RELU_MASK) {
            if (!s.mask || ((uintptr_t)s.mask & 15)) return rtn_fail(h, RTN_EINVAL, "conv: group %d RELU_MASK without an aligned mask", i);
            const long long mask_max = (long long)(d->batch - 1) * s.mask_img_stride + last_pix * s.mask_ld + d->N;
            if (mask_max > s.mask_elems) return rtn_fail(h, RTN_EBOUNDS, "conv: group %d mask reads reach %lld of %lld", i, mask_max, (long long)s.mask_elems);
            if (s.mask_ld % 8 || s.mask_img_stride % 8) vec_ok = false;
        }
        g.mask = (const char*)s.mask;
        g.mask_img_stride = s.mask_img_stride;
        g.mask_ld = s.mask_ld;
        g.out_step = ostep;
        g.out_pix_w = s.out_pix_w;
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
        mtiles += (M + TM - 1) / TM;
    }
    (void)out_f32;
    if (out8_scale != 0.f) {                           // bf16 layer with e4m3 output
        if (d->dtype != RTN_BF16 || (d->flags & (RTN_CONV_OUT_F32 | RTN_CONV_SIGMOID)) || !(out8_scale > 0.f))
            return rtn_fail(h, RTN_EINVAL, "conv fp8-out: a bf16 layer without OUT_F32 / SIGMOID and a positive scale are required");
        if (!vec_ok) return rtn_fail(h, RTN_EINVAL, "conv fp8-out: N, out_ld, output strides and offsets must be multiples of 8");
        p.out_fp8 = 1;
        p.out_scale = out8_scale;
    }
    if (q8 && !vec_ok) return rtn_fail(h, RTN_EINVAL, "conv fp8: output strides and offsets must be multiples of 8 elements");
    if (q8) { p.acc_scale = q8->acc_scale; p.out_scale = q8->out_scale; p.out_fp8 = q8->out_dtype == RTN_FP8 ? 1 : 0; }
    if ((d->flags & RTN_CONV_RELU_MASK) && !vec_ok) return rtn_fail(h, RTN_EINVAL, "conv: RELU_MASK needs N, out_ld, strides multiples of 8");
    int BN = impl >= 2 ? bn2 : (d->N <= 64 ? 64 : 128);
    if (impl >= 2) {
        const int bn_env = rtn_env_int("RTN_CONV_BN2", 0);   // A/B override of the 256-row kernel's tile width
        if ((bn_env == 64 || bn_env == 128 || bn_env == 256) && bn_env <= ((d->N + 63) / 64) * 64) BN = bn_env;
    }
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
    if (s2) {
        p.in2 = (const char*)s2->in;
        p.in2_bytes = (unsigned)(s2->in_elems * es);
        p.nkt1 = (int)(K1 * es / 128);
        p.in2_img_stride_b = s2->in_img_stride * es;
        p.in2_row_stride_b = (int)((long long)s2->in_row_stride * es);
        p.in2_pix_stride_b = (int)((long long)s2->pix_stride * es);
        p.in2_step = s2->step;
    }
    p.crun_mask = d->Crun - 1;
    p.taps_uniform = ((long long)d->Crun * es) % 128 == 0 ? 1 : 0;
    p.KW = d->KW;
    p.KH = d->KH;
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
    {
        bool dense_out = !(d->flags & RTN_CONV_RES_UPSAMPLE), dense_in = (d->KH == 1 && d->KW == 1 && d->sy == 1 && d->sx == 1 && d->pad_t == 0 && d->pad_l == 0);
        for (int i = 0; i < d->ngroups; ++i) {
            const rtn_conv_group_t& s = d->g[i];
            const long long cells = (long long)s.Hout * s.Wout;
            if (s.out_step > 1 || s.out_off != 0 || s.out_img_stride != cells * d->out_ld) dense_out = false;
            if (has_res && (s.res_img_stride != cells * s.res_ld)) dense_out = false;
            if ((d->flags & RTN_CONV_RELU_MASK) && (s.mask_img_stride != cells * s.mask_ld)) dense_out = false;
            if (s.Hin != s.Hout || s.Win != s.Wout || s.in_row_stride != (long long)s.Win * d->pix_stride ||
                s.in_img_stride != (long long)s.Hin * s.in_row_stride) dense_in = false;
        }
        p.dense_out = dense_out ? 1 : 0;
        p.dense_in = dense_in ? 1 : 0;
    }
    const long long grid = mtiles * p.ntiles_n;
    if (grid < 1 || grid > 0x7fffffffll) return rtn_fail(h, RTN_EINVAL, "conv: grid %lld", grid);

    dim3 gdim((unsigned)grid);
    // ---- tail split (see KParams): the grid is a few tiles over a whole number of rounds of the resident slots
    int tail_tiles = 0;
    if (impl >= 2 && d->ngroups == 1 && !s2 && !q8 && rtn_env_int("RTN_CONV_TAIL", 1) != 0) {
        long long slots = (long long)(h->num_cus > 0 ? h->num_cus : 256) * (BN == 64 ? 2 : 1);
        { const int sl_env = rtn_env_int("RTN_CONV_TAIL_SLOTS", 0); if (sl_env > 0) slots = sl_env; }   // tests: pretend a small chip
        const long long rest = grid % slots;
        const int units = impl == 3 ? d->KH * (d->Crun * es / 128) : p.nkt;      // (kh, chunk) groups | K steps
        const int min_units = impl == 3 ? 1 : 2;
        // measured (tools/ab_conv.py RTN_CONV_TAIL=0|1): -13 % on res4 3x3 / P4, -5 % on the res4 1x1 (one full round + a few tiles);
        // +3..5 % where two or more full rounds precede the tail (res2, res3, C3: the extra launch costs more than the short
        // last round), so only grids between one and two rounds are split
        if (grid > slots && grid < 2 * slots && rest > 0 && rest * 4 <= slots && units >= 4 * min_units) {
            long long S = slots / rest;
            if (S > units / min_units) S = units / min_units;
            if (S > 16) S = 16;
            const int per = (int)((units + S - 1) / S);
            S = (units + per - 1) / per;
            const long long slab_bytes = rest * S * (long long)BM2 * BN * 4;
            if (query && S >= 2 && slab_bytes <= ws_cap) { *query = (size_t)slab_bytes; return RTN_OK; }
            float* scratch = S >= 2 ? ws_ptr : nullptr;
            if (scratch && slab_bytes <= ws_cap) {
                tail_tiles = (int)rest;
                p.tail_mode = 1;
                p.tail_main = (int)(grid - rest);
                p.tail_tiles = tail_tiles;
                p.tail_slices = (int)S;
                p.tail_per = per;
                p.scratch = scratch;
                gdim = dim3((unsigned)(grid - rest + rest * S));
            }
        }
    }
    if (query && impl >= 2) return RTN_OK;
    for (int pass = 0; pass < (tail_tiles ? 2 : 1); ++pass) {
    if (pass == 1) {                                   // the finish launch: one workgroup per tail tile
        RTN_CHECK_LAUNCH(h, "conv (tail-split main launch)");
        p.tail_mode = 2;
        gdim = dim3((unsigned)tail_tiles);
    }
    if (impl == 3) {
        dim3 bdim(NT2);
        const unsigned epib = 8u * 32u * (unsigned)(BN / 2 + 4) * 4u;
        const int nbst = 2;      // (a third B stage fits beside the halos for every tile width and changed nothing measurable: round 1)
        p.nstages = nbst;
        unsigned ldsb = 2u * BM2 * 128u + (unsigned)nbst * (unsigned)BN * 128u;
        if (ldsb < epib) ldsb = epib;
        p.korder_chunks = d->Crun * es / 128;
        p.korder_kw = d->KW;
#define RTN_L3(E, B)                                                                                     \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_igemm3_kernel<E, B>,                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_igemm3_kernel<E, B>), gdim, bdim, ldsb, h->stream, p);                  \
    } while (0)
        if (es == 1) { if (BN == 64) RTN_L3(1, 64); else if (BN == 128) RTN_L3(1, 128); else RTN_L3(1, 256); }
        else if (es == 2) { if (BN == 64) RTN_L3(2, 64); else if (BN == 128) RTN_L3(2, 128); else RTN_L3(2, 256); }
        else         { if (BN == 64) RTN_L3(4, 64); else if (BN == 128) RTN_L3(4, 128); else RTN_L3(4, 256); }
#undef RTN_L3
    } else if (impl == 2) {
        dim3 bdim(NT2);
        // LDS: one stage when the K loop is a single step (the streaming 1x1 layers: more workgroups per CU), else two
        // LDS ring depth: as many stages as fit 160 KB (64-wide: 3 by default, 128-wide: 3, 256-wide: 2), never more
        // than the K loop has steps.
        const unsigned epib = 8u * 32u * (unsigned)(BN / 2 + 4) * 4u;     // the wave-private epilogue slices (EPI_BYTES)
        const unsigned sb = (unsigned)(BM2 + BN) * 128u;
        int nst = 2;   // measured: deeper rings lose (they cost the second resident workgroup on the 64-wide tile)
        while (nst > 1 && (unsigned)nst * sb > 160u * 1024u) --nst;
        if (nst > p.nkt) nst = p.nkt;
        if (nst < 2 && p.nkt > 1) nst = 2;
        // K-step order: (chunk, kw) inside a kernel row when a tap spans whole 128-byte chunks
        p.korder_chunks = p.nkt; p.korder_kw = 1;
        if (d->KW > 1 && (d->Crun * es) % 128 == 0) {
            p.korder_chunks = d->Crun * es / 128;
            p.korder_kw = d->KW;
        }
        p.nstages = nst < 2 ? 2 : nst;      // a single-step K loop never touches its second buffer
        unsigned ldsb = (unsigned)nst * sb;
        if (ldsb < epib) ldsb = epib;
#define RTN_L2K(E, B, I)                                                                                 \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_igemm2_kernel<E, B, I>,                     \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));           \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_igemm2_kernel<E, B, I>), gdim, bdim, ldsb, h->stream, p);               \
    } while (0)
#define RTN_L2(E, B)                                                                                     \
    do {                                                                                                 \
        RTN_L2K(E, B, false);   /* (DMA issues interleaved with the MFMA groups: measured 12 % slower in round 1, instance removed) */ \
    } while (0)
#define RTN_L2D(E, B)                                                                                    \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_igemm2_kernel<E, B, false, true>,           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_igemm2_kernel<E, B, false, true>), gdim, bdim, ldsb, h->stream, p);     \
    } while (0)
        if (s2) {
            if (es == 2) { if (BN == 64) RTN_L2D(2, 64); else if (BN == 128) RTN_L2D(2, 128); else RTN_L2D(2, 256); }
            else         { if (BN == 64) RTN_L2D(4, 64); else if (BN == 128) RTN_L2D(4, 128); else RTN_L2D(4, 256); }
        } else
        if (es == 1) { if (BN == 64) RTN_L2K(1, 64, false); else if (BN == 128) RTN_L2K(1, 128, false); else RTN_L2K(1, 256, false); }
        else if (es == 2) { if (BN == 64) RTN_L2(2, 64); else if (BN == 128) RTN_L2(2, 128); else RTN_L2(2, 256); }
        else         { if (BN == 64) RTN_L2(4, 64); else if (BN == 128) RTN_L2(4, 128); else RTN_L2(4, 256); }
#undef RTN_L2D
#undef RTN_L2
#undef RTN_L2K
    }
    }   // pass
    if (impl < 2) {
        if (query && !(d->ngroups == 1 && grid < 128 && p.nkt >= 16)) return RTN_OK;
        dim3 bdim(NT);
        // split-K: a long K loop on a grid that cannot fill the chip (P6: 36 workgroups x 288 K steps)
        int ksplit = 1;
        const int ksplit_env = rtn_env_int("RTN_CONV_SPLITK", -1);
        if (d->ngroups == 1 && grid < 128 && p.nkt >= 16 && ksplit_env != 0) {
            ksplit = (int)((512 + grid - 1) / grid);
            if (ksplit > p.nkt / 4) ksplit = p.nkt / 4;
            if (ksplit_env > 1) ksplit = ksplit_env < p.nkt ? ksplit_env : p.nkt;
            const long long slab = (long long)p.g[0].M * (((d->N + 7) / 8) * 8) * 4;
            while (ksplit > 1 && slab * ksplit > ws_cap) --ksplit;
            if (ksplit >= 2) {                                   // every slice must own at least one K step (it writes its whole slab)
                const int per = (p.nkt + ksplit - 1) / ksplit;
                ksplit = (p.nkt + per - 1) / per;
            }
            if (query) { if (ksplit >= 2) *query = (size_t)(slab * ksplit); return RTN_OK; }
            float* scratch = ksplit >= 2 ? ws_ptr : nullptr;
            if (ksplit < 2 || !scratch) ksplit = 1;
            p.scratch = scratch;
        }
        if (query) return RTN_OK;                      // query mode never launches (RTN_CONV_SPLITK=0 skips the block above)
        if (ksplit > 1) {
            p.ksplit = ksplit;
            p.kt_per_split = (p.nkt + ksplit - 1) / ksplit;
            p.scratch_ld = ((d->N + 7) / 8) * 8;
            gdim = dim3((unsigned)grid, (unsigned)ksplit);
        }
        if (es == 2) {
            if (BN == 64) hipLaunchKernelGGL((conv_igemm_kernel<2, 64>), gdim, bdim, 0, h->stream, p);
            else          hipLaunchKernelGGL((conv_igemm_kernel<2, 128>), gdim, bdim, 0, h->stream, p);
        } else {
            if (BN == 64) hipLaunchKernelGGL((conv_igemm_kernel<4, 64>), gdim, bdim, 0, h->stream, p);
            else          hipLaunchKernelGGL((conv_igemm_kernel<4, 128>), gdim, bdim, 0, h->stream, p);
        }
        if (ksplit > 1) {
            RTN_CHECK_LAUNCH(h, "conv_igemm_kernel (split-K)");
            const long long work = (long long)p.g[0].M * (p.scratch_ld / 8);
            long long fg = (work + 255) / 256;
            if (fg > 2048) fg = 2048;
            if (es == 2) hipLaunchKernelGGL((conv_splitk_finish_kernel<2>), dim3((unsigned)fg), dim3(256), 0, h->stream, p);
            else         hipLaunchKernelGGL((conv_splitk_finish_kernel<4>), dim3((unsigned)fg), dim3(256), 0, h->stream, p);
        }
    }
    RTN_CHECK_LAUNCH(h, "conv_igemm_kernel");
    h->last_conv_impl = impl;
    h->last_conv_tile = ((impl >= 2 ? BM2 : BM) << 16) | BN;
    return RTN_OK;
}

extern "C" int rtn_debug_last_conv_impl(rtn_handle_t h) { return h ? h->last_conv_impl : RTN_EINVAL; }

extern "C" int rtn_conv2d_fwd(rtn_handle_t h, const rtn_conv_desc_t* d) { return conv_launch(h, d); }

extern "C" size_t rtn_conv2d_workspace_bytes(rtn_handle_t h, const rtn_conv_desc_t* d) {
    size_t n = 0;
    if (!h || !d) return 0;
    // the K-split paths are taken by plain forward / dgrad launches and by the dual-source form (below), not by the fp8 variants
    if (conv_launch(h, d, nullptr, nullptr, 0.f, &n) != RTN_OK) return 0;
    return n ? n + RTN_CONV_SYNC_BYTES : 0;
}
extern "C" size_t rtn_conv1x1_dual_workspace_bytes(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_src2_t* s2) {
    size_t n = 0;
    if (!h || !d || !s2) return 0;
    if (conv_launch(h, d, s2, nullptr, 0.f, &n) != RTN_OK) return 0;
    return n ? n + RTN_CONV_SYNC_BYTES : 0;
}
extern "C" int rtn_conv_workspace_init(rtn_handle_t h, void* workspace, size_t workspace_bytes) {
    if (!h) return RTN_EINVAL;
    if (!workspace || ((uintptr_t)workspace & 15)) return rtn_fail(h, RTN_EINVAL, "conv workspace init: null / misaligned");
    if (workspace_bytes < RTN_CONV_SYNC_BYTES) return rtn_fail(h, RTN_ENOMEM, "conv workspace init: %zu < %d", workspace_bytes, RTN_CONV_SYNC_BYTES);
    RTN_HIP(h, hipMemsetAsync(workspace, 0, RTN_CONV_SYNC_BYTES, h->stream));
    RTN_HIP(h, hipStreamSynchronize(h->stream));       // zero on return: the buffer may then serve launches on any stream
    return RTN_OK;
}
extern "C" int rtn_debug_last_conv_streamk(rtn_handle_t h) { return h ? h->last_conv_streamk : RTN_EINVAL; }
extern "C" int rtn_debug_last_conv_tile(rtn_handle_t h) { return h ? h->last_conv_tile : RTN_EINVAL; }
extern "C" int rtn_debug_conv_sync_timeouts(rtn_handle_t h, const void* workspace, unsigned* count) {
    if (!h) return RTN_EINVAL;
    if (!workspace || !count) return rtn_fail(h, RTN_EINVAL, "conv sync timeouts: null argument");
    RTN_HIP(h, hipStreamSynchronize(h->stream));
    RTN_HIP(h, hipMemcpy(count, (const char*)workspace + RTN_CONV_SYNC_BYTES - 4, 4, hipMemcpyDeviceToHost));
    return RTN_OK;
}

extern "C" int rtn_conv1x1_dual_fwd(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_src2_t* s2) {
    if (!h) return RTN_EINVAL;
    if (!s2) return rtn_fail(h, RTN_EINVAL, "conv dual: null second source");
    return conv_launch(h, d, s2);
}

extern "C" int rtn_conv2d_fwd_fp8out(rtn_handle_t h, const rtn_conv_desc_t* d, float out_scale) {
    if (!h) return RTN_EINVAL;
    if (!(out_scale > 0.f)) return rtn_fail(h, RTN_EINVAL, "conv fp8-out: scale must be positive");
    return conv_launch(h, d, nullptr, nullptr, out_scale);
}

extern "C" int rtn_conv2d_fp8_fwd(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_fp8_t* q) {
    if (!h) return RTN_EINVAL;
    if (!q) return rtn_fail(h, RTN_EINVAL, "conv fp8: null scale block");
    return conv_launch(h, d, nullptr, q);
}

extern "C" int rtn_conv2d_dgrad(rtn_handle_t h, const rtn_conv_desc_t* d) {
    if (!h) return RTN_EINVAL;
    if (!d) return rtn_fail(h, RTN_EINVAL, "dgrad: null descriptor");
    if (d->sy != 1 || d->sx != 1) return rtn_fail(h, RTN_EINVAL, "dgrad: runs as a stride-1 convolution over dY (use out_step / rtn_zero_insert2)");
    if (d->flags & (RTN_CONV_RELU | RTN_CONV_SIGMOID | RTN_CONV_RES_UPSAMPLE)) return rtn_fail(h, RTN_EINVAL, "dgrad: forward-only epilogue flag set");
    return conv_launch(h, d);
}

namespace {
// w_d[c][(KH-1-kh, KW-1-kw, n)] = w[n][(kh, kw, c)];  rows c >= Cin and columns n >= N are zero.
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const T* __restrict__ wf, T* __restrict__ wd, int N, int KH, int KW,
                                                         int Cin, int Crun, int rows_d) {
    const long long Kd = (long long)KH * KW * Crun;
    const long long total = (long long)rows_d * Kd;
    const long long Kf = (long long)KH * KW * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / Kd);
        const int r = (int)(i - (long long)c * Kd);
        const int tap = r / Crun, n = r - tap * Crun;
        const int khd = tap / KW, kwd = tap - khd * KW;
        T v = T(0);
        if (c < Cin && n < N) v = wf[(long long)n * Kf + ((long long)(KH - 1 - khd) * KW + (KW - 1 - kwd)) * Cin + c];
        wd[i] = v;
    }
}
// All layers in one launch.  table[l] = {src, dst, N, KH, KW, Cin, Crun, rows_d, first flat element of layer l, -}.
// Per tap the repack is a TRANSPOSE of the (filter n, channel c) plane: w_d[c][tap'][n] = w[n][tap][c].  Round 2 did it one element
// per thread with a 2-byte read every K elements apart (0.28 ms per optimizer step for 72 MB); now a workgroup moves 64 x 64 tiles
// through LDS - 128-byte runs of c on the way in, 128-byte runs of n on the way out - and finds its tiles in a per-layer prefix that
// every workgroup builds once from the table (the API stays as it was: the host only knows the table's device address).
constexpr int PK_T = 64;
constexpr int PK_MAXL = 1024;
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_multi_kernel(const long long* __restrict__ table, int nlayers) {
    __shared__ long long pre[PK_MAXL + 1];
    __shared__ T tile[PK_T][PK_T + 2];
    const int t = threadIdx.x;
    if (t == 0) {
        long long acc = 0;
        for (int l = 0; l < nlayers; ++l) {
            const long long* e = table + l * 10;
            const long long taps = e[3] * e[4], tn = (e[6] + PK_T - 1) / PK_T, tc = (e[7] + PK_T - 1) / PK_T;
            pre[l] = acc;
            acc += taps * tn * tc;
        }
        pre[nlayers] = acc;
    }
    __syncthreads();
    const long long total = pre[nlayers];
    for (long long tt = blockIdx.x; tt < total; tt += gridDim.x) {
        int lo = 0, hi = nlayers - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pre[mid] <= tt) lo = mid; else hi = mid - 1;
        }
        const long long* e = table + lo * 10;
        const T* wf = reinterpret_cast<const T*>(e[0]);
        T* wd = reinterpret_cast<T*>(e[1]);
        const int N = (int)e[2], KH = (int)e[3], KW = (int)e[4], Cin = (int)e[5], Crun = (int)e[6], rows_d = (int)e[7];
        const int tn = (Crun + PK_T - 1) / PK_T, tc = (rows_d + PK_T - 1) / PK_T;
        long long j = tt - pre[lo];
        const int ci = (int)(j % tc); j /= tc;
        const int ni = (int)(j % tn);
        const int tap = (int)(j / tn);                        // destination tap (khd, kwd)
        const int khd = tap / KW, kwd = tap - khd * KW;
        const int tapf = (KH - 1 - khd) * KW + (KW - 1 - kwd);
        const long long Kd = (long long)KH * KW * Crun, Kf = (long long)KH * KW * Cin;
        const int n0 = ni * PK_T, c0 = ci * PK_T;
        const int col = t & 63, r4 = t >> 6;
#pragma unroll 4
        for (int i = 0; i < PK_T / 4; ++i) {                  // rows = filters n, columns = channels c (contiguous in the source)
            const int n = n0 + i * 4 + r4, c = c0 + col;
            T v = T(0);
            if (n < N && c < Cin) v = wf[(long long)n * Kf + (long long)tapf * Cin + c];
            tile[i * 4 + r4][col] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < PK_T / 4; ++i) {                  // rows = channels c, columns = filters n (contiguous in the destination)
            const int c = c0 + i * 4 + r4, n = n0 + col;
            if (c < rows_d && n < Crun) wd[(long long)c * Kd + (long long)tap * Crun + n] = tile[col][i * 4 + r4];
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int rtn_pack_dgrad_weights_multi(rtn_handle_t h, const int64_t* table_dev, int nlayers, int64_t total_elems, int dtype) {
    if (!h) return RTN_EINVAL;
    if (!table_dev || nlayers < 1 || total_elems < 1) return rtn_fail(h, RTN_EINVAL, "pack_dgrad_multi: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "pack_dgrad_multi: bad dtype");
    if (nlayers > PK_MAXL) return rtn_fail(h, RTN_EINVAL, "pack_dgrad_multi: at most %d layers per call", PK_MAXL);
    long long g = (total_elems + PK_T * PK_T - 1) / (PK_T * PK_T);      // about one 64 x 64 tile per workgroup, at most 8 per CU
    const long long cap = 8ll * (h->num_cus > 0 ? h->num_cus : 256);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    if (dtype == RTN_BF16)
        hipLaunchKernelGGL((pack_dgrad_multi_kernel<unsigned short>), dim3((unsigned)g), dim3(256), 0, h->stream, (const long long*)table_dev, nlayers);
    else
        hipLaunchKernelGGL((pack_dgrad_multi_kernel<float>), dim3((unsigned)g), dim3(256), 0, h->stream, (const long long*)table_dev, nlayers);
    RTN_CHECK_LAUNCH(h, "pack_dgrad_multi_kernel");
    return RTN_OK;
}

extern "C" int rtn_pack_dgrad_weights(rtn_handle_t h, const void* w_fwd, void* w_dgrad, int dtype, int N, int w_rows_fwd, int KH,
                                      int KW, int Cin, int Cout_run, int w_rows_dgrad) {
    if (!h) return RTN_EINVAL;
    if (!w_fwd || !w_dgrad || N < 1 || KH < 1 || KW < 1 || Cin < 1) return rtn_fail(h, RTN_EINVAL, "pack_dgrad: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "pack_dgrad: bad dtype");
    if (w_rows_fwd < N || Cout_run < N || w_rows_dgrad < Cin) return rtn_fail(h, RTN_EINVAL, "pack_dgrad: rows %d/%d/%d too small for N %d Cin %d", w_rows_fwd, Cout_run, w_rows_dgrad, N, Cin);
    const long long total = (long long)w_rows_dgrad * KH * KW * Cout_run;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (dtype == RTN_BF16)
        hipLaunchKernelGGL((pack_dgrad_kernel<unsigned short>), dim3((unsigned)g), dim3(256), 0, h->stream, (const unsigned short*)w_fwd,
                           (unsigned short*)w_dgrad, N, KH, KW, Cin, Cout_run, w_rows_dgrad);
    else
        hipLaunchKernelGGL((pack_dgrad_kernel<float>), dim3((unsigned)g), dim3(256), 0, h->stream, (const float*)w_fwd, (float*)w_dgrad,
                           N, KH, KW, Cin, Cout_run, w_rows_dgrad);
    RTN_CHECK_LAUNCH(h, "pack_dgrad_kernel");
    return RTN_OK;
}
