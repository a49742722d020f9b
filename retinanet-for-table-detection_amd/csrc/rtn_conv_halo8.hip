// rtn_conv_halo8.hip — the head-tower convolution kernel: stride-1 'same' KHxKW convolution with <= 256 output channels as a
// PERSISTENT 256 x 256 implicit GEMM on the staggered 8-phase schedule (cdna_hip_programming.md §5, "The 256^2 8-phase template").
//
// Replaces, for the layers it accepts, the third kernel generation of rtn_conv.hip (conv_igemm3_kernel<2, 256>) on
//   model/defineModel.py:101-117,155-163  the 4 x [3x3 conv 256 + ReLU] towers of both heads over P3..P7 (one grouped launch),
//   model/defineModel.py:183-203           P3 / P4 / P5 (3x3, 256 -> 256), keras_resnet's res4 branch2b (3x3, 256 -> 256).
//
// What is different from generation 3 (same halo idea: the KW taps of a kernel row read ONE staged image of 256 consecutive
// input pixels, shifted by one row of the image per tap):
//   * K step (64 deep) = 4 phases of {fragment reads + LDS-DMA issue | s_barrier | 16 MFMAs | s_barrier}.  Waves 4-7 run one
//     barrier behind waves 0-3 (SIMD partners are (w, w+4)), so one wave of every SIMD multiplies while the other reads.
//   * LDS-DMA stays in flight across barriers: B tiles in a ring of 3 (the B tile of step s+2 is issued during step s),
//     halos double-buffered one (kh, chunk) group ahead, ONE counted s_waitcnt vmcnt(N) per K step (phase 4), never 0.
//     A staged buffer is read one phase after the wait that retires it and restaged >= 2 phases after its last read.
//   * persistent: a workgroup walks its tiles as ONE stream of K steps; the first stages of the next tile are issued during the
//     last steps of the current one, so no prologue latency and no launch gap sits between tiles, and the epilogue's stores
//     drain while the next tile multiplies.
//   * no LDS in the epilogue: the B tile is staged with its weight rows PERMUTED (LDS position 16 j + c holds output channel
//     8 c + j), so that lane (q, c) of a wave ends up with 8 CONSECUTIVE channels of a pixel in its 8 column fragments and stores
//     16 bytes per row straight from registers (one wave-instruction = 4 pixel rows x 256 contiguous bytes).  Bias is the
//     accumulators' initial value.
//   * horizontal image edge: instead of zeroing A fragments in registers (16 v_cndmask per step), row 255 of every halo is a
//     ZERO row (TM = 256 - KW output rows per tile) and a lane whose tap leaves the image reads that row.
//
// LDS (160 KiB, all of it): B ring 3 x 32 KiB at 0, halos 2 x 32 KiB at 96 KiB.  One workgroup (512 threads) per CU.
#include "rtn_internal.h"

#ifndef RTN_H8_ABLATE
#define RTN_H8_ABLATE 0
#endif

namespace {

// -DRTN_H8_STAMP: in-kernel time stamps (s_memtime) of one K step of workgroup 0, waves 0 and 4, read back by rtn_debug_h8_stamps
// (tools/h8_stamps.py).  Not in production builds.
#ifdef RTN_H8_STAMP
__device__ unsigned long long g_h8_stamps[2][64];
#define H8_STAMP(K_)                                                                                 \
    if (stamp_on) {                                                                                  \
        unsigned long long t_;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                 \
        if (lane == 0 && stamp_n < 64) g_h8_stamps[grp][stamp_n] = t_;                               \
        ++stamp_n;                                                                                   \
    }
#else
#define H8_STAMP(K_)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned OOB = 0xFFFFFF00u;                 // beyond every descriptor: loads return zeros, stores are dropped
constexpr int H8_THREADS = 512;
constexpr int H8_LDS = 160 * 1024;
constexpr unsigned B_STAGE = 32768, A_BASE = 3 * 32768, A_TOGGLE = 0x18000u ^ 0x20000u;

struct H8Group {
    const char* in;
    char* out;
    const char* res;              // residual added to the result (RTN_CONV_RES_SAME: dense [M][res_ld]; may be `out` itself) or null
    const char* mask;             // RTN_CONV_RELU_MASK source (dense [M][mask_ld]) or null
    unsigned in_bytes, out_bytes, res_bytes, mask_bytes;
    int Hin, Win, M, tile_begin;
    int in_row_stride_b;
    int in_img_pad_b;             // bytes between the end of one image of `in` and the start of the next (0: dense; the data gradient of a
                                  // head output reads one level out of the level-concatenated dY)
    float inv_cells, inv_w;
};

struct H8Params {
    H8Group g[RTN_MAX_GROUPS];
    const char* w;
    const float* bias;
    unsigned w_bytes;
    int ngroups, ntiles;
    int N, Kbytes, KH, nchunk, pad_t, pad_l, relu, out_ld, pix_b;
    int res_ld, mask_ld, mask_pre;
    // work items: (row tile, column block of 256 channels, K slice).  ncb = S = 1 for the layers this kernel was built for; the
    // small-M long-K layers (res5 branch2b: 34 row tiles x 2 column blocks, 72 K steps) are cut into S slices of gps (kh, chunk)
    // groups whose f32 partial sums go to slab[s][M][slab_ld] and are summed in slice order by ksplit_finish_kernel
    int ncb, S, gps, nitems, xcd;
    int kh_fast;                  // group order: 1 = the kernel rows of a channel chunk back to back (the rows a tile re-reads for kh = 0, 1, 2 are
                                  // then 1 group = 33 KB per CU apart instead of nchunk groups: they stay in the XCD's L2), 0 = chunks of a kernel row
    float* slab;
    unsigned slab_slice_bytes;
    int slab_ld;
    // fp8 (OCP e4m3) operands (ES = 1): y = acc * acc_scale + bias; stored as bf16, or as e4m3 of clamp(y * out_scale, +-448)
    float acc_scale, out_scale;
    int out_fp8;
};

__device__ __forceinline__ i32x4 make_srd(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// 64 lanes x 16 B from (descriptor, per-lane byte offset `voff` + uniform `soff`) to LDS bytes [lds_addr, lds_addr + 1024).
// asm so that hipcc neither counts nor drains it; the kernel's own counted waits cover it.
__device__ __forceinline__ void dma16(const i32x4& srd, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd), "s"(soff)
                 : "memory");
}

// f / d for 0 <= f < 2^24 with inv = 1.0f / d: the float product is within one of the quotient
__device__ __forceinline__ void divmod24(int f, int d, float inv, int& q, int& r) {
    q = (int)((float)f * inv);
    r = f - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

// A 16-byte buffer store whose data registers the following VALU instructions rewrite needs two wait states on gfx940+; LLVM pads
// them except when the store's soffset is an SGPR (its hazard table treats that form as immune), which left ZERO wait states in the
// fused bottleneck kernel and corrupted dword 0 of such stores (profiles/r3_store_hazard_isa.txt).  Naming the data registers as
// inputs of an asm statement keeps them intact for four wait states whatever the compiler schedules next or wherever it keeps the
// offset; tools/scan_store_hazard.py checks the built library.
#define RTN_STORE_GUARD(V) asm volatile("s_nop 3" :: "v"(V.x), "v"(V.y), "v"(V.z), "v"(V.w));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// fp8 e4m3 x e4m3, K = 128 per instruction at twice the bf16 rate: a lane supplies 32 bytes of its row per operand - the two 16-byte
// fragments the bf16 loop reads for k halves 0 and 1 (which 32 of the row's 128 K positions a lane holds is free as long as A and
// B agree).  Block scales 2^0.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma_fp8(f32x4& acc, const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    // a plain concatenation (REG_SEQUENCE): built element by element the compiler shuffled dwords through VALU moves in front of
    // half of the MFMAs (v_pk_mov + s_nop 6 each)
    const i32x8 A = __builtin_shufflevector(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, a1), 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 B = __builtin_shufflevector(__builtin_bit_cast(i32x4, b0), __builtin_bit_cast(i32x4, b1), 0, 1, 2, 3, 4, 5, 6, 7);
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return w;
}

// MI = row fragments (16 rows) per wave: tiles of R = 64 MI rows (256 or 192).  The host picks the height that needs the fewest
// rounds of workgroups x rows (P3 at batch 8: 529 tiles of 256 rows = 3 rounds, 707 tiles of 192 rows = 3 shorter rounds).
// EPI: bit 0 = residual add (keras Add / accumulated gradient contributions), bit 1 = ReLU mask of the tensor being differentiated
// (the data-gradient launches of the training step, see rtn_conv2d_dgrad): 16 bytes per lane and row, loaded one row fragment ahead.
// SPLIT: the work items carry a column block and a K slice (see H8Params); accumulators start at zero, the bias is added by the
// epilogue (S = 1) or by the finish kernel (S > 1).
// NW: column fragments (16 channels) per wave.  8 = the 256-column tile above.  4 = a 128-column tile for the 65..128-channel layers
// (res3 branch2b): wave tile 64 MI/4.. rows x 64 columns, a K step is TWO phases of {MI + 4 fragment reads | 4 MI MFMAs} (the same
// reads-per-MFMA ratio as the wide tile), the B stage has two 64-row pieces, a lane ends up with 4 consecutive channels (8-byte stores).
// ES: bytes per element.  2 = bf16.  1 = fp8 e4m3 (rtn_conv2d_fp8_fwd): the same LDS bytes (a 128-byte row is 128 K positions), a K step
// is TWO phases of {fragment reads of both k halves | 4 MI MFMAs 16x16x128}, half the K steps per layer; bias / ReLU epilogue only.
// (Round 2-3 experiments on this loop, all measured slower or level and removed from the library in round 4 - the texts are in profiles/:
// two fat phases per bf16 K step, ONE barrier per K step (r2_v3_halo8_ablation.txt), the next phase's fragment reads inside the MFMA
// block (r3_halo8_prefetch.txt), the filters straight into registers (r3_gen7_filters_in_registers.txt).)
template <int KW, int MI, bool STAGGER, int EPI, bool SPLIT, int NW = 8, int ES = 2>
__global__ __launch_bounds__(H8_THREADS, 2) void conv_halo8_kernel(const H8Params p) {
    static_assert(NW == 8 || (NW == 4 && !SPLIT), "the half-width instance: no column blocks / K slices");
    static_assert(ES == 2 || (ES == 1 && NW == 8 && EPI == 0 && !SPLIT), "the fp8 instance: full width, plain epilogue");
    constexpr int NBP = NW / 2;                        // 64-row pieces of a B stage
    constexpr int R = 64 * MI;                         // rows of a tile's halo image
    constexpr int TM = R - KW;                         // output rows per tile; halo rows 0 .. R - 2, row R - 1 = zeros
    constexpr unsigned ZERO_ROW = (unsigned)(R - 1) * 128u;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2;                         // SIMD partners (w, w + 4) sit in different groups
#ifdef RTN_H8_STAMP
    bool stamp_on = false, first_item = true;
    int stamp_n = 0;
#endif
    const int wm = wave >> 1, wn = wave & 1;           // wave tile: rows [16 MI wm, + 16 MI) x columns [128 wn, +128)
    const int lr = lane >> 3, sc = (lane & 7) ^ lr;    // staging: row inside an 8-row piece, SOURCE chunk (swizzle on the source)
    const int lrow = lane & 15, kq = lane >> 4;        // fragment row / k quarter of this lane
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    const int G = p.KH * p.nchunk;                     // (kh, chunk) groups per tile
    const int nchunk = p.nchunk;
    const i32x4 w_srd = make_srd(p.w, p.w_bytes);

    // ---- B staging: piece d of this wave fills LDS positions P = 64 d + 8 wave + lr; position 128 wn' + 16 j + c holds weight
    // row 128 wn' + 8 c + j (the column permutation that makes the epilogue's stores contiguous)
    // (NW = 4: position 64 wn' + 16 j + c holds weight row 64 wn' + 4 c + j)
    unsigned wrow_off[NBP];
#pragma unroll
    for (int d = 0; d < NBP; ++d) {
        const int P = d * 64 + wave * 8 + lr;
        const int nrow = NW == 8 ? (P >> 7) * 128 + 8 * (P & 15) + ((P >> 4) & 7) : (P >> 6) * 64 + 4 * (P & 15) + ((P >> 4) & 3);
        wrow_off[d] = (unsigned)nrow * (unsigned)p.Kbytes + (unsigned)sc * 16u;
    }
    // B fragment read address of this lane inside a stage (position 16 NW wn + 16 j + lrow: + j * 2048), k half 0; k half 1 = ^ 64
    const unsigned b_lane = (unsigned)((wn * (16 * NW) + lrow) * 128 + ((kq ^ (lrow & 7)) << 4));

    // ---- per-tile state ------------------------------------------------------------------------------------------------
    // staging cursor (the tile whose halos are being staged): its group, first row, halo rows of this lane

    unsigned hbase[MI];
    int hiy[MI];
    i32x4 in_srd = w_srd;
    int st_Hin = 1, st_row_b = 0;
    auto stage_tile = [&](int T) {                     // T uniform; T >= ntiles: nothing to stage (zeros)
        if (T >= p.ntiles) {
#pragma unroll
            for (int i = 0; i < MI; ++i) { hiy[i] = -(1 << 28); hbase[i] = 0; }
            return;
        }
        int gi = 0;
#pragma unroll
        for (int i = 1; i < RTN_MAX_GROUPS; ++i)
            if (i < p.ngroups && T >= p.g[i].tile_begin) gi = i;

        const H8Group& Gs = p.g[gi];
        const int m0 = (T - Gs.tile_begin) * TM;
        const int cells = Gs.Hin * Gs.Win;
        in_srd = make_srd(Gs.in, Gs.in_bytes);
        st_Hin = Gs.Hin;
        st_row_b = Gs.in_row_stride_b;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int h = (i * 8 + wave) * 8 + lr;
            const int f = m0 + h - p.pad_l;
            if (h < R - 1 && f >= 0 && f < Gs.M) {
                int b, rem, y, x;
                divmod24(f, cells, Gs.inv_cells, b, rem);
                divmod24(rem, Gs.Win, Gs.inv_w, y, x);
                hiy[i] = y;
                hbase[i] = (unsigned)f * (unsigned)p.pix_b + (unsigned)b * (unsigned)Gs.in_img_pad_b + (unsigned)sc * 16u;
            } else {
                hiy[i] = -(1 << 28);
                hbase[i] = 0;
            }
        }
    };
    // one halo piece (i) of group (kh, cc) of the staging tile into halo buffer `abuf_addr`
    auto stage_a = [&](int i, int kh, int cc, unsigned abuf_addr) {
        const int dy = kh - p.pad_t;
        const unsigned delta = (unsigned)(dy * st_row_b + cc * 128);
        const bool ok = (unsigned)(hiy[i] + dy) < (unsigned)st_Hin;
#if !(RTN_H8_ABLATE & 2)          // timing ablations (wrong results): -DRTN_H8_ABLATE=1 no weight staging, 2 no halo staging, 4 no MFMAs
        dma16(in_srd, ok ? hbase[i] + delta : OOB, 0u, lds_base + abuf_addr + (unsigned)(wave * 1024 + i * 8192));
#endif
    };
    // one B piece (d) of the K column block `kcol` (bytes) into ring stage `bst`
    auto stage_b = [&](int d, unsigned kcol, int bst) {
#if !(RTN_H8_ABLATE & 1)
        dma16(w_srd, wrow_off[d], kcol, lds_base + (unsigned)bst * B_STAGE + (unsigned)(wave * 1024 + d * 8192));
#endif
    };

    // compute tile: A fragment read offsets per tap (k half 0; half 1 = ^ 64), the zero row where the tap leaves the image.
    // a_cur = LDS offset of the halo buffer the current group reads (toggles per group, also across tiles)
    unsigned a_cur = A_BASE;
    unsigned arow[KW][MI];
    auto compute_tile = [&](int T, int& gi_out, int& m0_out) {
        int gi = 0;
#pragma unroll
        for (int i = 1; i < RTN_MAX_GROUPS; ++i)
            if (i < p.ngroups && T >= p.g[i].tile_begin) gi = i;
        const H8Group& Gc = p.g[gi];
        const int m0 = (T - Gc.tile_begin) * TM;
        const int cells = Gc.Hin * Gc.Win;
        gi_out = gi;
        m0_out = m0;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rloc = wm * (16 * MI) + i * 16 + lrow;
            const int m = m0 + rloc;
            const int mc = m < Gc.M ? m : Gc.M - 1;
            int b, rem, y, x;
            divmod24(mc, cells, Gc.inv_cells, b, rem);
            divmod24(rem, Gc.Win, Gc.inv_w, y, x);
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int rr = rloc + kw;
                const bool off = rloc >= TM || (unsigned)(x + kw - p.pad_l) >= (unsigned)Gc.Win;
                arow[kw][i] = a_cur + (off ? ZERO_ROW : (unsigned)(rr * 128 + ((kq ^ (rr & 7)) << 4)));
            }
        }
    };

    f32x4 acc[MI][NW];
    float bias8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias8[j] = 0.f;
    if (!SPLIT && p.bias && wn * (16 * NW) + NW * lrow < p.N) {     // columns past N (N = 192 on the 256-column tile, ...): zero, never read
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + wn * (16 * NW) + NW * lrow);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
        if (NW == 8) {
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + wn * 128 + 8 * lrow + 4);
            bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
    }

    // ---- work items.  Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch), and neighbouring row tiles read each
    // other's halo rows (kh = 0 / 2 reach one image row up / down): item(v) hands every XCD a CONTIGUOUS range of items, so that
    // those re-reads hit the XCD's own L2 instead of fetching the rows again from HBM (measured with FETCH_SIZE: 274 MB per
    // head-tower launch against 93 MB of input before).  v -> item is a bijection on [0, nitems) for any grid size.
    auto item_of = [&](int v) {
        if (v >= p.nitems || !p.xcd) return v;
        const int x = v & 7, j = v >> 3, base = p.nitems >> 3, rem = p.nitems & 7;
        return x * base + (x < rem ? x : rem) + j;
    };
    // item -> row tile, byte offset of its column block inside the weights, first (kh, chunk) group of its K slice
    auto item_decode = [&](int it, int& rt, int& cb, int& sl, unsigned& cboff, int& kh0, int& cc0) {
        if (!SPLIT || it >= p.nitems) { rt = it >= p.nitems ? p.ntiles : it; cb = 0; sl = 0; cboff = 0u; kh0 = 0; cc0 = 0; return; }
        const int per = p.ncb * p.S;
        rt = it / per;
        const int rem = it - rt * per;
        cb = rem / p.S;
        sl = rem - cb * p.S;
        cboff = (unsigned)cb * 256u * (unsigned)p.Kbytes;
        const int g0 = sl * p.gps;
        if (p.kh_fast) { cc0 = g0 / p.KH; kh0 = g0 - cc0 * p.KH; }
        else { kh0 = g0 / nchunk; cc0 = g0 - kh0 * nchunk; }
    };
    const int gps = SPLIT ? p.gps : G;                 // groups per item

    // ---- prologue: halo of the first group, B tiles of steps 0 and 1 (the last piece of step 1 goes out in step 0, phase 1)
    int vidx = blockIdx.x;
    int rt, cb, sl, kh0, cc0;
    unsigned cboff;
    item_decode(item_of(vidx), rt, cb, sl, cboff, kh0, cc0);
    stage_tile(rt);
#pragma unroll
    for (int i = 0; i < MI; ++i) stage_a(i, kh0, cc0, A_BASE);
    {
        const unsigned kc0 = cboff + (unsigned)((kh0 * KW * nchunk + cc0) * 128);
#pragma unroll
        for (int d = 0; d < NBP; ++d) stage_b(d, kc0, 0);
#pragma unroll
        for (int d = 0; d < NBP - 1; ++d) stage_b(d, kc0 + (unsigned)(nchunk * 128), 1);
    }
    if (NW == 8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else         asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();


#if RTN_H8_ABLATE & 8              // timing ablation: no fragment reads (the MFMAs multiply whatever the registers hold)
#define H8_LDA(KWI, KS)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                                \
        asm volatile("" : "=v"(fa[i_].x), "=v"(fa[i_].y), "=v"(fa[i_].z), "=v"(fa[i_].w));
#define H8_LDB(KWI, KS, HALF)                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
        asm volatile("" : "=v"(fb[j_].x), "=v"(fb[j_].y), "=v"(fb[j_].z), "=v"(fb[j_].w));
#else
#define H8_LDA(KWI, KS)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                                \
        fa[i_] = *reinterpret_cast<const uint4*>(lds + (arow[KWI][i_] ^ ((KS) * 64u)));
#define H8_LDB(KWI, KS, HALF)                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
        fb[j_] = *reinterpret_cast<const uint4*>(lds + (b_lane ^ ((KS) * 64u)) + (KWI) * B_STAGE + ((HALF) * 4 + j_) * 2048);
#endif
#if RTN_H8_ABLATE & 4
#define H8_MMA(I_, J_, HALF) asm volatile("" :: "v"(fa[I_].x), "v"(fa[I_].w), "v"(fb[J_].x), "v"(fb[J_].w));
#else
#define H8_MMA(I_, J_, HALF)                                                                         \
    acc[I_][(HALF) * 4 + J_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                              \
        __builtin_bit_cast(bf16x8, fa[I_]), __builtin_bit_cast(bf16x8, fb[J_]), acc[I_][(HALF) * 4 + J_], 0, 0, 0);
#endif
#define H8_MFMA(HALF)                                                                                \
    H8_STAMP(1)                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                    \
    H8_STAMP(2)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                            \
            H8_MMA(i_, j_, HALF)                                                                     \
    __builtin_amdgcn_s_setprio(0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    H8_STAMP(3)                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                    \
    H8_STAMP(4)
    // One K step = tap KWI of the current group.  DMA slots (see the header): phase 1 the LAST piece of step s+1's B tile,
    // phases 2-4 the first three pieces of step s+2's, halo pieces of the next group in phases 2 and 4 of taps 0 and 1.
    // kc_n1 / kc_n2: K column blocks of steps s+1 / s+2.
#define H8_STEP(KWI)                                                                                 \
    {                                                                                                \
        const unsigned kc_n1 = (KWI) + 1 < KW ? kcol_g + ((KWI) + 1) * kw_stride : kcol_g1;          \
        const unsigned kc_n2 = (KWI) + 2 < KW ? kcol_g + ((KWI) + 2) * kw_stride : kcol_g1 + ((KWI) + 2 - KW) * kw_stride; \
        uint4 fa[MI], fb[4];                                                                         \
        /* phase 1 */                                                                                \
        H8_LDA(KWI, 0) H8_LDB(KWI, 0, 0)                                                             \
        stage_b(3, kc_n1, ((KWI) + 1) % 3);                                                          \
        H8_MFMA(0)                                                                                   \
        /* phase 2 */                                                                                \
        H8_LDB(KWI, 0, 1)                                                                            \
        stage_b(0, kc_n2, ((KWI) + 2) % 3);                                                          \
        if (2 * (KWI) < MI && (KWI) < 2) stage_a(2 * (KWI), kh1, cc1, a_cur ^ A_TOGGLE);                 \
        H8_MFMA(1)                                                                                   \
        /* phase 3 */                                                                                \
        H8_LDA(KWI, 1) H8_LDB(KWI, 1, 0)                                                             \
        stage_b(1, kc_n2, ((KWI) + 2) % 3);                                                          \
        H8_MFMA(0)                                                                                   \
        /* phase 4 */                                                                                \
        H8_LDB(KWI, 1, 1)                                                                            \
        stage_b(2, kc_n2, ((KWI) + 2) % 3);                                                          \
        if (2 * (KWI) + 1 < MI && (KWI) < 2) stage_a(2 * (KWI) + 1, kh1, cc1, a_cur ^ A_TOGGLE);         \
        /* B(s+1) must have landed: allowed in flight = what this step issued after its phase 1 (3 B pieces + its halo pieces) */ \
        {                                                                                            \
            constexpr int na_ = (KWI) < 2 ? ((2 * (KWI) < MI) + (2 * (KWI) + 1 < MI)) : 0;           \
            if (na_ == 2)      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                      \
            else if (na_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                      \
            else               asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                      \
        }                                                                                            \
        H8_MFMA(1)                                                                                   \
    }
    // The fp8 step: phase A = column fragments 0-3 (A fragments of BOTH k halves are read here and kept), phase B = fragments 4-7.
    // DMA slots: the last piece of step s+1's B tile in phase A, the first three of step s+2's in phase B (their ring slot was last
    // read in phase B of the step before, by the lagging wave group during this step's phase A).
#define H8_MFMA8(HALF)                                                                               \
    __builtin_amdgcn_s_barrier();                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                            \
            mma_fp8(acc[i_][(HALF) * 4 + j_], fa[i_], fa1[i_], fb[j_], fb1[j_]);                      \
    __builtin_amdgcn_s_setprio(0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_barrier();
#define H8_STEP8F(KWI)                                                                               \
    {                                                                                                \
        const unsigned kc_n1 = (KWI) + 1 < KW ? kcol_g + ((KWI) + 1) * kw_stride : kcol_g1;          \
        const unsigned kc_n2 = (KWI) + 2 < KW ? kcol_g + ((KWI) + 2) * kw_stride : kcol_g1 + ((KWI) + 2 - KW) * kw_stride; \
        uint4 fa[MI], fa1[MI], fb[4], fb1[4];                                                        \
        H8_LDA(KWI, 0)                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                            \
            fa1[i_] = *reinterpret_cast<const uint4*>(lds + (arow[KWI][i_] ^ 64u));                  \
        H8_LDB(KWI, 0, 0)                                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                             \
            fb1[j_] = *reinterpret_cast<const uint4*>(lds + (b_lane ^ 64u) + (KWI) * B_STAGE + j_ * 2048); \
        stage_b(3, kc_n1, ((KWI) + 1) % 3);                                                          \
        if (2 * (KWI) < MI && (KWI) < 2) stage_a(2 * (KWI), kh1, cc1, a_cur ^ A_TOGGLE);             \
        H8_MFMA8(0)                                                                                  \
        H8_LDB(KWI, 0, 1)                                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                             \
            fb1[j_] = *reinterpret_cast<const uint4*>(lds + (b_lane ^ 64u) + (KWI) * B_STAGE + (4 + j_) * 2048); \
        stage_b(0, kc_n2, ((KWI) + 2) % 3);                                                          \
        stage_b(1, kc_n2, ((KWI) + 2) % 3);                                                          \
        stage_b(2, kc_n2, ((KWI) + 2) % 3);                                                          \
        if (2 * (KWI) + 1 < MI && (KWI) < 2) stage_a(2 * (KWI) + 1, kh1, cc1, a_cur ^ A_TOGGLE);     \
        {                                                                                            \
            constexpr int na_ = (KWI) < 2 ? ((2 * (KWI) < MI) + (2 * (KWI) + 1 < MI)) : 0;           \
            if (na_ == 2)      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                      \
            else if (na_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                      \
            else               asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                      \
        }                                                                                            \
        H8_MFMA8(1)                                                                                  \
    }
    // The half-width step: phase A = k half 0 + the LAST piece of step s+1's B tile, phase B = k half 1 + the FIRST piece of step
    // s+2's (its ring slot was last read two phases ago), halo pieces of the next group in both phases of taps 0 and 1.
#define H8_STEP4(KWI)                                                                                \
    {                                                                                                \
        const unsigned kc_n1 = (KWI) + 1 < KW ? kcol_g + ((KWI) + 1) * kw_stride : kcol_g1;          \
        const unsigned kc_n2 = (KWI) + 2 < KW ? kcol_g + ((KWI) + 2) * kw_stride : kcol_g1 + ((KWI) + 2 - KW) * kw_stride; \
        uint4 fa[MI], fb[4];                                                                         \
        H8_LDA(KWI, 0) H8_LDB(KWI, 0, 0)                                                             \
        stage_b(1, kc_n1, ((KWI) + 1) % 3);                                                          \
        if (2 * (KWI) < MI && (KWI) < 2) stage_a(2 * (KWI), kh1, cc1, a_cur ^ A_TOGGLE);             \
        H8_MFMA(0)                                                                                   \
        H8_LDA(KWI, 1) H8_LDB(KWI, 1, 0)                                                             \
        stage_b(0, kc_n2, ((KWI) + 2) % 3);                                                          \
        if (2 * (KWI) + 1 < MI && (KWI) < 2) stage_a(2 * (KWI) + 1, kh1, cc1, a_cur ^ A_TOGGLE);     \
        {                                                                                            \
            constexpr int na_ = (KWI) < 2 ? ((2 * (KWI) < MI) + (2 * (KWI) + 1 < MI)) : 0;           \
            if (na_ == 2)      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                      \
            else if (na_ == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                      \
            else               asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                      \
        }                                                                                            \
        H8_MFMA(0)                                                                                   \
    }
    static_assert(KW == 3 && MI >= 2 && MI <= 4, "the B ring (3 stages) and the halo piece slots are laid out for KW = 3, MI <= 4");

    const unsigned kw_stride = (unsigned)(nchunk * 128);           // K bytes between the taps of a kernel row
    while (rt < p.ntiles) {
        int gi, m0;
        compute_tile(rt, gi, m0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j) acc[i][j] = (SPLIT || ES == 1) ? (f32x4){0.f, 0.f, 0.f, 0.f} : (f32x4){bias8[j], bias8[j], bias8[j], bias8[j]};
        int kh = kh0, cc = cc0;
        int rt1 = rt, cb1 = cb, sl1 = sl;
        unsigned cboff1 = cboff;
#pragma unroll 1
        for (int g = 0; g < gps; ++g) {
            // the next group (kh1, cc1): of this item, or the first group of the workgroup's next item
            int kh1 = kh, cc1 = cc;
            if (p.kh_fast) { if (++kh1 == p.KH) { kh1 = 0; ++cc1; } }
            else if (++cc1 == nchunk) { cc1 = 0; ++kh1; }
            if (g + 1 == gps) {
                vidx += (int)gridDim.x;
                item_decode(item_of(vidx), rt1, cb1, sl1, cboff1, kh1, cc1);
                stage_tile(rt1);
            }
#ifdef RTN_H8_STAMP
            stamp_on = blockIdx.x == 0 && (wave & 3) == 0 && first_item && g == 4;
#endif
            const unsigned kcol_g = cboff + (unsigned)((kh * KW * nchunk + cc) * 128);
            const unsigned kcol_g1 = cboff1 + (unsigned)((kh1 * KW * nchunk + cc1) * 128);
            if constexpr (ES == 1) {
                H8_STEP8F(0)
                H8_STEP8F(1)
                H8_STEP8F(2)
            } else if constexpr (NW == 8) {
                H8_STEP(0)
                H8_STEP(1)
                H8_STEP(2)
            } else {
                H8_STEP4(0)
                H8_STEP4(1)
                H8_STEP4(2)
            }
            // next group: the other halo buffer
#pragma unroll
            for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                for (int i = 0; i < MI; ++i) arow[kw][i] ^= A_TOGGLE;
            a_cur ^= A_TOGGLE;
            kh = kh1; cc = cc1;
        }
        kh0 = kh; cc0 = cc;                            // the next item starts at the group the cursor already points to
        // ---- epilogue: [mask] [+ residual] [mask] ReLU, bf16, 4 MI stores of 16 B per lane (rows beyond TM / M go to an out-of-range
        // offset and are dropped)
        if constexpr (ES == 1) {
            const H8Group& Gc = p.g[gi];
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)Gc.out, 0, (int)__builtin_amdgcn_readfirstlane((int)Gc.out_bytes), 0x00020000);
            const int ncol = wn * 128 + 8 * lrow;
            const bool col_ok = ncol < p.N;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rloc = wm * (16 * MI) + i * 16 + kq * 4 + r;
                    const int m = m0 + rloc;
                    const bool ok = col_ok && rloc < TM && m < Gc.M;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = acc[i][j][r] * p.acc_scale + bias8[j];
                        if (p.relu) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    }
                    if (p.out_fp8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float q = v[j] * p.out_scale;
                            v[j] = q > 448.f ? 448.f : (q < -448.f ? -448.f : q);
                        }
                        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                        u32x2 o;
                        o.x = pack_fp8x4(v[0], v[1], v[2], v[3]); o.y = pack_fp8x4(v[4], v[5], v[6], v[7]);
                        __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, (int)(ok ? (unsigned)m * (unsigned)p.out_ld + (unsigned)ncol : OOB), 0, 0);
                        asm volatile("s_nop 3" :: "v"(o.x), "v"(o.y));
                    } else {
                        u32x4 o;
                        o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
                        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, (int)(ok ? ((unsigned)m * (unsigned)p.out_ld + (unsigned)ncol) * 2u : OOB), 0, 0);
                        RTN_STORE_GUARD(o)
                    }
                }
        } else if constexpr (NW == 4) {
            // 4 consecutive channels per lane and row: 8-byte stores (and 8-byte residual / mask loads, all of a wave's in flight first)
            const H8Group& Gc = p.g[gi];
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)Gc.out, 0, (int)__builtin_amdgcn_readfirstlane((int)Gc.out_bytes), 0x00020000);
            const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 1) ? Gc.res : Gc.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 1) ? Gc.res_bytes : 0u)), 0x00020000);
            const __amdgpu_buffer_rsrc_t mask_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 2) ? Gc.mask : Gc.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 2) ? Gc.mask_bytes : 0u)), 0x00020000);
            const int ncol = wn * 64 + 4 * lrow;
            const bool col_ok = ncol < p.N;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 rq[(EPI & 1) ? MI : 1][4], mq[(EPI & 2) ? MI : 1][4];
            if (EPI) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rloc = wm * (16 * MI) + i * 16 + kq * 4 + r;
                        const int m = m0 + rloc;
                        const bool ok = col_ok && rloc < TM && m < Gc.M;
                        if (EPI & 1) rq[i][r] = __builtin_amdgcn_raw_buffer_load_b64(res_rsrc, (int)(ok ? ((unsigned)m * (unsigned)p.res_ld + (unsigned)ncol) * 2u : OOB), 0, 0);
                        if (EPI & 2) mq[i][r] = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc, (int)(ok ? ((unsigned)m * (unsigned)p.mask_ld + (unsigned)ncol) * 2u : OOB), 0, 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rloc = wm * (16 * MI) + i * 16 + kq * 4 + r;
                    const int m = m0 + rloc;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[i][j][r];
                    if (EPI) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const unsigned mj = (EPI & 2) ? mq[(EPI & 2) ? i : 0][r][j] : 0x3f803f80u, rj = (EPI & 1) ? rq[(EPI & 1) ? i : 0][r][j] : 0u;
                            const bool keep_lo = __uint_as_float(mj << 16) > 0.f, keep_hi = __uint_as_float(mj & 0xffff0000u) > 0.f;
                            if ((EPI & 2) && p.mask_pre) { if (!keep_lo) v[2 * j] = 0.f; if (!keep_hi) v[2 * j + 1] = 0.f; }
                            if (EPI & 1) { v[2 * j] += __uint_as_float(rj << 16); v[2 * j + 1] += __uint_as_float(rj & 0xffff0000u); }
                            if ((EPI & 2) && !p.mask_pre) { if (!keep_lo) v[2 * j] = 0.f; if (!keep_hi) v[2 * j + 1] = 0.f; }
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    }
                    u32x2 o;
                    o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]);
                    const bool ok = col_ok && rloc < TM && m < Gc.M;
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.out_ld + (unsigned)ncol) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, (int)off, 0, 0);
                    asm volatile("s_nop 3" :: "v"(o.x), "v"(o.y));
                }
        } else if (SPLIT && p.S > 1) {
            // f32 partial sums of this K slice -> slab[sl][m][256 cb + channel], 2 x 16 B per lane and row
            const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((char*)p.slab + (size_t)sl * p.slab_slice_bytes), 0, (int)__builtin_amdgcn_readfirstlane((int)p.slab_slice_bytes), 0x00020000);
            const int gcol = cb * 256 + wn * 128 + 8 * lrow;
            const int Mg = p.g[gi].M;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rloc = wm * (16 * MI) + i * 16 + kq * 4 + r;
                    const int m = m0 + rloc;
                    const bool ok = gcol < p.slab_ld && rloc < TM && m < Mg;
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.slab_ld + (unsigned)gcol) * 4u : OOB;
                    u32x4 o0, o1;
                    o0.x = __float_as_uint(acc[i][0][r]); o0.y = __float_as_uint(acc[i][1][r]); o0.z = __float_as_uint(acc[i][2][r]); o0.w = __float_as_uint(acc[i][3][r]);
                    o1.x = __float_as_uint(acc[i][4][r]); o1.y = __float_as_uint(acc[i][5][r]); o1.z = __float_as_uint(acc[i][6][r]); o1.w = __float_as_uint(acc[i][7][r]);
                    __builtin_amdgcn_raw_buffer_store_b128(o0, slab_rsrc, (int)off, 0, 0);
                    RTN_STORE_GUARD(o0)
                    __builtin_amdgcn_raw_buffer_store_b128(o1, slab_rsrc, (int)(ok ? off + 16u : OOB), 0, 0);
                    RTN_STORE_GUARD(o1)
                }
        } else {
            const H8Group& Gc = p.g[gi];
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)Gc.out, 0, (int)__builtin_amdgcn_readfirstlane((int)Gc.out_bytes), 0x00020000);
            const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 1) ? Gc.res : Gc.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 1) ? Gc.res_bytes : 0u)), 0x00020000);
            const __amdgpu_buffer_rsrc_t mask_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 2) ? Gc.mask : Gc.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 2) ? Gc.mask_bytes : 0u)), 0x00020000);
            const int ncol = (SPLIT ? cb * 256 : 0) + wn * 128 + 8 * lrow;
            const bool col_ok = ncol < p.N;
            float bias_e[8];                        // SPLIT: the accumulators started at zero
#pragma unroll
            for (int j = 0; j < 8; ++j) bias_e[j] = 0.f;
            if (SPLIT && p.bias && col_ok) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + ncol);
                const float4 b1 = *reinterpret_cast<const float4*>(p.bias + ncol + 4);
                bias_e[0] = b0.x; bias_e[1] = b0.y; bias_e[2] = b0.z; bias_e[3] = b0.w;
                bias_e[4] = b1.x; bias_e[5] = b1.y; bias_e[6] = b1.z; bias_e[7] = b1.w;
            }
            // residual / mask rows of the row fragments: with ONE of the two (the data gradients of the towers, P3, res4 branch2b take
            // the mask only) every fragment's rows are requested before the first is used - one round trip per tile instead of one per
            // fragment (the loop's fragment and address registers are dead here); with both, two fragments are in flight (ping-pong)
            constexpr int EDEPTH = (EPI == 3 || (EPI == 1 && MI == 4)) ? 2 : MI;     // (residual only, 256 rows: four fragments in flight spill)
            u32x4 rq[EDEPTH][4], mq[EDEPTH][4];
            auto fetch = [&](int i, int par) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rloc = wm * (16 * MI) + i * 16 + kq * 4 + r;
                    const int m = m0 + rloc;
                    const bool ok = col_ok && rloc < TM && m < Gc.M;
                    if (EPI & 1) rq[par][r] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, (int)(ok ? ((unsigned)m * (unsigned)p.res_ld + (unsigned)ncol) * 2u : OOB), 0, 0);
                    if (EPI & 2) mq[par][r] = __builtin_amdgcn_raw_buffer_load_b128(mask_rsrc, (int)(ok ? ((unsigned)m * (unsigned)p.mask_ld + (unsigned)ncol) * 2u : OOB), 0, 0);
                }
            };
            if (EPI) {
#pragma unroll
                for (int i = 0; i < EDEPTH - 1; ++i) fetch(i, i);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (EPI && i + EDEPTH - 1 < MI) fetch(i + EDEPTH - 1, (i + EDEPTH - 1) % EDEPTH);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rloc = wm * (16 * MI) + i * 16 + kq * 4 + r;
                    const int m = m0 + rloc;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = acc[i][j][r] + (SPLIT ? bias_e[j] : 0.f);
                    if (EPI) {
                        const u32x4 rw = rq[i % EDEPTH][r], mw = mq[i % EDEPTH][r];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned mj = (EPI & 2) ? mw[j] : 0x3f803f80u, rj = (EPI & 1) ? rw[j] : 0u;
                            const bool keep_lo = __uint_as_float(mj << 16) > 0.f, keep_hi = __uint_as_float(mj & 0xffff0000u) > 0.f;
                            if ((EPI & 2) && p.mask_pre) { if (!keep_lo) v[2 * j] = 0.f; if (!keep_hi) v[2 * j + 1] = 0.f; }
                            if (EPI & 1) { v[2 * j] += __uint_as_float(rj << 16); v[2 * j + 1] += __uint_as_float(rj & 0xffff0000u); }
                            if ((EPI & 2) && !p.mask_pre) { if (!keep_lo) v[2 * j] = 0.f; if (!keep_hi) v[2 * j + 1] = 0.f; }
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    }
                    u32x4 o;
                    o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
                    const bool ok = col_ok && rloc < TM && m < Gc.M;
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.out_ld + (unsigned)ncol) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, (int)off, 0, 0);
                    RTN_STORE_GUARD(o)
                }
            }
        }
        rt = rt1; cb = cb1; sl = sl1; cboff = cboff1;
#ifdef RTN_H8_STAMP
        first_item = false;
#endif
        // the staging cursor's halo rows already belong to the next item (switched in the last group); the pieces it issued for an
        // item past the end are zeros from out-of-range offsets
    }
#undef H8_STEP4
#undef H8_STEP8F
#undef H8_MFMA8
#undef H8_STEP
#undef H8_MFMA
#undef H8_MMA
#undef H8_LDB
#undef H8_LDA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS
    if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();
}

}  // namespace

// Launcher.  Returns RTN_OK after a launch, 1 when the layer is not one this kernel takes (the caller falls through to the other
// kernels), < 0 on a launch error.  `forced`: take every eligible layer (tests), otherwise the caller decides.  `ws` / `ws_cap`: the
// caller-owned workspace (K-slice slabs); `query` != nullptr: no launch, *query = workspace bytes this layer would use.
// `ksplit_force`: 0 = cost model, 1 = never slice, n > 1 = n slices when n divides the group count.
int rtn_conv_halo8_try(rtn_handle_t h, const rtn_conv_desc_t* d, int grid_limit, bool stagger, bool forced, int mi_force, float* ws,
                       long long ws_cap, size_t* query, int ksplit_force, const rtn_conv_fp8_t* q8) {
    if (query) *query = 0;
    if (q8 ? d->dtype != RTN_FP8 : d->dtype != RTN_BF16) return 1;
    const int es = q8 ? 1 : 2;                         // bytes per input / weight element
    if (q8 && (query || (d->flags & ~RTN_CONV_RELU) || d->N <= 128 || d->N > 256)) return 1;     // fp8: one full-width column block, bias / ReLU
    if (d->KW != 3 || d->KH < 1 || d->KH > 7 || d->sy != 1 || d->sx != 1) return 1;
    if (d->flags & ~(RTN_CONV_RELU | RTN_CONV_RES_SAME | RTN_CONV_RELU_MASK | RTN_CONV_MASK_PRE)) return 1;
    if ((d->flags & RTN_CONV_MASK_PRE) && !(d->flags & RTN_CONV_RELU_MASK)) return 1;
    const int epi = ((d->flags & RTN_CONV_RES_SAME) ? 1 : 0) | ((d->flags & RTN_CONV_RELU_MASK) ? 2 : 0);
    const int ncb = (d->N + 255) / 256;                // column blocks of 256 channels
    if (d->N <= 64 || ncb > 8 || d->w_rows < d->N || d->N % 8 || d->out_ld % 8) return 1;    // weight rows past w_rows read as zeros (descriptor range)
    if (ncb > 1 && epi) return 1;
    const bool half = d->N <= 128;                     // the 128-column instance (NW = 4): plain epilogue, no slices
    if (half && !forced && rtn_env_int("RTN_CONV_H8_HALF", 1) == 0) return 1;
    if (d->Crun != d->pix_stride || (d->Crun * es) % 128 || d->Crun <= 0) return 1;
    if (d->pad_l < 0 || d->pad_l >= d->KW || d->pad_t < 0 || d->pad_t >= d->KH) return 1;
    if (((uintptr_t)d->w & 15) || ((uintptr_t)d->bias & 15)) return 1;
    const int nchunk = d->Crun * es / 128;
    const int G = d->KH * nchunk;
    const long long Kbytes = (long long)d->KH * d->KW * d->Crun * es;
    if (Kbytes * 256 * ncb >= 0xFFFFFF00ll) return 1;                  // the staging offsets of the last column block
    H8Params p;
    memset(&p, 0, sizeof(p));
    const int cus = h->num_cus > 0 ? h->num_cus : 256;
    long long Mtot = 0;
    for (int i = 0; i < d->ngroups; ++i) Mtot += (long long)d->g[i].Hout * d->g[i].Wout * d->batch;
    // Tile height (rows = 64 mi) and K slices by a cost model in microseconds: rounds of workgroups x K steps of an item (+ 5 for its
    // prologue / epilogue) x 1.53 us per 256-row step, + for S > 1 the finish launch and the slabs' round trip at 4 TB/s.
    // RTN_CONV_H8_MI / RTN_CONV_H8_KSPLIT pin them (A/B, tests).
    const bool can_split = !half && !q8 && d->ngroups == 1 && epi == 0 && ksplit_force != 1 && (ws_cap > 0 || query);
    const long long slab_ld = 256ll * ncb;
    int mi = 0, S = 1;
    {
        double best = 0;
        for (int cand = 4; cand >= 3; --cand) {
            if (!q8 && mi_force >= 3 && mi_force <= 4 && cand != mi_force) continue;
            if (q8 && cand == 4) continue;                 // fp8 holds both k halves' A fragments: the 256-row instance spills, 192 rows fit (256 VGPRs)
            long long t = 0;
            for (int i = 0; i < d->ngroups; ++i) t += ((long long)d->g[i].Hout * d->g[i].Wout * d->batch + 64 * cand - 4) / (64 * cand - 3);
            const double step_us = 1.53 * (cand + 0.3) / 4.3;
            for (int sc = 1; sc <= 8; ++sc) {
                if (sc > 1 && (!can_split || G % sc || (ksplit_force > 1 && sc != ksplit_force))) continue;
                if (sc == 1 && ksplit_force > 1 && can_split && G % ksplit_force == 0) continue;
                if (sc > 1 && (double)sc * Mtot * slab_ld * 4 > (double)(query ? (256ll << 20) : ws_cap)) continue;
                const long long items = t * ncb * sc;
                double cost = (double)((items + cus - 1) / cus) * (3.0 * G / sc + 5.0) * step_us;
                if (sc > 1) cost += 4.0 + ((double)sc * Mtot * slab_ld * 4 + (double)Mtot * d->N * 2) / 4.0e6;
                if (mi == 0 || cost < best * 0.97) { best = cost; mi = cand; S = sc; }
            }
        }
        if (mi == 0) return 1;
    }
    if (S > 1 && !query && (!ws || ((uintptr_t)ws & 15))) return 1;
    const bool split = ncb > 1 || S > 1;
    const int TM = 64 * mi - 3;
    long long tiles = 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (s.Hin != s.Hout || s.Win != s.Wout || s.in_row_stride != (long long)s.Win * d->pix_stride ||
            s.in_img_stride < (long long)s.Hin * s.in_row_stride) return 1;
        if (s.in_elems < (long long)(d->batch - 1) * s.in_img_stride + (long long)s.Hin * s.in_row_stride) return 1;
        const long long cells = (long long)s.Hout * s.Wout, M = cells * d->batch;
        if (s.out_step > 1 || s.out_off != 0 || s.out_img_stride != cells * d->out_ld) return 1;      // dense [M][out_ld] output
        if (M >= (1ll << 24) || M < 1) return 1;
        if (!s.in || !s.out || ((uintptr_t)s.in & 15) || ((uintptr_t)s.out & 15)) return 1;
        if (s.in_elems < M * d->Crun || s.out_elems < (M - 1) * d->out_ld + d->N) return 1;
        if (s.in_elems * 2 >= 0xFFFFFF00ll || s.out_elems * 2 >= 0xFFFFFF00ll) return 1;
        if (epi & 1) {
            if (!s.res || ((uintptr_t)s.res & 15) || s.res_ld % 8 || s.res_img_stride != cells * s.res_ld) return 1;
            if (s.res_elems < (M - 1) * s.res_ld + d->N || s.res_elems * 2 >= 0xFFFFFF00ll) return 1;
        }
        if (epi & 2) {
            if (!s.mask || ((uintptr_t)s.mask & 15) || s.mask_ld % 8 || s.mask_img_stride != cells * s.mask_ld) return 1;
            if (s.mask_elems < (M - 1) * s.mask_ld + d->N || s.mask_elems * 2 >= 0xFFFFFF00ll) return 1;
        }
        if (i > 0 && ((epi & 1) && s.res_ld != d->g[0].res_ld)) return 1;
        if (i > 0 && ((epi & 2) && s.mask_ld != d->g[0].mask_ld)) return 1;
        H8Group& g = p.g[i];
        g.res = (epi & 1) ? (const char*)s.res : nullptr;
        g.mask = (epi & 2) ? (const char*)s.mask : nullptr;
        g.res_bytes = (epi & 1) ? (unsigned)(s.res_elems * 2) : 0u;
        g.mask_bytes = (epi & 2) ? (unsigned)(s.mask_elems * 2) : 0u;
        g.in = (const char*)s.in;
        g.out = (char*)s.out;
        g.in_bytes = (unsigned)(s.in_elems * es);
        g.out_bytes = (unsigned)(s.out_elems * ((q8 && q8->out_dtype == RTN_FP8) ? 1 : 2));
        g.Hin = s.Hin; g.Win = s.Win; g.M = (int)M;
        g.tile_begin = (int)tiles;
        g.in_row_stride_b = (int)(s.in_row_stride * es);
        g.in_img_pad_b = (int)((s.in_img_stride - (long long)s.Hin * s.in_row_stride) * es);
        g.inv_cells = 1.0f / (float)cells;
        g.inv_w = 1.0f / (float)s.Win;
        tiles += (M + TM - 1) / TM;
    }
    const long long items = tiles * ncb * S;
    if (tiles < 1 || items > 0x3fffffff) return 1;
    // one 256 x 256 tile per CU: below a quarter of the chip the narrow tiles of generation 2 (four times the workgroups) are
    // faster (P5 at batch 8 unsliced: 34 tiles, 0.049 ms here against 0.037 ms); from half the chip on this kernel wins (res4 3x3,
    // P4: 133 tiles, 0.054 against 0.067 ms)
    if (!forced && items * 4 < cus) return 1;
    const long long slice_bytes = Mtot * slab_ld * 4;
    if (S > 1 && slice_bytes >= 0xFFFFFF00ll) return 1;
    if (query) { *query = S > 1 ? (size_t)(slice_bytes * S) : 0; return RTN_OK; }
    p.w = (const char*)d->w;
    p.bias = d->bias;
    p.w_bytes = (unsigned)(Kbytes * d->w_rows);
    p.ngroups = d->ngroups;
    p.ntiles = (int)tiles;
    p.N = d->N;
    p.Kbytes = (int)Kbytes;
    p.KH = d->KH;
    p.nchunk = nchunk;
    p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.relu = (d->flags & RTN_CONV_RELU) ? 1 : 0;
    p.out_ld = d->out_ld;
    p.pix_b = d->pix_stride * es;
    if (q8) { p.acc_scale = q8->acc_scale; p.out_scale = q8->out_scale; p.out_fp8 = q8->out_dtype == RTN_FP8 ? 1 : 0; }
    p.res_ld = (epi & 1) ? d->g[0].res_ld : 0;
    p.mask_ld = (epi & 2) ? d->g[0].mask_ld : 0;
    p.mask_pre = (d->flags & RTN_CONV_MASK_PRE) ? 1 : 0;
    p.ncb = ncb; p.S = S; p.gps = G / S; p.nitems = (int)items;
    p.xcd = 1;                                         // XCD-contiguous item order (0 = workgroup b starts at item b: round-2 A/B, 274 -> 202 MB fetched per tower launch)
    p.kh_fast = 1;                                     // the kernel rows of a channel chunk back to back (round-2 A/B: profiles/r2_v3_ab_xcd_item_order.txt)
    p.slab = S > 1 ? ws : nullptr;
    p.slab_slice_bytes = S > 1 ? (unsigned)slice_bytes : 0u;
    p.slab_ld = (int)slab_ld;
    int grid = cus;
    if (grid_limit > 0 && grid_limit < grid) grid = grid_limit;
    if (grid > p.nitems) grid = p.nitems;
#define RTN_H8_LAUNCH(M_, ST, EP, SP)                                                                    \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_halo8_kernel<3, M_, ST, EP, SP>,            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, H8_LDS));         \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_halo8_kernel<3, M_, ST, EP, SP>), dim3((unsigned)grid), dim3(H8_THREADS), H8_LDS, h->stream, p); \
    } while (0)
#define RTN_H8_LAUNCH4E(M_, EP)                                                                          \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_halo8_kernel<3, M_, true, EP, false, 4>,    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, H8_LDS));         \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_halo8_kernel<3, M_, true, EP, false, 4>), dim3((unsigned)grid), dim3(H8_THREADS), H8_LDS, h->stream, p); \
    } while (0)
#define RTN_H8_LAUNCH4(M_)                                                                               \
    do {                                                                                                 \
        if (epi == 0) RTN_H8_LAUNCH4E(M_, 0); else if (epi == 1) RTN_H8_LAUNCH4E(M_, 1);                 \
        else if (epi == 2) RTN_H8_LAUNCH4E(M_, 2); else RTN_H8_LAUNCH4E(M_, 3);                          \
    } while (0)
#define RTN_H8_LAUNCH8F(M_)                                                                              \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_halo8_kernel<3, M_, true, 0, false, 8, 1>,  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, H8_LDS));         \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_halo8_kernel<3, M_, true, 0, false, 8, 1>), dim3((unsigned)grid), dim3(H8_THREADS), H8_LDS, h->stream, p); \
    } while (0)
#define RTN_H8_PICK(M_)                                                                                  \
    do {                                                                                                 \
        if (q8) RTN_H8_LAUNCH8F(3);                                                                      \
        else if (half) RTN_H8_LAUNCH4(M_);                                                               \
        else if (split) RTN_H8_LAUNCH(M_, true, 0, true);                                                \
        else if (!stagger && epi == 0) RTN_H8_LAUNCH(M_, false, 0, false);      /* lockstep variant: A/B only */ \
        else if (epi == 0) RTN_H8_LAUNCH(M_, true, 0, false);                                            \
        else if (epi == 1) RTN_H8_LAUNCH(M_, true, 1, false);                                            \
        else if (epi == 2) RTN_H8_LAUNCH(M_, true, 2, false);                                            \
        else RTN_H8_LAUNCH(M_, true, 3, false);                                                          \
    } while (0)
    if (mi == 4) RTN_H8_PICK(4); else RTN_H8_PICK(3);
#undef RTN_H8_PICK
#undef RTN_H8_LAUNCH8F
#undef RTN_H8_LAUNCH4
#undef RTN_H8_LAUNCH4E
#undef RTN_H8_LAUNCH
    RTN_CHECK_LAUNCH(h, "conv_halo8_kernel");
    h->last_conv_tile = ((64 * mi) << 16) | (half ? 128 : 256);
    if (S > 1) return rtn_conv_ksplit_finish(h, ws, S, Mtot, d->N, (int)slab_ld, d->bias, p.relu, d->g[0].out, d->out_ld);
    return RTN_OK;
}

#ifdef RTN_H8_STAMP
extern "C" int rtn_debug_h8_stamps(unsigned long long* out128) {
    return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_h8_stamps), sizeof(unsigned long long) * 128, 0, hipMemcpyDeviceToHost);
}
#endif
