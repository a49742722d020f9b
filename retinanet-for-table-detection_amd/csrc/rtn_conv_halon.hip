// rtn_conv_halon.hip — the head OUTPUT convolutions: stride-1 3x3 'same' convolution with at most 48 output channels and an f32
// result written into the concatenated (B, anchors, 4 | classes) tensor:
//   model/defineModel.py:111-123   pyramid_classification (3x3, 256 -> 9 anchors x classes, sigmoid),
//   model/defineModel.py:163-166   pyramid_regression     (3x3, 256 -> 9 x 4),
//   model/defineModel.py:208-228   applied to P3..P7 and concatenated along the anchor axis (one grouped launch per head).
//
// These layers are 0.03 / 0.007 TFLOP on 91 MB of input per launch (batch 8, 800x1333): they are bound by moving pixels, not by
// MFMA.  The wide kernels (rtn_conv.hip generations 1-3) run them as 64-column tiles of which 9 or 36 columns are real and
// restage every tap.  Here:
//   * persistent, one workgroup (8 waves) per CU walking 253-pixel tiles as ONE stream of (kernel row, 64-channel chunk) groups;
//   * a group's stage = the 256-pixel halo image of that kernel row (32 KiB, shared by its 3 taps, zero row for the horizontal
//     image edge as in rtn_conv_halo8.hip) + the 3 x 16 NF weight rows of its taps (6 NF KiB); ring of 3 stages filled by LDS-DMA
//     TWO groups ahead, one counted s_waitcnt and ONE s_barrier per group (24 NF MFMAs per wave between barriers);
//   * transposed products: the weight rows are the MFMA A operand, the pixels its columns, so a lane ends up with 4 CONSECUTIVE
//     output channels of one pixel: one 16-byte f32 store per fragment straight from registers (dword stores when the channel
//     count is not a multiple of 4), bias as the accumulators' initial value, sigmoid in registers.
// LDS: 3 x (32 KiB + 6 NF KiB) + 1 KiB that swallows the surplus DMA pieces = 151 KiB for NF = 3.
#include "rtn_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned HN_OOB = 0xFFFFFF00u;              // beyond every descriptor: loads return zeros, stores are dropped
constexpr int HN_THREADS = 512;
constexpr int HN_R = 256, HN_TM = HN_R - 3;            // halo rows / output pixels of a tile (row 255 of a halo = zeros)
constexpr unsigned HN_HALO = 32768, HN_ZERO_ROW = (HN_R - 1) * 128u;

template <int NF>
struct HNCfg {
    static constexpr int BR = 48 * NF;                // weight rows of a stage: 3 taps x 16 NF channels
    static constexpr int BP = BR / 8;                 // ... as 1-KiB DMA pieces (8 rows of 128 bytes)
    static constexpr int BPW = (BP + 7) / 8;          // pieces per wave (the surplus ones land in the dump block)
    static constexpr unsigned STAGE = HN_HALO + BR * 128u;
    static constexpr unsigned DUMP = 3 * STAGE;
    static constexpr int LDS = 3 * (int)STAGE + 1024;
    static constexpr int PP = 4 + BPW;                // DMA instructions per wave and stage
};

struct HNGroup {
    const char* in;
    char* out;
    unsigned in_bytes, out_bytes;
    int Hin, Win, M, tile_begin;
    int in_row_stride_b;
    int cells;
    unsigned out_img_stride, out_off;                 // elements
    float inv_cells, inv_w;
};

struct HNParams {
    HNGroup g[RTN_MAX_GROUPS];
    const char* w;
    const float* bias;
    unsigned w_bytes;
    int ngroups, ntiles;
    int N, Kbytes, nchunk, pad_t, pad_l, relu, sigmoid, out_ld, pix_b, vec, xcd, kh_fast;
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
    const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr), so = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(la), "s"(srd), "s"(so)
                 : "memory");
}

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
#define RTN_STORE_GUARD1(V) asm volatile("s_nop 1" :: "v"(V));

template <int NF>
__global__ __launch_bounds__(HN_THREADS, 2) void conv_halon_kernel(const HNParams p) {
    typedef HNCfg<NF> Cfg;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lr = lane >> 3, sc = (lane & 7) ^ lr;    // staging: row inside an 8-row piece, SOURCE chunk (swizzle on the source)
    const int lrow = lane & 15, kq = lane >> 4;        // fragment row (weights) / column (pixels), k quarter
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int nchunk = p.nchunk;
    const int G = 3 * nchunk;                          // (kernel row, chunk) groups per tile
    const i32x4 w_srd = make_srd(p.w, p.w_bytes);

    // ---- weight staging: piece P = 8 d + wave holds stage rows 8 P + lr = tap kw, channel n (row kw * 16 NF + n)
    unsigned wvoff[Cfg::BPW];
#pragma unroll
    for (int d = 0; d < Cfg::BPW; ++d) {
        const int row = (d * 8 + wave) * 8 + lr;
        const int kw = row / (16 * NF), n = row - kw * (16 * NF);
        wvoff[d] = row < Cfg::BR ? (unsigned)n * (unsigned)p.Kbytes + (unsigned)(kw * nchunk * 128) + (unsigned)sc * 16u : HN_OOB;
    }

    // ---- staging cursor: the (tile, kernel row, chunk) group whose stage goes out next, two groups ahead of the multiplies
    unsigned hbase[4];
    int hiy[4];
    i32x4 in_srd = w_srd;
    int st_Hin = 1, st_row_b = 0;
    auto stage_tile = [&](int T) {                     // T uniform; T >= ntiles: zeros
        if (T >= p.ntiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { hiy[i] = -(1 << 28); hbase[i] = 0; }
            return;
        }
        int gi = 0;
#pragma unroll
        for (int i = 1; i < RTN_MAX_GROUPS; ++i)
            if (i < p.ngroups && T >= p.g[i].tile_begin) gi = i;
        const HNGroup& Gs = p.g[gi];
        const int m0 = (T - Gs.tile_begin) * HN_TM;
        in_srd = make_srd(Gs.in, Gs.in_bytes);
        st_Hin = Gs.Hin;
        st_row_b = Gs.in_row_stride_b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = (i * 8 + wave) * 8 + lr;
            const int f = m0 + h - p.pad_l;
            if (h < HN_R - 1 && f >= 0 && f < Gs.M) {
                int b, rem, y, x;
                divmod24(f, Gs.cells, Gs.inv_cells, b, rem);
                divmod24(rem, Gs.Win, Gs.inv_w, y, x);
                hiy[i] = y;
                hbase[i] = (unsigned)f * (unsigned)p.pix_b + (unsigned)sc * 16u;
            } else {
                hiy[i] = -(1 << 28);
                hbase[i] = 0;
            }
        }
    };
    // workgroup b runs on XCD b % 8: tile_of hands every XCD a contiguous range of tiles, so that the halo rows neighbouring tiles
    // share (kernel rows 0 / 2 reach one image row up / down) are fetched into one L2 only (see rtn_conv_halo8.hip)
    auto tile_of = [&](int v) {
        if (v >= p.ntiles || !p.xcd) return v;
        const int x = v & 7, j = v >> 3, base = p.ntiles >> 3, rem = p.ntiles & 7;
        return x * base + (x < rem ? x : rem) + j;
    };
    int st_v = blockIdx.x, st_kh = 0, st_cc = 0;
    unsigned st_ring = 0;                              // LDS offset of the stage being filled
    // one of the Cfg::PP (<= 7) DMA instructions of the stage being filled (pieces 0-3: halo, 4..: weights); advance_stage() after the
    // last.  A macro with a literal piece number: the descriptor operand of the asm must stay in SGPRs.
#define HN_ISSUE(I)                                                                                            \
    if ((I) < 4) {                                                                                             \
        const int dy_ = st_kh - p.pad_t;                                                                       \
        const unsigned delta_ = (unsigned)(dy_ * st_row_b + st_cc * 128);                                      \
        const bool ok_ = (unsigned)(hiy[(I) & 3] + dy_) < (unsigned)st_Hin;                                    \
        dma16(in_srd, ok_ ? hbase[(I) & 3] + delta_ : HN_OOB, 0u, lds_base + st_ring + (unsigned)(wave * 1024 + ((I) & 3) * 8192)); \
    } else if ((I) < Cfg::PP) {                                                                                \
        constexpr int d_ = (I) >= 4 && (I) - 4 < Cfg::BPW ? (I) - 4 : 0;                                       \
        const unsigned kcol_ = (unsigned)((st_kh * 3 * nchunk + st_cc) * 128);                                 \
        const int P_ = d_ * 8 + wave;                                                                          \
        dma16(w_srd, wvoff[d_], kcol_, lds_base + (P_ < Cfg::BP ? st_ring + HN_HALO + (unsigned)P_ * 1024u : Cfg::DUMP)); \
    }
    auto advance_stage = [&]() {
        st_ring = st_ring == 2 * Cfg::STAGE ? 0u : st_ring + Cfg::STAGE;
        // kernel rows of a channel chunk back to back: the image rows a tile re-reads for kh = 0, 1, 2 are then one stage apart and
        // still in the XCD's L2
        if (p.kh_fast) {
            if (++st_kh == 3) {
                st_kh = 0;
                if (++st_cc == nchunk) { st_cc = 0; st_v += (int)gridDim.x; stage_tile(tile_of(st_v)); }
            }
        } else if (++st_cc == nchunk) {
            st_cc = 0;
            if (++st_kh == 3) {
                st_kh = 0;
                st_v += (int)gridDim.x;
                stage_tile(tile_of(st_v));
            }
        }
    };
    auto issue_stage = [&]() {
        HN_ISSUE(0) HN_ISSUE(1) HN_ISSUE(2) HN_ISSUE(3) HN_ISSUE(4) HN_ISSUE(5) HN_ISSUE(6)
        advance_stage();
    };

    // bias of this lane's channels 16 f + 4 kq + r
    f32x4 bias4[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        bias4[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4[f] = *reinterpret_cast<const f32x4*>(p.bias + 16 * f + 4 * kq);
    }
    // weight fragment address inside a stage: row kw * 16 NF + 16 f + lrow, chunk (4 ks + kq) ^ (lrow & 7)
    const unsigned w_lane = HN_HALO + (unsigned)(lrow * 128 + ((kq ^ (lrow & 7)) << 4));

    stage_tile(tile_of(st_v));
    issue_stage();
    issue_stage();

    unsigned c_ring = 0;                               // stage the multiplies read
    int cv = blockIdx.x;
    bool first = true;
    while (cv < p.ntiles) {
        const int tile = tile_of(cv);
        int gi = 0;
#pragma unroll
        for (int i = 1; i < RTN_MAX_GROUPS; ++i)
            if (i < p.ngroups && tile >= p.g[i].tile_begin) gi = i;
        const HNGroup& Gc = p.g[gi];
        const int m0 = (tile - Gc.tile_begin) * HN_TM;
        // pixel fragment read offsets per tap (k half 0; half 1 = ^ 64), the zero row where the tap leaves the image
        unsigned arow[3][2];
        unsigned obase[2];                             // element offset of this lane's pixel in `out`, HN_OOB when it has none
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rloc = wave * 32 + i * 16 + lrow;
            const int m = m0 + rloc;
            const int mc = m < Gc.M ? m : Gc.M - 1;
            int b, rem, y, x;
            divmod24(mc, Gc.cells, Gc.inv_cells, b, rem);
            divmod24(rem, Gc.Win, Gc.inv_w, y, x);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int rr = rloc + kw;
                const bool off = rloc >= HN_TM || (unsigned)(x + kw - p.pad_l) >= (unsigned)Gc.Win;
                arow[kw][i] = off ? HN_ZERO_ROW : (unsigned)(rr * 128 + ((kq ^ (rr & 7)) << 4));
            }
            obase[i] = (rloc < HN_TM && m < Gc.M) ? (unsigned)b * Gc.out_img_stride + Gc.out_off + (unsigned)rem * (unsigned)p.out_ld : HN_OOB;
        }
        f32x4 acc[2][NF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[i][f] = bias4[f];

#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            // this group's stage must have landed; in flight behind it: the next group's stage and, on the first group of a later
            // tile, the stores of the tile before
            if (g == 0 && !first) {
                if (p.vec) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::PP + 2 * NF) : "memory");
                else       asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::PP + 8 * NF) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::PP) : "memory");
            }
            __builtin_amdgcn_s_barrier();              // every wave's pieces have landed and every wave has left the group before
            // The stage two groups ahead goes out (into the stage the group before read) BETWEEN this group's six (tap, k half)
            // steps, one or two DMA instructions behind each step's fragment reads: issued back to back they fill the vector-memory
            // queue and every wave sat ~750 clocks in issue before its first MFMA, while the queue ran dry under the MFMAs
            // (profiles/r2_v3_halon_stage_timeline.txt).  The asm DMA is a memory fence to the compiler, so the next step's
            // fragments are read ahead of it by hand.
            const char* sb = lds + c_ring;
            uint4 fx[2][2], fw[2][NF];
#define HN_READ(STEP, SET)                                                                                     \
            {                                                                                                  \
                const int kw_ = ((STEP) >> 1) % 3;                                                             \
                const unsigned ks_ = ((STEP) & 1) * 64u;                                                       \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) fx[SET][i] = *reinterpret_cast<const uint4*>(sb + (arow[kw_][i] ^ ks_)); \
                _Pragma("unroll") for (int f = 0; f < NF; ++f)                                                 \
                    fw[SET][f] = *reinterpret_cast<const uint4*>(sb + ((w_lane ^ ks_) + (unsigned)((kw_ * NF + f) * 2048))); \
            }
#define HN_STEP(STEP)                                                                                          \
            {                                                                                                  \
                if ((STEP) + 1 < 6) HN_READ((STEP) + 1, ((STEP) + 1) & 1)                                      \
                if (0 * 6 / Cfg::PP == (STEP)) { HN_ISSUE(0) }                                                 \
                if (1 * 6 / Cfg::PP == (STEP)) { HN_ISSUE(1) }                                                 \
                if (2 * 6 / Cfg::PP == (STEP)) { HN_ISSUE(2) }                                                 \
                if (3 * 6 / Cfg::PP == (STEP)) { HN_ISSUE(3) }                                                 \
                if (4 * 6 / Cfg::PP == (STEP)) { HN_ISSUE(4) }                                                 \
                if (5 < Cfg::PP && 5 * 6 / Cfg::PP == (STEP)) { HN_ISSUE(5) }                                  \
                if (6 < Cfg::PP && 6 * 6 / Cfg::PP == (STEP)) { HN_ISSUE(6) }                                  \
                _Pragma("unroll") for (int f = 0; f < NF; ++f)                                                 \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                              \
                        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[(STEP) & 1][f]),      \
                                                                           __builtin_bit_cast(bf16x8, fx[(STEP) & 1][i]), acc[i][f], 0, 0, 0); \
            }
            HN_READ(0, 0)
            HN_STEP(0) HN_STEP(1) HN_STEP(2) HN_STEP(3) HN_STEP(4) HN_STEP(5)
#undef HN_STEP
#undef HN_READ
            advance_stage();
            c_ring = c_ring == 2 * Cfg::STAGE ? 0u : c_ring + Cfg::STAGE;
        }
        // ---- epilogue: channels 16 f + 4 kq .. + 3 of pixel (i, lrow); the same number of store instructions for every wave
        {
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)Gc.out, 0, (int)__builtin_amdgcn_readfirstlane((int)Gc.out_bytes), 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    f32x4 v = acc[i][f];
                    if (p.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                    }
                    if (p.sigmoid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + expf(-v[r]));
                    }
                    const int ch0 = 16 * f + 4 * kq;
                    const int cnt = p.N - ch0;
                    if (p.vec) {
                        const unsigned off = (obase[i] != HN_OOB && cnt >= 4) ? (obase[i] + (unsigned)ch0) * 4u : HN_OOB;
                        u32x4 o = __builtin_bit_cast(u32x4, v);
                        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, (int)off, 0, 0);
                        RTN_STORE_GUARD(o)
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned off = (obase[i] != HN_OOB && r < cnt) ? (obase[i] + (unsigned)(ch0 + r)) * 4u : HN_OOB;
                            const unsigned o = __float_as_uint(v[r]);
                            __builtin_amdgcn_raw_buffer_store_b32(o, out_rsrc, (int)off, 0, 0);
                            RTN_STORE_GUARD1(o)
                        }
                    }
                }
        }
        cv += (int)gridDim.x;
        first = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS
#undef HN_ISSUE
}

}  // namespace

// Launcher: RTN_OK after a launch, 1 when the layer is not one this kernel takes, < 0 on error.
int rtn_conv_halon_try(rtn_handle_t h, const rtn_conv_desc_t* d, int grid_limit, bool forced) {
    if (d->dtype != RTN_BF16) return 1;
    if (d->KW != 3 || d->KH != 3 || d->sy != 1 || d->sx != 1) return 1;
    if (!(d->flags & RTN_CONV_OUT_F32) || (d->flags & ~(RTN_CONV_OUT_F32 | RTN_CONV_RELU | RTN_CONV_SIGMOID))) return 1;
    if (d->N < 1 || d->N > 48 || d->out_ld < d->N) return 1;
    const int nf = (d->N + 15) / 16;
    if (d->w_rows < 16 * nf) return 1;
    if (d->Crun != d->pix_stride || (d->Crun * 2) % 128 || d->Crun <= 0) return 1;
    if (d->pad_l < 0 || d->pad_l >= 3 || d->pad_t < 0 || d->pad_t >= 3) return 1;
    if (((uintptr_t)d->w & 15) || ((uintptr_t)d->bias & 15)) return 1;
    const int nchunk = d->Crun * 2 / 128;
    const long long Kbytes = 9ll * d->Crun * 2;
    if (Kbytes * d->w_rows >= 0xFFFFFF00ll) return 1;
    HNParams p;
    memset(&p, 0, sizeof(p));
    const int cus = h->num_cus > 0 ? h->num_cus : 256;
    long long tiles = 0;
    bool vec = d->N % 4 == 0 && d->out_ld % 4 == 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (s.Hin != s.Hout || s.Win != s.Wout || s.in_row_stride != (long long)s.Win * d->pix_stride ||
            s.in_img_stride != (long long)s.Hin * s.in_row_stride) return 1;
        const long long cells = (long long)s.Hout * s.Wout, M = cells * d->batch;
        if (s.out_step > 1 || s.out_off < 0 || M >= (1ll << 24) || M < 1) return 1;
        if (!s.in || !s.out || ((uintptr_t)s.in & 15) || ((uintptr_t)s.out & 3)) return 1;
        if (s.in_elems < M * d->Crun || s.in_elems * 2 >= 0xFFFFFF00ll || s.out_elems * 4 >= 0xFFFFFF00ll) return 1;
        if (s.out_img_stride < 0 || (d->batch - 1) * s.out_img_stride + s.out_off + (cells - 1) * d->out_ld + d->N > s.out_elems) return 1;
        if (((uintptr_t)s.out & 15) || s.out_img_stride % 4 || s.out_off % 4) vec = false;
        HNGroup& g = p.g[i];
        g.in = (const char*)s.in;
        g.out = (char*)s.out;
        g.in_bytes = (unsigned)(s.in_elems * 2);
        g.out_bytes = (unsigned)(s.out_elems * 4);
        g.Hin = s.Hin; g.Win = s.Win; g.M = (int)M;
        g.cells = (int)cells;
        g.tile_begin = (int)tiles;
        g.in_row_stride_b = (int)(s.in_row_stride * 2);
        g.out_img_stride = (unsigned)s.out_img_stride;
        g.out_off = (unsigned)s.out_off;
        g.inv_cells = 1.0f / (float)cells;
        g.inv_w = 1.0f / (float)s.Win;
        tiles += (M + HN_TM - 1) / HN_TM;
    }
    if (tiles < 1 || tiles > 0x3fffffff) return 1;
    if (!forced && tiles * 4 < cus) return 1;
    p.w = (const char*)d->w;
    p.bias = d->bias;
    p.w_bytes = (unsigned)(Kbytes * d->w_rows);
    p.ngroups = d->ngroups;
    p.ntiles = (int)tiles;
    p.N = d->N;
    p.Kbytes = (int)Kbytes;
    p.nchunk = nchunk;
    p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.relu = (d->flags & RTN_CONV_RELU) ? 1 : 0;
    p.sigmoid = (d->flags & RTN_CONV_SIGMOID) ? 1 : 0;
    p.out_ld = d->out_ld;
    p.pix_b = d->pix_stride * 2;
    p.vec = vec ? 1 : 0;
    p.xcd = 1;
    p.kh_fast = 1;
    int grid = cus;
    if (grid_limit > 0 && grid_limit < grid) grid = grid_limit;
    if (grid > p.ntiles) grid = p.ntiles;
#define RTN_HN_LAUNCH(NF_)                                                                               \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_halon_kernel<NF_>,                          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, HNCfg<NF_>::LDS)); \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_halon_kernel<NF_>), dim3((unsigned)grid), dim3(HN_THREADS), HNCfg<NF_>::LDS, h->stream, p); \
    } while (0)
    if (nf == 1) RTN_HN_LAUNCH(1); else if (nf == 2) RTN_HN_LAUNCH(2); else RTN_HN_LAUNCH(3);
#undef RTN_HN_LAUNCH
    RTN_CHECK_LAUNCH(h, "conv_halon_kernel");
    return RTN_OK;
}
