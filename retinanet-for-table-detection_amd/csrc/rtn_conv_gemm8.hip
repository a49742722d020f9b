// rtn_conv_gemm8.hip — the 1x1 convolutions with >= 256 output channels as a PERSISTENT implicit GEMM on the staggered 8-phase
// schedule of rtn_conv_halo8.hip (cdna_hip_programming.md §5): fifth kernel generation.
//
// Replaces conv_igemm2_kernel (rtn_conv.hip) on the keras_resnet bottleneck 1x1 layers behind model/defineModel.py:376-380 and the
// FPN laterals (model/defineModel.py:183-195) that have a bias / ReLU epilogue:
//   res{3,4,5}*_branch2a (also the stride-2 'valid' form of a stage's first block), the first blocks' branch2c with the projection
//   shortcut appended along K (rtn_conv1x1_dual_fwd), C5_reduced.
// Those layers are short GEMMs (M = 8,400 .. 133,600 rows at batch 8, K = 128 .. 2048): with one workgroup per 256 x 64 tile the
// per-workgroup prologue (first LDS-DMA round trip), the epilogue and the launch ramp cost more than the K loop.  Here a workgroup
// walks its tiles as ONE stream of K steps, exactly like generation 4:
//   * tile = 64 MI rows (192 or 128) x 256 columns, 8 waves as 4 (M) x 2 (N), wave tile 16 MI x 128, K step 64 = 4 phases of
//     {fragment reads + LDS-DMA issue | s_barrier | 4 MI MFMAs | s_barrier}; waves 4-7 run one barrier behind waves 0-3;
//   * A (activation) tiles in a ring of 3, B (weight) tiles in a ring of 2.  vmcnt retires in order, so the operand that needs the
//     long prefetch is issued LAST in every step: B(s+1) goes out in phases 2-3 (weights are L2-resident: two phases cover them),
//     A(s+2) in phases 3-4 (HBM: it gets two whole steps), and the step's one counted wait, vmcnt(MI), leaves exactly the A pieces
//     in flight.  A buffer is restaged >= 2 phases after its last read and read one phase after the wait that retires it;
//   * the K loop walks two sources back to back (K-concatenated projection shortcut) and either source may be sampled with a
//     stride (the stride-2 1x1 'valid' convolutions): both are per-row byte offsets computed once per tile;
//   * weight rows permuted at staging (position 16 j + c <-> channel 8 c + j): bias-initialised accumulators, ReLU, bf16 and
//     16-byte stores straight from registers, as in generation 4; the bias vector sits in the unused tail of the first A stage.
// LDS: A ring 3 x 32 KiB (24 KiB used, bias in the tail of stage 0), B ring 2 x 32 KiB = 160 KiB.  One workgroup per CU.
//
// STREAM-K (SK, round 4).  The small-M layers of stages 4 / 5 and the FPN top (M = 8,400 .. 33,600 rows) have 44 .. 175 row tiles for
// 256 CUs and 16 .. 32 sequential K steps per tile: most of the chip idles while a third of it walks long K loops.  With SK the
// launch's ntiles x nk K steps are ONE sequence (column block, row tile, k) cut into gridDim.x equal contiguous ranges; a workgroup
// walks its range as the same uninterrupted stream of steps, crossing tile boundaries with the cursors it already has.  A range
// that starts inside a tile first finishes that tile's tail: f32 partial sums in REGISTER ORDER ([slot][thread] x 16 B: every store
// instruction writes one contiguous KiB) go to the workgroup's slab in the caller's workspace with write-through (sc1) stores, every
// wave drains (vmcnt 0), the workgroup meets at barriers, one lane sets the workgroup's flag (relaxed agent-scope store).  A range
// that ends inside a tile holds that tile's HEAD (k = 0 ..): it is the tile's owner - bias-initialised accumulators, polls the flags
// of the workgroups that hold the rest of the tile (relaxed agent-scope loads by one wave, then a barrier), adds their slabs in K
// order with sc1 loads (every load of handed-off bytes: cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Hand-offs measured
// with sc1 loads") and runs the ordinary epilogue.  The order of the adds is fixed by (grid, ntiles, nk): same bits every run.
// Who waits for whom: logical workgroup L = gridDim.x - 1 - blockIdx.x owns range L, the owner L of a tile waits for L + 1, L + 2, ...,
// i.e. for LOWER block ids, which the dispatcher starts first, and only at the very END of its own range, while the tails it waits for
// are the FIRST thing their workgroups do: no wait can depend on a wait.  The poll is bounded (2 s of the 100 MHz clock); a timeout
// counts up the error word of the sync block and goes on (wrong values, never a hang).  Flags are reset by the owner after the poll:
// the sync block (first 4 KiB of the workspace) is zero before and after every launch.
#include "rtn_internal.h"

namespace {

// -DRTN_G8_STAMP: per-workgroup time stamps (100 MHz s_memrealtime) at the stations of the kernel, read back by rtn_debug_g8_stamps
// (tools/g8_stamps.py).  Not in production builds.
#ifdef RTN_G8_STAMP
__device__ unsigned long long g_g8_stamps[1024][8];
// slot 7: shader clocks (s_memtime) between stations 0 and 6, the whole workgroup -> its average frequency
#define G8_STAMP(I_) if (t == 0) { g_g8_stamps[blockIdx.x][I_] = __builtin_amdgcn_s_memrealtime();                                        \
        if ((I_) == 0 || (I_) == 6) { unsigned long long c_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_) :: "memory");  \
            g_g8_stamps[blockIdx.x][7] = (I_) == 0 ? c_ : c_ - g_g8_stamps[blockIdx.x][7]; } }
#else
#define G8_STAMP(I_)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr unsigned G8_OOB = 0xFFFF0000u;              // beyond every descriptor even with the largest uniform offset (K bytes) added
constexpr int G8_THREADS = 512;
constexpr int G8_LDS = 160 * 1024;
constexpr unsigned G8_STAGE = 32768, G8_B_BASE = 3 * 32768, G8_BIAS_OFF = 24576;
constexpr unsigned G8_SYNC_OFF = G8_STAGE + 24576;    // SK: two words in the unused tail of A stage 1 (tiles have at most 192 rows = 24 KiB per stage)

struct G8Src {
    const char* ptr;
    unsigned bytes;
    long long img_stride_b;       // bytes between images
    int row_stride_b, pix_b;      // bytes between rows / pixels
    int step;                     // output pixel (oy, ox) reads input pixel (oy * step, ox * step)
};

struct G8Params {
    G8Src s1, s2;                 // s2.ptr == nullptr: single source
    const char* w;
    const float* bias;
    char* out;
    const char* res;              // RTN_CONV_RES_SAME: dense [M][res_ld], may be `out` itself; null without
    const char* mask;             // RTN_CONV_RELU_MASK source: dense [M][mask_ld]; null without
    unsigned w_bytes, out_bytes, res_bytes, mask_bytes;
    int M, N, Kbytes, nk, nk1;    // K steps in all / from the first source
    int ntiles_m, ntiles_n, ntiles;
    int Hout, Wout;
    float inv_cells, inv_w;
    int relu, out_ld, res_ld, mask_ld, mask_pre;
    // out_step > 1 (data gradient of a stride-2 1x1 'valid' convolution): row m = (b, oy, ox) of the GEMM is pixel
    // b * out_img_pix + oy * out_step * out_pix_w + ox * out_step of `out`, `res` and `mask` (all three live on the forward INPUT grid)
    int out_step, out_pix_w;
    unsigned out_img_pix;
    // RTN_CONV_RES_UPSAMPLE (EPI bit 0 with res_up): the residual is the coarser pyramid level, read at
    // (min(floor(oy * rs_h), Hres - 1), min(floor(ox * rs_w), Wres - 1)) — UpsampleLike + Add of the FPN laterals (model/layers.py:89-98)
    int res_up, Hres, Wres;
    unsigned res_img_stride;      // elements between images of `res`
    float rs_h, rs_w;
    // stream-K (SK instances): sk_total = ntiles * nk K steps shared evenly by the grid; slabs of one partial tile per workgroup in
    // register order; flags[L] = 1 once workgroup L's slab is published; flags[SK_ERR_WORD] counts poll timeouts
    int sk_total;
    char* sk_slab;
    unsigned sk_slab_bytes;
    unsigned* sk_flags;
};
constexpr int SK_ERR_WORD = 1023;

__device__ __forceinline__ i32x4 make_srd(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

__device__ __forceinline__ void dma16(const i32x4& srd, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd), "s"(soff)
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

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// EPI: bit 0 = residual add (the `Add` closing a bottleneck block; accumulated gradient contributions in training), bit 1 = ReLU mask
// of the tensor being differentiated (rtn_conv2d_dgrad): 16 bytes per lane and row, loaded one row fragment ahead of their use.
// NW = 8: 256-column tile (wave tile 16 MI x 128), B ring of two 32 KiB stages, four phases per K step as described above.
// NW = 4: 128-column tile (wave tile 16 MI x 64) for the N = 128 layers (res3 branch2a and the matching data gradients), which on the
//   256-column tile spent half of their B staging, fragment reads and MFMAs on zero columns.  TWO phases per K step ({A and B fragments
//   of one k half | barrier | 4 MI MFMAs | barrier}), B ring of THREE 16 KiB stages: B(s+2) and A(s+2) are both issued in the second
//   phase of step s (B first: it is L2-resident), into the slots step s-1 read last - one full phase after the lagging wave group's
//   last read of them - and the step's one wait, vmcnt(2 + MI), retires everything older, i.e. A(s+1) and B(s+1), one phase before
//   their first read.  A lane ends up with 4 consecutive channels of a pixel (position 16 j + c <-> channel 4 c + j): 8-byte stores.
template <int MI, bool STAGGER, bool DUAL, int EPI, int NW = 8, bool SK = false>
__global__ __launch_bounds__(G8_THREADS, 2) void conv_gemm8_kernel(const G8Params p) {
    static_assert(!SK || (NW == 8 && !(EPI & 2)), "stream-K: the 256-column tile, bias / ReLU / residual epilogues");
    constexpr int R = 64 * MI;                         // rows of a tile
    constexpr int CT = 32 * NW;                        // columns of a tile
    constexpr unsigned BSTG = NW == 8 ? G8_STAGE : G8_STAGE / 2;      // bytes of a B stage
    constexpr int NBD = NW / 2;                        // 1-KiB staging pieces of a B stage per wave
    static_assert(NW == 8 || (NW == 4 && !DUAL), "the 128-column instance: single source");
    static_assert(MI >= 2 && MI <= 3, "the bias table lives in the tail of A stage 0: tiles of at most 192 rows");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane >> 3, sc = (lane & 7) ^ lr;
    const int lrow = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    G8_STAMP(0)
    const int nk = p.nk;
    const i32x4 w_srd = make_srd(p.w, p.w_bytes);
    const i32x4 s1_srd = make_srd(p.s1.ptr, p.s1.bytes);
    const i32x4 s2_srd = DUAL ? make_srd(p.s2.ptr, p.s2.bytes) : s1_srd;

    // bias -> LDS (unused tail of A stage 0), once
    {
        float* bl = reinterpret_cast<float*>(lds + G8_BIAS_OFF);
        for (int i = t; i < p.N; i += G8_THREADS) bl[i] = p.bias ? p.bias[i] : 0.f;
    }
    // SK: [0] counts the waves that have drained their slab stores (the last one raises the global flag), [1] is set by wave 0 once the
    // flags this workgroup waits for are up.  Each is used at most once per launch; published by the prologue's barrier.
    unsigned* const sk_sync = reinterpret_cast<unsigned*>(lds + G8_SYNC_OFF);
    if (SK && t == 0) { sk_sync[0] = 0u; sk_sync[1] = 0u; }

    // workgroup -> first tile: tiles that share an A row block (same m tile) are neighbours; workgroups b and b + 8 share an XCD
    int tile;
    {
        const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tstride = (int)gridDim.x;
    // SK: logical workgroup Lw owns K steps [u0, u1) of the sequence (tile, k), tile = nt * ntiles_m + mt (column blocks outermost:
    // the workgroups that walk the same rows of A for different column blocks are gridDim.x / ntiles_n block ids apart - a multiple
    // of 8 for the grids the host picks, i.e. on one XCD at the same time)
    const int Lw = SK ? (int)gridDim.x - 1 - (int)blockIdx.x : 0;
    const int u0 = SK ? (int)((long long)Lw * p.sk_total / (int)gridDim.x) : 0;
    const int u1 = SK ? (int)((long long)(Lw + 1) * p.sk_total / (int)gridDim.x) : 0;
    auto tile_mn = [&](int T, int& mt, int& nt) {
        if (SK) { nt = T / p.ntiles_m; mt = T - nt * p.ntiles_m; }
        else    { mt = T / p.ntiles_n; nt = T - mt * p.ntiles_n; }
    };

    // ---- staging cursors.  A: per-row byte offsets of this lane's MI rows in both sources (out of range past M);
    //      B: per-piece byte offsets of the weight rows of the tile's 256 columns (permuted, see the header)
    unsigned hoff1[MI], hoff2[DUAL ? MI : 1];
    auto a_tile = [&](int T) {
        int mt, nt_;
        tile_mn(T, mt, nt_);
        const int m0 = mt * R;
        const int cells = p.Hout * p.Wout;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + (i * 8 + wave) * 8 + lr;
            if (T < p.ntiles && m < p.M) {
                int b, rem, oy, ox;
                divmod24(m, cells, p.inv_cells, b, rem);
                divmod24(rem, p.Wout, p.inv_w, oy, ox);
                hoff1[i] = (unsigned)((long long)b * p.s1.img_stride_b + (long long)(oy * p.s1.step) * p.s1.row_stride_b +
                                      (long long)(ox * p.s1.step) * p.s1.pix_b) + (unsigned)sc * 16u;
                if (DUAL)
                    hoff2[i] = (unsigned)((long long)b * p.s2.img_stride_b + (long long)(oy * p.s2.step) * p.s2.row_stride_b +
                                          (long long)(ox * p.s2.step) * p.s2.pix_b) + (unsigned)sc * 16u;
            } else {
                hoff1[i] = G8_OOB;
                if (DUAL) hoff2[i] = G8_OOB;
            }
        }
    };
    unsigned wrow_off[NBD];
    auto b_tile = [&](int T) {
        int mt_ = 0, nt = 0;
        if (T < p.ntiles) tile_mn(T, mt_, nt);
#pragma unroll
        for (int d = 0; d < NBD; ++d) {
            const int P = d * 64 + wave * 8 + lr;
            const int nrow = NW == 8 ? nt * 256 + (P >> 7) * 128 + 8 * (P & 15) + ((P >> 4) & 7)
                                     : nt * 128 + (P >> 6) * 64 + 4 * (P & 15) + ((P >> 4) & 3);
            wrow_off[d] = (unsigned)nrow * (unsigned)p.Kbytes + (unsigned)sc * 16u;
        }
    };
    auto stage_a = [&](int i, int k, unsigned slot_addr) {          // piece i of the A tile of K step k of the A cursor's tile
        if (DUAL && k >= p.nk1) dma16(s2_srd, hoff2[DUAL ? i : 0], (unsigned)(k - p.nk1) * 128u, lds_base + slot_addr + (unsigned)(wave * 1024 + i * 8192));
        else                    dma16(s1_srd, hoff1[i], (unsigned)k * 128u, lds_base + slot_addr + (unsigned)(wave * 1024 + i * 8192));
    };
    auto stage_b = [&](int d, int k, unsigned slot_addr) {
        dma16(w_srd, wrow_off[d], (unsigned)k * 128u, lds_base + slot_addr + (unsigned)(wave * 1024 + d * 8192));
    };

    // fragment read offsets (fixed for the whole kernel): A row 16 MI wm + 16 i + lrow, B position 128 wn + 16 j + lrow
    unsigned arow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * (16 * MI) + i * 16 + lrow;
        arow[i] = (unsigned)(row * 128 + ((kq ^ (row & 7)) << 4));
    }
    const unsigned b_lane = G8_B_BASE + (unsigned)((wn * (16 * NW) + lrow) * 128 + ((kq ^ (lrow & 7)) << 4));

    // ---- prologue: A tiles of steps 0 and 1, B tile of step 0
    int ta = tile, ka = 0;          // A cursor: the step whose A tile is staged next (two ahead of the multiply)
    int tb = tile, kb = 0;          // B cursor (one ahead)
    int ua = u0;                    // SK: the A cursor's position in the step sequence; past u1 it stages nothing (out-of-range rows)
    if (SK) { ta = u0 / nk; ka = u0 - ta * nk; tb = ta; kb = ka; }
    a_tile(SK && u0 >= u1 ? p.ntiles : ta);
    b_tile(tb);
    unsigned a_st = 0, b_st = 0;    // ring slots the cursors stage into next (byte offsets)
#define G8_ADV_A()                                                                                   \
    {                                                                                                \
        a_st = a_st == 2 * G8_STAGE ? 0u : a_st + G8_STAGE;                                          \
        if (SK) {                                                                                    \
            ++ua;                                                                                    \
            if (ua >= u1) { if (ua == u1) a_tile(p.ntiles); }                                        \
            else if (++ka == nk) { ka = 0; ++ta; a_tile(ta); }                                       \
        } else if (++ka == nk) { ka = 0; ta += tstride; a_tile(ta); }                                \
    }
#define G8_ADV_B()                                                                                   \
    {                                                                                                \
        if (NW == 8) b_st ^= G8_STAGE; else b_st = b_st == 2 * BSTG ? 0u : b_st + BSTG;              \
        if (++kb == nk) { kb = 0; tb += SK ? 1 : tstride; b_tile(tb); }                              \
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < MI; ++i) stage_a(i, ka, a_st);
        G8_ADV_A()
    }
#pragma unroll
    for (int s = 0; s < (NW == 8 ? 1 : 2); ++s) {      // the 128-column instance runs its B stream two steps ahead, like A
#pragma unroll
        for (int d = 0; d < NBD; ++d) stage_b(d, kb, G8_B_BASE + b_st);
        G8_ADV_B()
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // the LDS-DMA pieces and this wave's bias stores
    if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                      // also publishes the bias table
    G8_STAMP(1)

    unsigned a_cur = 0, b_cur = 0;                     // ring slots of the step being multiplied
    f32x4 acc[MI][NW];
    const float* bias_l = reinterpret_cast<const float*>(lds + G8_BIAS_OFF);

#define G8_LDA(KS)                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                                \
        fa[i_] = *reinterpret_cast<const uint4*>(lds + a_cur + (arow[i_] ^ ((KS) * 64u)));
#define G8_LDB(KS, HALF)                                                                             \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
        fb[j_] = *reinterpret_cast<const uint4*>(lds + b_cur + (b_lane ^ ((KS) * 64u)) + ((HALF) * 4 + j_) * 2048);
#define G8_MFMA(HALF)                                                                                \
    __builtin_amdgcn_s_barrier();                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_)                                            \
            acc[i_][(HALF) * 4 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
                __builtin_bit_cast(bf16x8, fa[i_]), __builtin_bit_cast(bf16x8, fb[j_]), acc[i_][(HALF) * 4 + j_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_barrier();

    int u = u0;                                        // SK: the next step of this workgroup's range
    while (SK ? u < u1 : tile < p.ntiles) {
        int k0 = 0, k1 = nk;                           // the K steps of this tile that this workgroup multiplies
        if (SK) {
            tile = u / nk;
            k0 = u - tile * nk;
            k1 = k0 + (u1 - u) < nk ? k0 + (u1 - u) : nk;
        }
        int mt, nt;
        tile_mn(tile, mt, nt);
        const int m0 = mt * R, n0 = nt * CT;
        {
            const float* bp = bias_l + n0 + wn * (16 * NW) + NW * lrow;
            f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + (NW == 8 ? 4 : 0));
            if (SK && k0 != 0) { b0 = (f32x4){0.f, 0.f, 0.f, 0.f}; b1 = b0; }     // not the tile's head: a plain partial sum
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = (f32x4){b0[j], b0[j], b0[j], b0[j]};
                    if (NW == 8) acc[i][(NW == 8 ? 4 : 0) + j] = (f32x4){b1[j], b1[j], b1[j], b1[j]};
                }
            }
        }
#pragma unroll 1
        for (int k = k0; k < k1; ++k) {
            uint4 fa[MI], fb[4];
            if constexpr (NW == 4) {
                // phase 1: fragments of k half 0, no staging (the stages freed by step k-1 may still be read by the lagging wave group)
                G8_LDA(0) G8_LDB(0, 0)
                G8_MFMA(0)
                // phase 2: fragments of k half 1; B(s+2), then A(s+2); everything older has landed after the wait
                G8_LDA(1) G8_LDB(1, 0)
#pragma unroll
                for (int d = 0; d < NBD; ++d) stage_b(d, kb, G8_B_BASE + b_st);
                G8_ADV_B()
#pragma unroll
                for (int i = 0; i < MI; ++i) stage_a(i, ka, a_st);
                G8_ADV_A()
                if (MI == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else         asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                G8_MFMA(0)
                a_cur = a_cur == 2 * G8_STAGE ? 0u : a_cur + G8_STAGE;
                b_cur = b_cur == 2 * BSTG ? 0u : b_cur + BSTG;
            } else {
            // phase 1: no staging (the B stage freed by step k-1 may still be read by the lagging wave group)
            G8_LDA(0) G8_LDB(0, 0)
            G8_MFMA(0)
            // phase 2: first half of B(s+1)
            G8_LDB(0, 1)
            stage_b(0, kb, G8_B_BASE + b_st);
            stage_b(1, kb, G8_B_BASE + b_st);
            G8_MFMA(1)
            // phase 3: second half of B(s+1), first piece of A(s+2)
            G8_LDA(1) G8_LDB(1, 0)
            stage_b(NBD - 2, kb, G8_B_BASE + b_st);
            stage_b(NBD - 1, kb, G8_B_BASE + b_st);
            G8_ADV_B()
            stage_a(0, ka, a_st);
            G8_MFMA(0)
            // phase 4: the rest of A(s+2); everything but those MI pieces must have landed (B(s+1), and A(s+1) from the step before)
            G8_LDB(1, 1)
#pragma unroll
            for (int i = 1; i < MI; ++i) stage_a(i, ka, a_st);
            G8_ADV_A()
            if (MI == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else         asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            G8_MFMA(1)
            a_cur = a_cur == 2 * G8_STAGE ? 0u : a_cur + G8_STAGE;
            b_cur ^= G8_STAGE;
            }
        }
        // The leading wave group's closing barrier (it pairs with the lagging group's last K-loop barrier) comes BEFORE the last
        // epilogue of the workgroup instead of after it: the lagging group's last epilogue then runs beside the leading group's, not
        // behind it (nothing below touches the LDS rings or meets at a barrier again).
        G8_STAMP(2)
        if (STAGGER && grp == 0 && (SK ? u + (k1 - k0) >= u1 : tile + tstride >= p.ntiles)) __builtin_amdgcn_s_barrier();
        G8_STAMP(3)
        bool run_epilogue = true;
        if constexpr (SK) {
            u += k1 - k0;
            constexpr unsigned SLOTS = MI * 4 * 2;                                 // 16-byte slots per thread of a partial tile
            constexpr unsigned SLAB = SLOTS * G8_THREADS * 16u;                    // bytes per workgroup
            const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)p.sk_slab, 0, (int)__builtin_amdgcn_readfirstlane((int)p.sk_slab_bytes), 0x00020000);
            if (k0 != 0) {
                // ---- a tail / middle piece of the tile: publish the partial sums (write-through), then the flag
                const unsigned base = (unsigned)Lw * SLAB + (unsigned)t * 16u;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        u32x4 o0, o1;
                        o0.x = __float_as_uint(acc[i][0][r]); o0.y = __float_as_uint(acc[i][1][r]); o0.z = __float_as_uint(acc[i][2][r]); o0.w = __float_as_uint(acc[i][3][r]);
                        o1.x = __float_as_uint(acc[i][4][r]); o1.y = __float_as_uint(acc[i][5][r]); o1.z = __float_as_uint(acc[i][6][r]); o1.w = __float_as_uint(acc[i][7][r]);
                        const unsigned slot = (unsigned)((i * 4 + r) * 2);
                        __builtin_amdgcn_raw_buffer_store_b128(o0, slab_rsrc, (int)(base + slot * (G8_THREADS * 16u)), 0, 16);            // aux 16 = sc1
                        RTN_STORE_GUARD(o0)
                        __builtin_amdgcn_raw_buffer_store_b128(o1, slab_rsrc, (int)(base + (slot + 1) * (G8_THREADS * 16u)), 0, 16);
                        RTN_STORE_GUARD(o1)
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // EVERY storing wave drains before the flag may go up
                // ... and says so in LDS; the wave whose add comes last raises the flag.  No barrier: the two wave groups keep their offset
                // and the leading group goes straight on to the next tile's steps.
                unsigned arrived = 0;
                if (lane == 0) arrived = __hip_atomic_fetch_add(sk_sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane == 0 && arrived == (unsigned)(G8_THREADS / 64 - 1))
                    __hip_atomic_store(p.sk_flags + Lw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run_epilogue = false;
            } else if (k1 != nk) {
                // ---- the head of a tile whose other pieces belong to workgroups Lw + 1 .. Lw + np (lower block ids): wait, add in K order
                const int tile_end = (tile + 1) * nk, G = (int)gridDim.x;
                int np = 0;
                while (Lw + 1 + np < G && (int)((long long)(Lw + 1 + np) * p.sk_total / G) < tile_end) ++np;
                if (wave == 0) {
                    unsigned* f = p.sk_flags + Lw + 1 + lane;
                    bool pending = lane < np;
                    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                    while (__builtin_amdgcn_ballot_w64(pending) != 0ull) {
                        if (pending && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) pending = false;
                        if (__builtin_amdgcn_s_memrealtime() - t_start > 200000000ull) {       // 2 s at 100 MHz: give up, count it, go on
                            if (pending) __hip_atomic_fetch_add(p.sk_flags + SK_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (lane < np) __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // clean for the next launch
                    if (lane == 0) __hip_atomic_store(sk_sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    while (__hip_atomic_load(sk_sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // (no instruction: keeps the slab loads below the poll / the LDS word)
                __builtin_amdgcn_sched_barrier(0);
                for (int qn = 1; qn <= np; ++qn) {
                    const unsigned base = (unsigned)(Lw + qn) * SLAB + (unsigned)t * 16u;
                    // one row fragment at a time (8 x 16 B per lane in flight): the registers of a second batch would spill
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        u32x4 buf[8];
#pragma unroll
                        for (int s8 = 0; s8 < 8; ++s8)
                            buf[s8] = __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, (int)(base + (unsigned)(i * 8 + s8) * (G8_THREADS * 16u)), 0, 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                acc[i][j][r] += __uint_as_float(buf[r * 2 + (j >> 2)][j & 3]);
                    }
                }
            }
        }
        G8_STAMP(4)
        // ---- epilogue: [mask] [+ residual] [mask] ReLU, bf16, 4 MI stores of 16 B per lane
        if (run_epilogue) {
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)p.out, 0, (int)__builtin_amdgcn_readfirstlane((int)p.out_bytes), 0x00020000);
            const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 1) ? p.res : p.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 1) ? p.res_bytes : 0u)), 0x00020000);
            const __amdgpu_buffer_rsrc_t mask_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 2) ? p.mask : p.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 2) ? p.mask_bytes : 0u)), 0x00020000);
            const int ncol = n0 + wn * (16 * NW) + NW * lrow;
            const bool col_ok = ncol < p.N;
            auto scatter_pix = [&](int m) -> unsigned {        // pixel of GEMM row m in out / res / mask
                if (p.out_step <= 1) return (unsigned)m;
                int b, rem, oy, ox;
                divmod24(m, p.Hout * p.Wout, p.inv_cells, b, rem);
                divmod24(rem, p.Wout, p.inv_w, oy, ox);
                return (unsigned)b * p.out_img_pix + (unsigned)(oy * p.out_step) * (unsigned)p.out_pix_w + (unsigned)(ox * p.out_step);
            };
            // residual / mask rows of every row fragment: all MI x 4 loads of a wave go out before the first is used (these layers are
            // bound by their pixel traffic: more loads in flight per wave).  NW = 4: 8 bytes per lane and row instead of 16.
            constexpr int NWD = NW / 2;                       // dwords per lane and row
            u32x4 rq[MI][4], mq[(EPI & 2) ? MI : 1][4];
            auto fetch = [&](int i) {
                const int par = i, parm = (EPI & 2) ? i : 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * (16 * MI) + i * 16 + kq * 4 + r;
                    const bool ok = col_ok && m < p.M;
                    const unsigned pm = scatter_pix(ok ? m : 0);
                    if (EPI & 1) {
                        unsigned rrow = pm * (unsigned)p.res_ld;
                        if (p.res_up) {
                            int b, rem, oy, ox;
                            divmod24(ok ? m : 0, p.Hout * p.Wout, p.inv_cells, b, rem);
                            divmod24(rem, p.Wout, p.inv_w, oy, ox);
                            int sy_ = (int)floorf((float)oy * p.rs_h), sx_ = (int)floorf((float)ox * p.rs_w);
                            sy_ = sy_ < p.Hres - 1 ? sy_ : p.Hres - 1;
                            sx_ = sx_ < p.Wres - 1 ? sx_ : p.Wres - 1;
                            rrow = (unsigned)b * p.res_img_stride + (unsigned)(sy_ * p.Wres + sx_) * (unsigned)p.res_ld;
                        }
                        const int roff = (int)(ok ? (rrow + (unsigned)ncol) * 2u : G8_OOB);
                        if (NW == 8) rq[par][r] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, roff, 0, 0);
                        else { const u32x2 t2 = __builtin_amdgcn_raw_buffer_load_b64(res_rsrc, roff, 0, 0); rq[par][r] = (u32x4){t2.x, t2.y, 0u, 0u}; }
                    }
                    if (EPI & 2) {
                        const int moff = (int)(ok ? (pm * (unsigned)p.mask_ld + (unsigned)ncol) * 2u : G8_OOB);
                        if (NW == 8) mq[parm][r] = __builtin_amdgcn_raw_buffer_load_b128(mask_rsrc, moff, 0, 0);
                        else { const u32x2 t2 = __builtin_amdgcn_raw_buffer_load_b64(mask_rsrc, moff, 0, 0); mq[parm][r] = (u32x4){t2.x, t2.y, 0u, 0u}; }
                    }
                }
            };
            if (EPI) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fetch(i);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * (16 * MI) + i * 16 + kq * 4 + r;
                    float v[NW];
#pragma unroll
                    for (int j = 0; j < NW; ++j) v[j] = acc[i][j][r];
                    if (EPI) {
                        const u32x4 rw = rq[i][r], mw = mq[(EPI & 2) ? i : 0][r];
#pragma unroll
                        for (int j = 0; j < NWD; ++j) {
                            const unsigned mj = (EPI & 2) ? mw[j] : 0x3f803f80u, rj = (EPI & 1) ? rw[j] : 0u;
                            const bool keep_lo = __uint_as_float(mj << 16) > 0.f, keep_hi = __uint_as_float(mj & 0xffff0000u) > 0.f;
                            if ((EPI & 2) && p.mask_pre) { if (!keep_lo) v[2 * j] = 0.f; if (!keep_hi) v[2 * j + 1] = 0.f; }
                            if (EPI & 1) { v[2 * j] += __uint_as_float(rj << 16); v[2 * j + 1] += __uint_as_float(rj & 0xffff0000u); }
                            if ((EPI & 2) && !p.mask_pre) { if (!keep_lo) v[2 * j] = 0.f; if (!keep_hi) v[2 * j + 1] = 0.f; }
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int j = 0; j < NW; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    }
                    const unsigned off = (col_ok && m < p.M) ? (scatter_pix(m < p.M ? m : 0) * (unsigned)p.out_ld + (unsigned)ncol) * 2u : G8_OOB;
                    if constexpr (NW == 8) {
                        u32x4 o;
                        o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
                        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, (int)off, 0, 0);
                        RTN_STORE_GUARD(o)
                    } else {
                        u32x2 o;
                        o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]);
                        __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, (int)off, 0, 0);
                        asm volatile("s_nop 3" :: "v"(o.x), "v"(o.y));
                    }
                }
            }
        }
        G8_STAMP(5)
        if (!SK) tile += tstride;
    }
#undef G8_MFMA
#undef G8_LDB
#undef G8_LDA
#undef G8_ADV_B
#undef G8_ADV_A
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS
    G8_STAMP(6)
}

}  // namespace

// Launcher: RTN_OK after a launch, 1 when the layer is not one this kernel takes, < 0 on error.
// `sync` / `ws` / `ws_cap`: the caller-owned workspace of the stream-K form - its 4-KiB sync block (zero before and after every launch)
// and the slab area behind it; `query` != nullptr: no launch, *query = slab bytes this layer would use (0: no stream-K for it).
// `sk_mode`: -1 = cost model, 0 = never, 1 = wherever the shape allows (tests, A/B).
int rtn_conv_gemm8_try(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_src2_t* s2, int grid_limit, bool stagger, bool forced,
                       int mi_force, unsigned* sync, void* ws, long long ws_cap, size_t* query, int sk_mode) {
    if (query) *query = 0;
    if (d->dtype != RTN_BF16 || d->ngroups != 1) return 1;
    if (d->KH != 1 || d->KW != 1 || d->pad_t != 0 || d->pad_l != 0 || d->sy != d->sx || d->sy < 1 || d->sy > 2) return 1;
    if (d->flags & ~(RTN_CONV_RELU | RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE | RTN_CONV_RELU_MASK | RTN_CONV_MASK_PRE)) return 1;
    if ((d->flags & RTN_CONV_MASK_PRE) && !(d->flags & RTN_CONV_RELU_MASK)) return 1;
    if ((d->flags & RTN_CONV_RES_SAME) && (d->flags & RTN_CONV_RES_UPSAMPLE)) return 1;
    const bool res_up = d->flags & RTN_CONV_RES_UPSAMPLE;
    const int epi = ((d->flags & (RTN_CONV_RES_SAME | RTN_CONV_RES_UPSAMPLE)) ? 1 : 0) | ((d->flags & RTN_CONV_RELU_MASK) ? 2 : 0);
    if (epi && s2) return 1;
    // N = 128: half of the 256-column tile multiplies zeros (weight rows past w_rows come from the descriptor's range check).  The
    // layers this is for are bound by their pixel traffic: res3 branch2a in the step's sequence 0.063 -> 0.055 ms (RTN_CONV_G8_N128=0: off)
    const bool n128 = d->N == 128 && (forced || rtn_env_int("RTN_CONV_G8_N128", 1) != 0);
    if (!n128 && (d->N < 256 || d->N % 256)) return 1;
    if (d->N > 2048 || d->w_rows != d->N || d->out_ld % 8) return 1;
    if ((d->Crun * 2) % 128 || d->Crun <= 0 || d->pix_stride < d->Crun || (d->pix_stride * 2) % 16) return 1;
    if (((uintptr_t)d->w & 15) || ((uintptr_t)d->bias & 15)) return 1;
    const rtn_conv_group_t& g = d->g[0];
    if (!g.in || !g.out || ((uintptr_t)g.in & 15) || ((uintptr_t)g.out & 15)) return 1;
    const long long cells = (long long)g.Hout * g.Wout, M = cells * d->batch;
    if (M < 1 || M >= (1ll << 24)) return 1;
    const bool scatter = g.out_step > 1;                  // dgrad of a stride-2 1x1 'valid' conv: rows land on the stride grid
    if (scatter && (res_up || g.out_pix_w < (g.Wout - 1) * g.out_step + 1 || g.out_img_stride % d->out_ld || g.out_img_stride <= 0)) return 1;
    const long long img_pix = scatter ? g.out_img_stride / d->out_ld : cells;            // pixels per image of out / res / mask
    const long long last_pix = scatter ? (long long)(g.Hout - 1) * g.out_step * g.out_pix_w + (long long)(g.Wout - 1) * g.out_step : cells - 1;
    if (scatter && last_pix >= img_pix) return 1;
    // measured (tools/profile_train.py, batch 8): the scatter pays on the long-K gradients (res3a / res5a branch1: 0.096 -> 0.077,
    // 0.071 -> 0.063 ms) and loses on the short-K ones (res3a / res4a branch2a, K = 128 / 256: 0.061 -> 0.076, 0.037 -> 0.049 against
    // generation 2's 64-wide tiles): taken from K = 512 on
    if (scatter && !forced && d->Crun < 512) return 1;
    if (g.out_off != 0 || (!scatter && g.out_img_stride != cells * d->out_ld)) return 1;
    if ((long long)(g.Hout - 1) * d->sy >= g.Hin || (long long)(g.Wout - 1) * d->sx >= g.Win) return 1;
    if (g.in_elems * 2 >= (long long)G8_OOB || g.out_elems * 2 >= (long long)G8_OOB) return 1;
    const long long max_pix = (long long)(d->batch - 1) * img_pix + last_pix;             // the farthest pixel a row maps to
    if (g.out_elems < max_pix * d->out_ld + d->N) return 1;
    if ((epi & 1) && !res_up) {
        if (!g.res || ((uintptr_t)g.res & 15) || g.res_ld % 8 || g.res_img_stride != img_pix * g.res_ld) return 1;
        if (g.res_elems < max_pix * g.res_ld + d->N || g.res_elems * 2 >= (long long)G8_OOB) return 1;
    }
    if (res_up) {
        if (!g.res || ((uintptr_t)g.res & 15) || g.res_ld % 8 || g.res_img_stride % 8 || g.Hres < 1 || g.Wres < 1) return 1;
        if (g.res_img_stride < 0 || (long long)(d->batch - 1) * g.res_img_stride + ((long long)g.Hres * g.Wres - 1) * g.res_ld + d->N > g.res_elems) return 1;
        if (g.res_elems * 2 >= (long long)G8_OOB) return 1;
    }
    if (epi & 2) {
        if (!g.mask || ((uintptr_t)g.mask & 15) || g.mask_ld % 8 || g.mask_img_stride != img_pix * g.mask_ld) return 1;
        if (g.mask_elems < max_pix * g.mask_ld + d->N || g.mask_elems * 2 >= (long long)G8_OOB) return 1;
    }
    const long long in_max = (long long)(d->batch - 1) * g.in_img_stride + (long long)(g.Hout - 1) * d->sy * g.in_row_stride +
                             (long long)(g.Wout - 1) * d->sx * d->pix_stride + d->Crun;
    if (in_max > g.in_elems) return 1;
    long long Kel = d->Crun;
    G8Params p;
    memset(&p, 0, sizeof(p));
    p.s1.ptr = (const char*)g.in;
    p.s1.bytes = (unsigned)(g.in_elems * 2);
    p.s1.img_stride_b = g.in_img_stride * 2;
    p.s1.row_stride_b = (int)((long long)g.in_row_stride * 2);
    p.s1.pix_b = d->pix_stride * 2;
    p.s1.step = d->sy;
    p.nk1 = d->Crun * 2 / 128;
    if (s2) {
        if (!s2->in || ((uintptr_t)s2->in & 15) || s2->C < 1 || (s2->C * 2) % 128 || s2->step < 1) return 1;
        if ((long long)(g.Hout - 1) * s2->step >= s2->Hin || (long long)(g.Wout - 1) * s2->step >= s2->Win) return 1;
        const long long in2_max = (long long)(d->batch - 1) * s2->in_img_stride + (long long)(g.Hout - 1) * s2->step * s2->in_row_stride +
                                  (long long)(g.Wout - 1) * s2->step * s2->pix_stride + s2->C;
        if (in2_max > s2->in_elems || s2->in_elems * 2 >= (long long)G8_OOB) return 1;
        p.s2.ptr = (const char*)s2->in;
        p.s2.bytes = (unsigned)(s2->in_elems * 2);
        p.s2.img_stride_b = s2->in_img_stride * 2;
        p.s2.row_stride_b = (int)((long long)s2->in_row_stride * 2);
        p.s2.pix_b = s2->pix_stride * 2;
        p.s2.step = s2->step;
        Kel += s2->C;
    }
    const long long Kbytes = Kel * 2;
    if (Kbytes > 16384 || Kbytes * d->N >= (long long)G8_OOB) return 1;
    const int cus = h->num_cus > 0 ? h->num_cus : 256;
    const int ntn = n128 ? 1 : (d->N + 255) / 256;         // N = 128: the 128-column instance, one column tile
    const long long nk_all = Kbytes / 128;
    int mi = mi_force;
    double plain_cost = 0;                             // in units of one 64-row fragment row x one K step
    {                                                  // tile height by rounds x K steps x (rows + a fixed per-step cost)
        double best = 0;
        int pick = 0;
        for (int cand = 3; cand >= 2; --cand) {
            if (mi_force >= 2 && mi_force <= 3 && cand != mi_force) continue;
            const long long tl = ((M + 64 * cand - 1) / (64 * cand)) * ntn;
            const double cost = (double)((tl + cus - 1) / cus) * (cand + 0.4);
            if (pick == 0 || cost < best * 0.97) { best = cost; pick = cand; }
        }
        mi = pick;
        plain_cost = best * (double)nk_all;
    }
    // Stream-K: every workgroup an equal share of the tiles x K steps (see the header).  Taken when it saves at least TEN K steps per
    // workgroup against whole rounds of whole tiles - measured at batch 8, 800 x 1333 (profiles/r4_streamk_gemm8.txt): the hand-off costs
    // 8-10 us whatever the layer (every workgroup publishes one partial tile and owns one: 50 MB out and back at the end of the launch,
    // all owners at once), and a K step does not stay 1.25 us when more CUs run it - between 175 and 256 busy CUs it grows to 1.6-1.9 us
    // (shader clock 2.08 -> 2.01 GHz only; the staging rate of the chip, 8-9 TB/s L2 -> LDS, is the same in both cases).  So the layers
    // with 175 of 256 tiles (stage 4, C4_reduced: 5 steps saved) LOSE 4-7 us, and the ones on 44-88 CUs (stage 5 branch2a 21 steps,
    // C5_reduced 26, res5a branch2c + shortcut 15) gain 6-15 us.  Grid: the CUs, at most 8 pieces per tile.
    int sk_grid = 0;
    const bool sk_shape = !n128 && !(epi & 2) && !scatter && stagger && nk_all >= 2 && nk_all <= 256;
    if (sk_shape && sk_mode != 0 && (query || (sync && ws && !((uintptr_t)ws & 15) && !((uintptr_t)sync & 15)))) {
        double best = 0;
        int pick = 0, pick_grid = 0;
        for (int cand = 3; cand >= 2; --cand) {
            if (mi_force >= 2 && mi_force <= 3 && cand != mi_force) continue;
            const long long tl = ((M + 64 * cand - 1) / (64 * cand)) * ntn, total = tl * nk_all;
            long long G = cus;
            if (grid_limit > 0 && grid_limit < G) G = grid_limit;
            if (G > tl * 8) G = tl * 8;
            if (G > total) G = total;
            if (G > SK_ERR_WORD) G = SK_ERR_WORD;
            if (G < 1 || total >= (1ll << 30)) continue;
            const long long slab_need = G * (long long)(cand * 4 * 2) * G8_THREADS * 16;
            if (!query && slab_need > ws_cap) continue;
            if (slab_need >= (long long)G8_OOB) continue;
            const long long share_i = (total + G - 1) / G;
            const double share = (double)share_i;
            const bool whole = total % G == 0 && share_i % nk_all == 0;               // every range = whole tiles: nothing is handed over
            const double partners = (double)((nk_all + share_i - 1) / share_i);      // pieces the longest-cut tile has besides its head
            const double fix = whole ? 0.0 : 10.0 + 2.0 * (partners > 2.0 ? partners - 2.0 : 0.0);
            const double cost = (share + fix) * (cand + 0.4);
            if (pick == 0 || cost < best * 0.97) { best = cost; pick = cand; pick_grid = (int)G; }
        }
        if (pick && (sk_mode > 0 || best <= plain_cost)) { mi = pick; sk_grid = pick_grid; }
    }
    const long long ntm = (M + 64 * mi - 1) / (64 * mi), tiles = ntm * ntn;
    if (tiles > 0x3fffffff) return 1;
    if (query) {
        if (!sk_grid) return 1;                        // nothing to reserve: the caller's other paths answer the query
        *query = (size_t)sk_grid * (size_t)(mi * 4 * 2) * G8_THREADS * 16;
        return RTN_OK;
    }
    // too few tiles to be worth one workgroup per CU: below ~5/8 of the chip the 64-wide tiles of generation 2 (four times the
    // workgroups) win (res5 branch2a, 132 tiles: 0.049 ms here against 0.043; C5_reduced, 66 tiles: 0.046 against 0.041), above
    // it this kernel does (res4 branch2a, 175 tiles: 0.037 against 0.046).  (Round 3 re-measured: back-to-back launches of ONE layer
    // say the opposite for res5 branch2a - 0.034 against 0.039 - because its 2 MB of filters then sit in the L2; in the step's own
    // sequence, tools/profile_layers.py, it is 0.049 against 0.046-0.047 as before.  Judge such layers in sequence.)
    if (!forced && !sk_grid && tiles * 8 < cus * 5) return 1;
    p.w = (const char*)d->w;
    p.bias = d->bias;
    p.out = (char*)g.out;
    p.w_bytes = (unsigned)(Kbytes * d->N);
    p.out_bytes = (unsigned)(g.out_elems * 2);
    p.M = (int)M; p.N = d->N; p.Kbytes = (int)Kbytes; p.nk = (int)(Kbytes / 128);
    p.ntiles_m = (int)ntm; p.ntiles_n = ntn; p.ntiles = (int)tiles;
    p.Hout = g.Hout; p.Wout = g.Wout;
    p.inv_cells = 1.0f / (float)cells;
    p.inv_w = 1.0f / (float)g.Wout;
    p.relu = (d->flags & RTN_CONV_RELU) ? 1 : 0;
    p.out_ld = d->out_ld;
    if (epi & 1) { p.res = (const char*)g.res; p.res_bytes = (unsigned)(g.res_elems * 2); p.res_ld = g.res_ld; }
    if (epi & 2) { p.mask = (const char*)g.mask; p.mask_bytes = (unsigned)(g.mask_elems * 2); p.mask_ld = g.mask_ld; }
    p.mask_pre = (d->flags & RTN_CONV_MASK_PRE) ? 1 : 0;
    p.out_step = scatter ? g.out_step : 1; p.out_pix_w = g.out_pix_w; p.out_img_pix = (unsigned)img_pix;
    if (res_up) {
        p.res_up = 1; p.Hres = g.Hres; p.Wres = g.Wres;
        p.res_img_stride = (unsigned)g.res_img_stride;
        p.rs_h = (float)g.Hres / (float)g.Hout;
        p.rs_w = (float)g.Wres / (float)g.Wout;
    }
    int grid = cus;
    if (grid_limit > 0 && grid_limit < grid) grid = grid_limit;
    if (grid > p.ntiles) grid = p.ntiles;
    if (sk_grid) {
        grid = sk_grid;
        p.sk_total = (int)(tiles * nk_all);
        p.sk_slab = (char*)ws;
        p.sk_slab_bytes = (unsigned)((size_t)sk_grid * (size_t)(mi * 4 * 2) * G8_THREADS * 16);
        p.sk_flags = sync;
    }
#define RTN_G8_LAUNCH_SK(M_, DU, EP)                                                                     \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_gemm8_kernel<M_, true, DU, EP, 8, true>,    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));         \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_gemm8_kernel<M_, true, DU, EP, 8, true>), dim3((unsigned)grid), dim3(G8_THREADS), G8_LDS, h->stream, p); \
    } while (0)
#define RTN_G8_PICK_SK(M_)                                                                               \
    do {                                                                                                 \
        if (s2) RTN_G8_LAUNCH_SK(M_, true, 0);                                                           \
        else if (epi == 0) RTN_G8_LAUNCH_SK(M_, false, 0);                                               \
        else RTN_G8_LAUNCH_SK(M_, false, 1);                                                             \
    } while (0)
#define RTN_G8_LAUNCH(M_, ST, DU, EP)                                                                    \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_gemm8_kernel<M_, ST, DU, EP>,               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));         \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_gemm8_kernel<M_, ST, DU, EP>), dim3((unsigned)grid), dim3(G8_THREADS), G8_LDS, h->stream, p); \
    } while (0)
#define RTN_G8_PICK(M_)                                                                                  \
    do {                                                                                                 \
        if (s2) { if (stagger) RTN_G8_LAUNCH(M_, true, true, 0); else RTN_G8_LAUNCH(M_, false, true, 0); } \
        else if (!stagger && epi == 0) RTN_G8_LAUNCH(M_, false, false, 0);  /* lockstep variant: A/B only */ \
        else if (epi == 0) RTN_G8_LAUNCH(M_, true, false, 0);                                            \
        else if (epi == 1) RTN_G8_LAUNCH(M_, true, false, 1);                                            \
        else if (epi == 2) RTN_G8_LAUNCH(M_, true, false, 2);                                            \
        else RTN_G8_LAUNCH(M_, true, false, 3);                                                          \
    } while (0)
#define RTN_G8_LAUNCH4(M_, EP)                                                                           \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                    \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_gemm8_kernel<M_, true, false, EP, 4>,       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));         \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_gemm8_kernel<M_, true, false, EP, 4>), dim3((unsigned)grid), dim3(G8_THREADS), G8_LDS, h->stream, p); \
    } while (0)
#define RTN_G8_PICK4(M_)                                                                                 \
    do {                                                                                                 \
        if (epi == 0) RTN_G8_LAUNCH4(M_, 0); else if (epi == 1) RTN_G8_LAUNCH4(M_, 1);                   \
        else if (epi == 2) RTN_G8_LAUNCH4(M_, 2); else RTN_G8_LAUNCH4(M_, 3);                            \
    } while (0)
    const bool narrow = n128 && !s2 && stagger && rtn_env_int("RTN_CONV_G8_NARROW", 1) != 0;      // 0: N = 128 on the 256-column tile (A/B)
    if (sk_grid) { if (mi == 3) RTN_G8_PICK_SK(3); else RTN_G8_PICK_SK(2); }
    else if (narrow) { if (mi == 3) RTN_G8_PICK4(3); else RTN_G8_PICK4(2); }
    else if (mi == 3) RTN_G8_PICK(3); else RTN_G8_PICK(2);
#undef RTN_G8_PICK_SK
#undef RTN_G8_LAUNCH_SK
#undef RTN_G8_PICK4
#undef RTN_G8_LAUNCH4
#undef RTN_G8_PICK
#undef RTN_G8_LAUNCH
    RTN_CHECK_LAUNCH(h, "conv_gemm8_kernel");
    h->last_conv_streamk = sk_grid;
    h->last_conv_tile = ((64 * mi) << 16) | (narrow ? 128 : 256);
    return RTN_OK;
}

#ifdef RTN_G8_STAMP
extern "C" int rtn_debug_g8_stamps(unsigned long long* out8192) {
    return (int)hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_g8_stamps), sizeof(unsigned long long) * 8192, 0, hipMemcpyDeviceToHost);
}
#endif
