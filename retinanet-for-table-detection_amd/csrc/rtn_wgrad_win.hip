// rtn_wgrad_win.hip — the weight gradient of the stride-1 3x3 'same' layers with ALL NINE TAPS in one output tile
// (Conv2DBackpropFilter of the head towers, P3-P5 and the ResNet branch2b layers; RetinaNet.py:125-131,280 => TF autodiff of
// model/defineModel.py:101-117,155-163,183-203 and keras_resnet's bottleneck blocks):
//
//     dW[n][(kh, kw, c)] += sum over pixels m of  dY[m][n] * X[m + (kh - 1) * W + (kw - 1)][c]
//
// The per-tap kernels (rtn_backward.hip: 256 x 256 tile per tap; rtn_wgrad_halo.hip: one kernel row per tile) stage 7.6 / 5.3 KB of
// operands per MFLOP and read 0.75 / 0.83 transposed fragments per MFMA.  Here a workgroup owns 128 filters x 64 channels x 9 taps
// (288 KB of f32 accumulators = 144 registers per lane in 8 waves) and walks the pixels ONCE:
//   * the pixel stream of a level is PADDED with one non-existent pixel after every image row and one non-existent row after every
//     image (they stage as zeros), so that tap (kh, kw) of padded slot u is slot u + (kh - 1) * (W + 1) + (kw - 1) with no test at all;
//   * X lives in a RING of blocks of 64 slots (128 B = 64 channels per slot): one step = 64 slots; the step multiplies the 64 dY rows
//     that arrived with it against nine shifted 64-row views of the ring, which holds the slots from one image row above to one
//     below (2 D + 1 blocks, D = ceil((W + 2) / 64)).  Every X row is staged ONCE per workgroup (plus 2 D blocks of run-in per pixel
//     split): 384 B per slot for 147 KFLOP = 2.6 KB per MFLOP, and 13 fragment reads per 36 MFMAs (the 4 dY^T fragments of a
//     32-slot half step are shared by the 9 taps);
//   * LDS-DMA two steps ahead (dY: 3 stages of 64 x 256 B; X: the ring itself), one counted s_waitcnt and ONE barrier per step =
//     72 MFMAs (16x16x32 bf16) per wave; the two waves of a SIMD issue their three LDS-DMA pieces at different points of the step
//     (waves 0-3 behind the barrier, waves 4-7 between the half steps), so one of them keeps the matrix core fed meanwhile;
//   * fragments are read transposed (ds_read_b64_tr_b16: the reduction index, pixels, is the row index in memory); conflict-free
//     through an XOR of the 32-byte granule with key(row) = (row & 3) | ((row >> 3) & 1) << 2 on the 256-byte dY rows and
//     key(row) = ((row >> 1) & 1) | ((row >> 3) & 1) << 1 on the 128-byte X rows — both depend on row mod 16 only, so a view shifted by
//     any number of slots stays conflict-free.  Ring addresses wrap once per step and tap (add, subtract, min); the second row of
//     a fragment (+ 4 slots) and the second half step (+ 32 slots) use fixed distances from it, reading up to 4.6 KB past the ring's
//     end: ring block 0 is staged a second time behind the ring (the mirror);
//   * the level changes inside a pixel split without draining the pipeline: the steps of a split are a list of runs (level, first
//     block, last block), the first 2 D steps of a run only load;
//   * pixel splits x output tiles are laid out so that all tiles of a split run on ONE XCD (they read the same pixels: one L2);
//   * no atomics: every (split, tile) stores its accumulator fragments as they lie in the registers (16 B per lane, 1 KiB per store)
//     and rtn_wgrad_finish adds the splits in a fixed order and puts them in place; the bias gradient is one extra MFMA against a
//     ones fragment that the channel tiles x 4 waves sharing a dY fragment take in turns.
// LDS: X ring (2 D + 4 blocks of 8 KiB) + mirror 8 KiB + dY 48 KiB = 136 KiB for rows of up to 190 pixels (D = 3), 152 KiB up to 254.
// One workgroup of 8 waves = 2 (filter halves of 64) x 4 (channel slices of 16) per CU.
// PS = true is the form for layers with 64 filters (res2 branch2b; the head outputs, whose dY is padded to 64 columns and lies
// level after level inside one tensor): the tile is 64 filters x 64 channels x 9 taps, the two wave groups take the two 32-slot
// halves of a step instead of two filter halves and add their accumulators through LDS at the end; dY rows are 128 B (24 KiB of
// stages), which leaves room for a ring of 16 blocks: image rows of up to 382 pixels.
#include "rtn_internal.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr unsigned WW_OOB = 0xFFFFFF00u;
constexpr int WW_THREADS = 512;
constexpr int WW_SP = 64;                              // pixel slots per step
constexpr unsigned WW_XBLK = WW_SP * 128;              // one X block: 64 slots x 64 channels = 8 KiB
constexpr unsigned WW_MIRROR = WW_XBLK;                // ring block 0 once more behind the ring
constexpr int WW_LA = 2;                               // steps a load runs ahead of its use
constexpr int WW_NDY = WW_LA + 1;
constexpr int WW_LDS_MAX = 160 * 1024;
constexpr unsigned ww_dyst(bool ps) { return WW_SP * (ps ? 128u : 256u); }   // one dY stage: 64 slots x 128 (64) filters = 16 (8) KiB
constexpr int ww_nblk(int D) { return 2 * D + 1 + WW_LA + 1; }
constexpr int ww_lds_bytes(int D, bool ps) { return (int)(ww_nblk(D) * WW_XBLK + WW_MIRROR + WW_NDY * ww_dyst(ps)); }

struct WWSeg {
    const char* x;
    const char* dy;
    unsigned x_bytes, dy_bytes;
    unsigned x_img_b, dy_img_b;              // bytes between the images of x / dY
    int H, W, Mp;                            // Mp: slots of the padded stream, batch * (H + 1) * (W + 1)
    int stage_begin, nst, D;                 // first output stage (64 slots) of the level in the launch, their number, ceil((W + 2) / 64)
    unsigned cells_p, mg_cells, sh_cells;    // (H + 1) * (W + 1) and the multiply-shift pair dividing by it (exact below 2^24)
    unsigned mg_w1, sh_w1;                   // ... by W + 1
};

struct WWParams {
    WWSeg g[RTN_MAX_GROUPS];
    float* slab;                  // [S][tile][wave (PS: 4 waves)][tap][4][64 lanes][4]: accumulator fragments
    float* bslab;                 // [S * ncb][N] partial column sums of dY (fused BiasAddGrad) or null
    int ngroups, total_stages, stages_per_split, S;
    int ntiles, ncb;              // output tiles = (N / 128, PS: 1) x ncb channel blocks of 64
    int N, C, Ktot;
    int pix_b, dy_ld_b;
    unsigned xring;               // bytes of the X ring
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

__device__ __forceinline__ void dma16(const i32x4& srd, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(la), "s"(srd)
                 : "memory");
}

// transposed fragment: 4 + 4 reduction rows x 16 columns; the addresses are LDS byte offsets (the kernel's dynamic LDS starts at 0)
__device__ __forceinline__ s16x8 read_tr_at(unsigned lo, unsigned hi) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(__UINTPTR_TYPE__)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(__UINTPTR_TYPE__)hi);
    return (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// One cursor over the steps of a pixel split: run = the part of a level inside the split's range of output stages [glo, ghi);
// its steps are the X blocks k = s_lo .. s_hi - 1 + 2 D (block k = padded slots 64 (k - D) .. + 63), step k multiplies output stage
// k - 2 D when that is >= s_lo (the first 2 D steps of a run only fill the ring).
struct Cursor {
    int g, k, k_end, s_lo, D, done;
};

// EXP (timing experiments, wrong results): 1 = no staging in the loop, 2 = no fragment reads / MFMAs, 3 = neither
template <int EXP, bool PS, bool STAGGER = true>
__global__ __launch_bounds__(WW_THREADS) void conv_wgrad_win_kernel(const WWParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr unsigned WW_DYST = ww_dyst(PS), DY_ROW = PS ? 128u : 256u, DY_HALF = 32u * DY_ROW;
    constexpr int NPIECE = PS ? 2 : 3;                 // LDS-DMA pieces per wave and step (one more when the block has a mirror)
    const unsigned XRING = p.xring, DY_BASE = XRING + WW_MIRROR;
    // workgroup -> (output tile, pixel split): the tiles of one split sit on one XCD
    const int L = blockIdx.x, xcd = L & 7, jx = L >> 3;
    const int sl = jx / p.ntiles, tile = jx - sl * p.ntiles, split = sl * 8 + xcd;
    const int glo = split * p.stages_per_split;
    int ghi = glo + p.stages_per_split;
    ghi = ghi < p.total_stages ? ghi : p.total_stages;
    if (split >= p.S || glo >= ghi) return;
    const int tn = tile / p.ncb, cb = tile - tn * p.ncb;
    const int n0 = PS ? 0 : tn * 128, c0 = cb * 64;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wk = wave & 3;
    // the fragment reads below take LDS offsets as addresses: the dynamic LDS of a kernel without static LDS starts at 0
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds != 0u) __builtin_trap();

    auto next_run = [&](Cursor& c, int g_from) {
        c.done = 1;
        c.g = p.ngroups;
        for (int g = g_from; g < p.ngroups; ++g) {
            const int b = p.g[g].stage_begin, n = p.g[g].nst;
            const int lo = glo > b ? glo : b, hi = ghi < b + n ? ghi : b + n;
            if (lo < hi) {
                c.g = g; c.s_lo = lo - b; c.k = c.s_lo; c.D = p.g[g].D; c.k_end = hi - b + 2 * c.D; c.done = 0;
                break;
            }
        }
    };

    // ---- staging role: three pieces per wave and step.  dY: rows 4 wave + lane / 16 and that + 32 (4 rows of 256 B per piece),
    // LDS chunk position lane % 16.  X: rows 8 wave + lane / 8 (8 rows of 128 B), chunk position lane % 8; a fourth piece writes the
    // X rows once more behind the ring when the block is ring block 0.
    // (PS: dY rows are 128 B like the X rows: one piece of 8 rows per wave, the X swizzle.)
    const int x_row = 8 * wave + (lane >> 3);
    unsigned x_col, dy_col;
    const int dy_row = PS ? x_row : 4 * wave + (lane >> 4);
    {
        const int key = ((x_row >> 1) & 1) | (((x_row >> 3) & 1) << 1), cpos = lane & 7;
        const unsigned chunk = (unsigned)((((cpos >> 1) ^ key) << 1) | (cpos & 1));
        x_col = (unsigned)((c0 + chunk * 8) * 2);
        dy_col = chunk * 16u;
    }
    if (!PS) {
        const int key = (dy_row & 3) | (((dy_row >> 3) & 1) << 2), cpos = lane & 15;
        dy_col = (unsigned)((n0 + ((((cpos >> 1) ^ key) << 1) | (cpos & 1)) * 8) * 2);
    }
    Cursor lc;                                         // the staging cursor: WW_LA steps ahead of the multiplying one
    next_run(lc, 0);
    i32x4 xs = make_srd(p.g[0].x, p.g[0].x_bytes), ys = make_srd(p.g[0].dy, p.g[0].dy_bytes);
    int gH = 1, gW = 1, gMp = 0;
    unsigned g_cells = 1, g_mgc = 0, g_shc = 0, g_mgw = 0, g_shw = 0, g_ximg = 0, g_dyimg = 0;
    auto load_level = [&](int gi) {
        const WWSeg& G = p.g[gi < p.ngroups ? gi : 0];
        xs = make_srd(G.x, G.x_bytes);
        ys = make_srd(G.dy, G.dy_bytes);
        gH = G.H; gW = G.W; gMp = G.Mp; g_ximg = G.x_img_b; g_dyimg = G.dy_img_b;
        g_cells = G.cells_p; g_mgc = G.mg_cells; g_shc = G.sh_cells; g_mgw = G.mg_w1; g_shw = G.sh_w1;
    };
    load_level(lc.g);
    unsigned xl_ring = 0, dyl_ring = 0;
    // padded slot -> byte offset of its pixel (or WW_OOB for the padding column / row and for slots outside the level)
    auto slot_offset = [&](int u, bool live, unsigned img_b, unsigned pitch_b, unsigned col) -> unsigned {
        const unsigned uu = (unsigned)(u < 0 ? 0 : u);
        const unsigned b = __umulhi(uu, g_mgc) >> g_shc, rem = uu - b * g_cells;
        const unsigned y = __umulhi(rem, g_mgw) >> g_shw, x = rem - y * ((unsigned)gW + 1u);
        const bool ok = live && u >= 0 && u < gMp && x < (unsigned)gW && y < (unsigned)gH;
        return ok ? b * img_b + (rem - y) * pitch_b + col : WW_OOB;      // rem - y: the pixel's index in its image (one padding slot per completed row)
    };
    auto issue_step = [&]() {
        const bool live = !lc.done;
        const int s = lc.k - 2 * lc.D;
        const bool dy_live = live && s >= lc.s_lo;
        const unsigned v0 = slot_offset(s * WW_SP + dy_row, dy_live, g_dyimg, (unsigned)p.dy_ld_b, dy_col);
        dma16(ys, v0, DY_BASE + dyl_ring + (unsigned)wave * 1024u);
        if (!PS) {
            const unsigned v1 = slot_offset(s * WW_SP + 32 + dy_row, dy_live, g_dyimg, (unsigned)p.dy_ld_b, dy_col);
            dma16(ys, v1, DY_BASE + dyl_ring + 8192u + (unsigned)wave * 1024u);
        }
        const unsigned vx = slot_offset((lc.k - lc.D) * WW_SP + x_row, live, g_ximg, (unsigned)p.pix_b, x_col);
        dma16(xs, vx, xl_ring + (unsigned)wave * 1024u);
        if (xl_ring == 0) dma16(xs, vx, XRING + (unsigned)wave * 1024u);
        xl_ring = xl_ring == XRING - WW_XBLK ? 0u : xl_ring + WW_XBLK;
        dyl_ring = dyl_ring == (WW_NDY - 1) * WW_DYST ? 0u : dyl_ring + WW_DYST;
        if (live && ++lc.k == lc.k_end) {
            next_run(lc, lc.g + 1);
            if (!lc.done) load_level(lc.g);
        }
    };

    // ---- fragment roles (transposed reads): lane (g, q, pp) supplies rows 8 g + q and 8 g + q + 4 of a 32-slot half step and 4
    // channels.  dY^T fragment i = filters n0 + 64 wm + 16 i ..: fixed offsets inside the dY stage (second row + 1 KiB: same swizzle
    // key; second half step + 8 KiB).  X^T fragment of tap (kh, kw): rows shifted by (kh - 1)(W + 1) + (kw - 1) slots, counted from
    // the first row of the block that arrived with this step; the second row sits 512 B further, +- 64 B where the shift carries into
    // row bit 3; the second half step 4 KiB further.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    unsigned a_off[4];
    {
        const int row = 8 * g + q;
        const int key = PS ? ((row >> 1) & 1) | (((row >> 3) & 1) << 1) : (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i] = DY_BASE + (unsigned)(row * (int)DY_ROW + ((((PS ? 0 : wm * 4) + i) ^ key) << 5) + 8 * pp);
    }
    unsigned x_cst[9], x_dl[9];                        // per run: byte offset of tap j's first row relative to the step's block (swizzle folded in); second row - first
    Cursor cc;
    next_run(cc, 0);
    auto run_consts = [&]() {
        const int W1 = p.g[cc.g < p.ngroups ? cc.g : 0].W + 1;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int c = 8 * g + q + (kh - 1) * W1 + (kw - 1) - WW_SP * cc.D;       // in [-128 D, 63]
                const int c4 = c + 4;
                const int key = ((c >> 1) & 1) | (((c >> 3) & 1) << 1), key4 = ((c4 >> 1) & 1) | (((c4 >> 3) & 1) << 1);
                const int lo = c * 128 + ((wk ^ key) << 5) + 8 * pp, hi = c4 * 128 + ((wk ^ key4) << 5) + 8 * pp;
                x_cst[kh * 3 + kw] = (unsigned)lo;
                x_dl[kh * 3 + kw] = (unsigned)(hi - lo);
            }
    };
    run_consts();

    f32x4 acc[4][9], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bslab != nullptr;
    int bias_turn = 0;                                 // the channel tile whose waves add this step's dY rows into the bias gradient
    const s16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

    // ---- prologue: WW_LA steps in flight
#pragma unroll
    for (int i = 0; i < WW_LA; ++i) issue_step();
    unsigned xc_ring = XRING, dyc_ring = 0;            // xc_ring: ring offset of the step's block + XRING (keeps the sums below non-negative)
    const bool early = !STAGGER || wave < 4;           // the wave of each SIMD that stages right behind the barrier (STAGGER = false: both, measured 6.5 % slower)

    // one step: MUL = false for the load-only steps at the start of a run
    auto step = [&](auto mul_c) {
        constexpr bool MUL = decltype(mul_c)::value && !(EXP & 2);
        // the step has landed once at most the next step's pieces are in flight: 3 per wave, 4 when that step fills ring block 0
        if (xc_ring == 2 * XRING - WW_XBLK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's pieces; and every wave has left the step before
        if (!(EXP & 1) && (early || !MUL)) issue_step();      // into the dY stage / ring block the multiply loop no longer reads
        if (MUL) {
            unsigned xl[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const unsigned a = xc_ring + x_cst[j], a2 = a - XRING;         // mod XRING: the sum is in [0, 2 XRING)
                xl[j] = a < a2 ? a : a2;
            }
            const bool bias_now = do_bias && bias_turn == cb;
#pragma unroll
            for (int hs0 = 0; hs0 < (PS ? 1 : 2); ++hs0) {
                const unsigned hs = PS ? (unsigned)wm : (unsigned)hs0;           // PS: the wave group's own half of the step
                const unsigned xo = hs * 4096u, ao = dyc_ring + hs * DY_HALF;
                constexpr int ND = 2;                          // taps a B fragment is requested ahead of its MFMAs (3: no faster)
                s16x8 af[4], bf[ND + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = read_tr_at(ao + a_off[i], ao + a_off[i] + 4u * DY_ROW);
#pragma unroll
                for (int j = 0; j < ND; ++j) bf[j] = read_tr_at(xl[j] + xo, xl[j] + x_dl[j] + xo);
                if (bias_now)                          // BiasAddGrad on the matrix cores: dY^T x ones = the column sums of dY in every column
                    accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wk == 0 ? af[0] : wk == 1 ? af[1] : wk == 2 ? af[2] : af[3]),
                                                                   __builtin_bit_cast(bf16x8, ones), accb, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    if (j + ND < 9) bf[(j + ND) % (ND + 1)] = read_tr_at(xl[j + ND] + xo, xl[j + ND] + x_dl[j + ND] + xo);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j % (ND + 1)]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (PS && j == 3 && !(EXP & 1) && !early) issue_step();
                }
                if (!PS && hs0 == 0 && !(EXP & 1) && !early) issue_step();
            }
        }
        xc_ring = xc_ring == 2 * XRING - WW_XBLK ? XRING : xc_ring + WW_XBLK;
        dyc_ring = dyc_ring == (WW_NDY - 1) * WW_DYST ? 0u : dyc_ring + WW_DYST;
        bias_turn = bias_turn + 1 == p.ncb ? 0 : bias_turn + 1;
    };
    while (!cc.done) {
        const int n_load = cc.s_lo + 2 * cc.D - cc.k;  // load-only steps of this run (2 D)
#pragma unroll 1
        for (int i = 0; i < n_load; ++i) step(std::false_type{});
        const int n_mul = cc.k_end - cc.k - n_load;
#pragma unroll 1
        for (int i = 0; i < n_mul; ++i) step(std::true_type{});
        next_run(cc, cc.g + 1);
        if (!cc.done) run_consts();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS

    // ---- this split's tile -> slab[split][tile][wave][tap j][i][lane][4]: the accumulator fragments as they lie in the registers,
    // 16 B per lane and 1 KiB per store instruction; rtn_wgrad_finish puts them in place (frag layout)
    if (PS) {
        // the two wave groups hold the two halves of every step: waves 4-7 hand their sums to waves 0-3 through LDS (fixed order)
        __builtin_amdgcn_s_barrier();                  // every wave has left the multiply loop
        f32x4* ex = reinterpret_cast<f32x4*>(lds) + (size_t)wk * 37 * 64 + lane;
        if (wm == 1) {
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) ex[(j * 4 + i) * 64] = acc[i][j];
            ex[36 * 64] = accb;
        }
        __syncthreads();
        if (wm == 1) return;
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] += ex[(j * 4 + i) * 64];
        accb += ex[36 * 64];
    }
    float* sp = p.slab + (size_t)split * p.N * p.Ktot + ((size_t)(PS ? tile * 4 + wk : tile * 8 + wave) * 36) * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sp + (j * 4 + i) * 256) = acc[i][j];
    if (do_bias && (lane & 15) == 0) {
        const int lr = (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) p.bslab[((size_t)split * p.ncb + cb) * p.N + n0 + (PS ? 0 : wm * 64) + 16 * wk + lr + r] = accb[r];
    }
}

// q = umulhi(f, *mg) >> *sh == f / d for every f < 2^24 (d >= 2): mg = ceil(2^(32 + sh) / d) with 2^sh < d <= 2^(sh + 1)
void magic24(unsigned d, unsigned* mg, unsigned* sh) {
    unsigned s = 0;
    while ((2u << s) < d) ++s;
    *sh = s;
    *mg = (unsigned)((((unsigned long long)1 << (32 + s)) + d - 1) / d);
}

// Shape / layout test and the split plan (no handle: rtn_conv2d_wgrad_workspace_bytes has none; sized for the 256 CUs of an MI355X).
// *ps_out: the 64-filter form.
bool plan(const rtn_conv_desc_t* d, int* S_out, long long* stages_out, bool* ps_out) {
    if (d->dtype != RTN_BF16 || d->KH != 3 || d->KW != 3 || d->sy != 1 || d->sx != 1 || d->pad_t != 1 || d->pad_l != 1) return false;
    const bool ps = d->N == 64;
    if ((!ps && (d->N < 128 || d->N % 128)) || d->Crun < 64 || d->Crun % 64 || d->pix_stride != d->Crun || d->out_ld < d->N || d->out_ld % 8) return false;
    long long stages = 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (s.Hin != s.Hout || s.Win != s.Wout || s.in_row_stride != (long long)s.Win * d->pix_stride ||
            s.in_img_stride != (long long)s.Hin * s.in_row_stride) return false;
        const long long cells = (long long)s.Hout * s.Wout;
        // dY: pixels of an image out_ld apart, images out_img_stride apart, the level out_off into the tensor (head outputs)
        if (s.out_step > 1 || s.out_off < 0 || s.out_off % 8 || s.out_img_stride % 8 || s.out_img_stride < cells * d->out_ld) return false;
        if (s.Win < 1 || s.Hin < 1 || ww_lds_bytes((s.Win + 2 + WW_SP - 1) / WW_SP, ps) > WW_LDS_MAX) return false;
        if (s.in_elems * 2 >= (long long)WW_OOB || s.out_elems * 2 >= (long long)WW_OOB) return false;
        const long long Mp = (long long)d->batch * (s.Hout + 1) * (s.Wout + 1);        // the padded stream
        if (Mp >= (1ll << 24) - 64) return false;
        stages += (Mp + WW_SP - 1) / WW_SP;
    }
    const long long ntiles = (ps ? 1 : (long long)(d->N / 128)) * (d->Crun / 64);
    long long S = 8 * (32 / ntiles > 1 ? 32 / ntiles : 1);
    while (S > 8 && stages / S < 12) S -= 8;           // short pixel ranges: fewer, longer splits (every split pays 2 D steps of run-in)
    if (stages < 8) return false;
    *S_out = (int)S;
    *stages_out = stages;
    *ps_out = ps;
    return true;
}

}  // namespace

size_t rtn_wgrad_win_workspace_bytes(const rtn_conv_desc_t* d) {
    int S;
    long long stages;
    bool ps;
    if (!plan(d, &S, &stages, &ps)) return 0;
    return (size_t)S * d->N * ((size_t)9 * d->Crun + (size_t)(d->Crun / 64)) * sizeof(float);
}

// RTN_OK after the launches, 1 when the layer is not one this kernel takes, < 0 on error.
int rtn_wgrad_win_try(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, void* workspace, size_t workspace_bytes) {
    int S;
    long long stages;
    bool ps;
    if (!plan(d, &S, &stages, &ps)) return 1;
    if (!dW || ((uintptr_t)dW & 15) || !workspace || ((uintptr_t)workspace & 15)) return 1;
    const size_t need = rtn_wgrad_win_workspace_bytes(d);
    if (workspace_bytes < need) return 1;
    WWParams p;
    memset(&p, 0, sizeof(p));
    long long sb = 0;
    int Dmax = 1;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (!s.in || !s.out || ((uintptr_t)s.in & 15) || ((uintptr_t)s.out & 15)) return 1;
        const long long cells = (long long)s.Hout * s.Wout, M = cells * d->batch;
        if (s.in_elems < M * d->Crun) return 1;
        if (s.out_elems < (long long)(d->batch - 1) * s.out_img_stride + s.out_off + (cells - 1) * d->out_ld + d->N) return 1;
        WWSeg& g = p.g[i];
        g.x = (const char*)s.in;
        g.dy = (const char*)s.out + s.out_off * 2;
        g.x_bytes = (unsigned)(s.in_elems * 2);
        g.dy_bytes = (unsigned)((s.out_elems - s.out_off) * 2);
        g.x_img_b = (unsigned)(s.in_img_stride * 2);
        g.dy_img_b = (unsigned)(s.out_img_stride * 2);
        const long long Mp = (long long)d->batch * (s.Hin + 1) * (s.Win + 1);
        g.H = s.Hin; g.W = s.Win; g.Mp = (int)Mp;
        g.stage_begin = (int)sb;
        g.nst = (int)((Mp + WW_SP - 1) / WW_SP);
        g.D = (s.Win + 2 + WW_SP - 1) / WW_SP;
        Dmax = g.D > Dmax ? g.D : Dmax;
        g.cells_p = (unsigned)((s.Hin + 1) * (s.Win + 1));
        magic24(g.cells_p, &g.mg_cells, &g.sh_cells);
        magic24((unsigned)s.Win + 1u, &g.mg_w1, &g.sh_w1);
        sb += g.nst;
    }
    const int Ktot = 9 * d->Crun;
    p.ncb = d->Crun / 64;
    p.slab = (float*)workspace;
    p.bslab = db ? p.slab + (size_t)S * d->N * Ktot : nullptr;
    p.ngroups = d->ngroups;
    p.total_stages = (int)stages;
    p.stages_per_split = (int)((stages + S - 1) / S);
    p.S = S;
    p.ntiles = (ps ? 1 : d->N / 128) * p.ncb;
    p.N = d->N; p.C = d->Crun; p.Ktot = Ktot;
    p.pix_b = d->pix_stride * 2;
    p.dy_ld_b = d->out_ld * 2;
    // every split must own at least one stage: the finish adds all S slabs
    const int S_used = (int)((stages + p.stages_per_split - 1) / p.stages_per_split);
    int lds_bytes = ww_lds_bytes(Dmax, ps);
    if (lds_bytes > WW_LDS_MAX) return 1;
    if (ps && lds_bytes < 4 * 37 * 1024) lds_bytes = 4 * 37 * 1024;       // the exchange of the two wave groups' accumulators
    p.xring = (unsigned)ww_nblk(Dmax) * WW_XBLK;
    const unsigned grid = (unsigned)(p.ntiles * ((S + 7) / 8) * 8);
#define RTN_WW_LAUNCH(E_, PS_)                                                                                                    \
    do {                                                                                                                          \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                                                                                             \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                                                                                                          \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<E_, PS_>, hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS_MAX)); \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((conv_wgrad_win_kernel<E_, PS_>), dim3(grid), dim3(WW_THREADS), lds_bytes, h->stream, p);              \
    } while (0)
    // (timing ablations EXP = 1 / 2 / 3 and the unstaggered staging were round-3 A/Bs: profiles/r3_wgrad_win_ab.txt; only the
    // production instances are built)
    if (ps) RTN_WW_LAUNCH(0, true); else RTN_WW_LAUNCH(0, false);
#undef RTN_WW_LAUNCH
    RTN_CHECK_LAUNCH(h, "conv_wgrad_win_kernel");
    const rtn_wgrad_frag_t fr = {p.ncb, p.C, Ktot, ps ? 4 : 8, ps ? 64 : 128};
    return rtn_wgrad_finish(h, dW, p.slab, S_used, (long long)d->N * Ktot, db, p.bslab, d->N, db ? db_n : 0, S_used * p.ncb, &fr);
}
