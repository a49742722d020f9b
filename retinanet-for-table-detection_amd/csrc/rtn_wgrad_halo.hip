// rtn_wgrad_halo.hip — the weight gradient of the stride-1 3x3 'same' layers (Conv2DBackpropFilter of the head towers, P3-P5 and
// the ResNet branch2b layers with >= 128 channels; RetinaNet.py:125-131,280 => TF autodiff of model/defineModel.py:101-117,155-163,
// 183-203 and keras_resnet's bottleneck blocks):
//
//     dW[n][(kh, kw, c)] += sum over pixels m of  dY[m][n] * X[m + (kh - pad_t) * W + (kw - pad_l)][c]
//
// The general kernel (rtn_backward.hip, conv_wgrad_dma_kernel) gives every 256 x 256 output tile to a workgroup that stages 64 KB of
// operands per 64 pixels for each of its (n, k) tiles — each of the nine taps re-reads all pixels — waits for vmcnt(0) and a block
// barrier on every step, and adds its tile into dW with float atomics.  It runs at half the forward kernel's rate (0.31 ms per head
// layer at batch 8 against 0.15) and is bound by the L2->LDS volume.  This one:
//   * output tile = 128 filters x (one kernel row: 3 taps x 128 channels) = 128 x 384.  The three taps of a kernel row read ONE
//     staged image of 64 consecutive input pixels, shifted by one row per tap (the halo idea of rtn_conv_halo8.hip, on the B operand):
//     32 KB per 62 pixels for 6.1 MFLOP instead of 64 KB per 64 pixels for 8.4 MFLOP;
//   * a stage = 62 output pixels: 64 rows of dY (the last two are zeros) + 64 rows of X starting one pixel early.  Pixel slots whose
//     tap leaves the image row read a ZERO row instead (the reduction runs over pixels, so a zero operand row is a masked term);
//   * ring of 4 stages filled by LDS-DMA three stages ahead, addresses computed on the fly (no row-info table: no global loads in the
//     loop), ONE counted s_waitcnt and ONE barrier per stage (48 MFMAs per wave between barriers);
//   * fragments are read transposed (ds_read_b64_tr_b16: the reduction index, pixels, is the row index in memory), XOR swizzle on the
//     32-byte granule with key(row) = (row & 3) | ((row >> 3) & 1) << 2 as in the general kernel, on 256-byte rows;
//   * pixel splits x output tiles are laid out so that all tiles of a split run on ONE XCD (they read the same pixels: one L2);
//   * no atomics: every (split, tile) stores its f32 tile into slab[split][n][k] and wgrad_halo_finish_kernel adds the splits IN ORDER
//     into dW (and the fused bias gradient into db): repeated training steps give the same bits.
// LDS: 4 x (32 KiB + 512 B).  8 waves = 2 (filter halves of 64) x 4 (k quarters of 96 columns); 96 accumulator registers.
#include "rtn_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr unsigned WH_OOB = 0xFFFFFF00u;
constexpr int WH_THREADS = 512;
constexpr int WH_PX = 62;                             // output pixels per stage
constexpr unsigned WH_OP = 64 * 256;                  // one operand image: 64 rows x 256 B
constexpr unsigned WH_STAGE = 2 * WH_OP + 512;        // + X rows 64, 65: zeros, never staged (what the last two, empty, pixel slots read)
constexpr int WH_NST = 4;
constexpr int WH_LDS = WH_NST * WH_STAGE;

struct WHGroup {
    const char* x;
    const char* dy;
    unsigned x_bytes, dy_bytes;
    int H, W, Mp, stage_begin, x_row_b;      // Mp: pixels of the PADDED stream, batch * H * (W + 1)
    unsigned cells_p, mg_cells, sh_cells;    // H * (W + 1) and the multiply-shift pair dividing by it (exact below 2^24)
    unsigned mg_w1, sh_w1;                   // ... by W + 1
};

struct WHParams {
    WHGroup g[RTN_MAX_GROUPS];
    float* slab;                  // [S][N][Ktot]
    float* bslab;                 // [S][N] column sums of dY (fused BiasAddGrad) or null
    int ngroups, total_stages, stages_per_split, S;
    int ntiles, ncb;              // output tiles = (N / 128) x 3 kernel rows x ncb channel blocks of 128
    int N, C, Ktot;
    int pix_b, dy_ld_b, pad_t;
    int dbg;                      // RTN_WGRAD_HALO_DBG (timing ablations, wrong results): 1 = no staging in the loop, 2 = no fragment reads / MFMAs
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

__device__ __forceinline__ s16x8 read_tr(const char* lds0, unsigned off_lo, unsigned off_hi) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds0 + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds0 + off_hi));
    return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// The pixel stream of a level is PADDED: every image row is followed by one pixel that does not exist (it stages as zeros), so that
// the left neighbour of a row's first pixel and the right neighbour of its last one are zeros without any per-tap test in the
// multiply loop: the loop reads fixed LDS addresses.  Costs 1 / W more pixel slots (0.6 % on P3).
__global__ __launch_bounds__(WH_THREADS, 2) void conv_wgrad_halo_kernel(const WHParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // workgroup -> (output tile, pixel split): the tiles of one split sit on one XCD
    const int L = blockIdx.x, xcd = L & 7, jx = L >> 3;
    const int sl = jx / p.ntiles, tile = jx - sl * p.ntiles, split = sl * 8 + xcd;
    const int slo = split * p.stages_per_split;
    int shi = slo + p.stages_per_split;
    shi = shi < p.total_stages ? shi : p.total_stages;
    if (split >= p.S || slo >= shi) return;
    const int tn = tile / (3 * p.ncb), trem = tile - tn * 3 * p.ncb;
    const int kh = trem / p.ncb, cb = trem - kh * p.ncb;
    const int n0 = tn * 128, c0 = cb * 128;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wk = wave & 3;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    if (t < 32 * WH_NST) *reinterpret_cast<uint4*>(lds + (t >> 5) * WH_STAGE + 2 * WH_OP + (t & 31) * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // ---- staging role: pieces P = wave, wave + 8 (4 rows of 256 B each): row 4 P + lane / 16, LDS chunk position lane % 16
    const int srow0 = 4 * wave + (lane >> 4);          // second piece: + 32 (same swizzle key)
    constexpr int kb = 3;                              // (bit 4 instead - pairing lanes 0-15 with 32-47 - measured 7 % slower: the half-waves are lanes 0-31 / 32-63)
    const int skey = (srow0 & 3) | (((srow0 >> kb) & 1) << 2);
    const int cpos = lane & 15;
    const int schunk = (((cpos >> 1) ^ skey) << 1) | (cpos & 1);       // source chunk (8 elements) this lane fetches
    const unsigned dy_col = (unsigned)((n0 + schunk * 8) * 2), x_col = (unsigned)((c0 + schunk * 8) * 2);
    const int dyy = kh - p.pad_t;
    unsigned st_ring = 0;
    // the pyramid level the staging cursor is in, held in registers (scalar loads of p.g[gi] on every stage would sit on the critical path)
    i32x4 xs = make_srd(p.g[0].x, p.g[0].x_bytes), ys = make_srd(p.g[0].dy, p.g[0].dy_bytes);
    int gH = 1, gW = 1, gMp = 0, g_begin = 0, g_end = -1, g_xrow = 0;
    unsigned g_cells = 1, g_mgc = 0, g_shc = 0, g_mgw = 0, g_shw = 0;
    auto issue_stage = [&](int st) {
        if (st >= g_end) {                             // uniform, once per level
            int gi = 0;
#pragma unroll
            for (int i = 1; i < RTN_MAX_GROUPS; ++i)
                if (i < p.ngroups && st >= p.g[i].stage_begin) gi = i;
            const WHGroup& G = p.g[gi];
            xs = make_srd(G.x, G.x_bytes);
            ys = make_srd(G.dy, G.dy_bytes);
            gH = G.H; gW = G.W; gMp = G.Mp; g_begin = G.stage_begin; g_xrow = G.x_row_b;
            g_cells = G.cells_p; g_mgc = G.mg_cells; g_shc = G.sh_cells; g_mgw = G.mg_w1; g_shw = G.sh_w1;
            g_end = gi + 1 < p.ngroups ? p.g[gi + 1 < RTN_MAX_GROUPS ? gi + 1 : gi].stage_begin : 0x7fffffff;
        }
        const int t0 = (st - g_begin) * WH_PX;         // first padded pixel of the stage
        const unsigned W1 = (unsigned)gW + 1u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow0 + 32 * i;
            // X row r holds padded pixel u = t0 - 1 + r, dY row r padded pixel u + 1
            unsigned vx, vdy;
            {
                const int u = t0 - 1 + r;
                const unsigned uu = (unsigned)(u < 0 ? 0 : u);
                const unsigned b = __umulhi(uu, g_mgc) >> g_shc, rem = uu - b * g_cells;
                const unsigned y = __umulhi(rem, g_mgw) >> g_shw, x = rem - y * W1;
                const bool ok = st < shi && u >= 0 && u < gMp && x < (unsigned)gW && (unsigned)((int)y + dyy) < (unsigned)gH;
                const unsigned real = uu - (b * (unsigned)gH + y);             // one padding pixel per completed row
                vx = ok ? real * (unsigned)p.pix_b + (unsigned)(dyy * g_xrow) + x_col : WH_OOB;
            }
            {
                const unsigned uu = (unsigned)(t0 + r);
                const unsigned b = __umulhi(uu, g_mgc) >> g_shc, rem = uu - b * g_cells;
                const unsigned y = __umulhi(rem, g_mgw) >> g_shw, x = rem - y * W1;
                const bool ok = st < shi && r < WH_PX && (int)uu < gMp && x < (unsigned)gW;
                const unsigned real = uu - (b * (unsigned)gH + y);
                vdy = ok ? real * (unsigned)p.dy_ld_b + dy_col : WH_OOB;
            }
            const unsigned piece = (unsigned)((wave + 8 * i) * 1024);
            dma16(ys, vdy, lds_base + st_ring + piece);
            dma16(xs, vx, lds_base + st_ring + WH_OP + piece);
        }
        st_ring = st_ring == (WH_NST - 1) * WH_STAGE ? 0u : st_ring + WH_STAGE;
    };

    // ---- fragment roles (transposed reads): lane (g, q, pp) supplies row 8 g + q (+ 4) of each 32-pixel half and 4 channels.  All
    // addresses are fixed inside a stage: the row's bytes, its swizzle key and the 8 pp bytes folded into one offset per (row, tap).
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    unsigned a_off[2][2][4];                           // dY^T fragments i: filters n0 + 64 wm + 16 i + 4 pp ..; [half][lo / hi][i]
    unsigned x_off[2][2][3];                           // X^T: [half][lo / hi][kw], to be XORed with the fragment's granule << 5
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            const int row = 32 * s + 8 * g + q + 4 * hl;
            const int key = (row & 3) | (((row >> kb) & 1) << 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) a_off[s][hl][i] = (unsigned)(row * 256 + (((wm * 4 + i) ^ key) << 5) + 8 * pp);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int rr = row + kw;               // X row of pixel slot `row` under tap kw; rows 64, 65 are the stage's zero rows
                const int kr = (rr & 3) | (((rr >> kb) & 1) << 2);
                x_off[s][hl][kw] = (WH_OP + (unsigned)(rr * 256 + 8 * pp)) ^ ((unsigned)kr << 5);
            }
        }
    // X^T fragments j = 0..5 of this wave: tap kw = j / 2, channel granule (16 channels) cf = 2 wk + j % 2
    const unsigned cf5[2] = {(unsigned)(2 * wk) << 5, (unsigned)(2 * wk + 1) << 5};

    f32x4 acc[4][6], accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.bslab != nullptr && kh == 0 && cb == 0 && wk == 0;      // wave-uniform
    const s16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

    // ---- prologue: three stages in flight.  (A two-phase variant with the wave groups one barrier apart - reads + staging against
    // MFMAs, as in rtn_conv_halo8.hip - measured 7-12 % slower here: its read phase carries the four LDS-DMA issues of a stage,
    // ~100-185 cycles each beside ds_reads (MI355X_MICROARCH.md), and is longer than the 48 MFMAs it should hide behind.)
    issue_stage(slo);
    issue_stage(slo + 1);
    issue_stage(slo + 2);
    unsigned c_ring = 0;
#pragma unroll 1
    for (int st = slo; st < shi; ++st) {
        // stage `st` has landed once at most the two younger stages' pieces (4 per wave each) are in flight
        if (p.dbg & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's pieces; and every wave has left the stage before
        if (!(p.dbg & 1)) issue_stage(st + 3);         // into the ring slot the stage before occupied
        if (p.dbg & 2) continue;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            s16x8 af[4], bf[6];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_tr(lds, c_ring + a_off[s][0][i], c_ring + a_off[s][1][i]);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const unsigned ol = c_ring + x_off[s][0][kw], oh = c_ring + x_off[s][1][kw];
                bf[2 * kw] = read_tr(lds, ol ^ cf5[0], oh ^ cf5[0]);
                bf[2 * kw + 1] = read_tr(lds, ol ^ cf5[1], oh ^ cf5[1]);
            }
            if (do_bias) {                             // BiasAddGrad on the matrix cores: dY^T x ones = the column sums of dY in every column
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, ones), accb[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]),
                                                                        acc[i][j], 0, 0, 0);
        }
        c_ring = c_ring == (WH_NST - 1) * WH_STAGE ? 0u : c_ring + WH_STAGE;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS

    // ---- this split's tile -> slab[split][n][(kh, kw, c)]
    const int lr = (lane >> 4) * 4, lc = lane & 15;
    float* sp = p.slab + (size_t)split * p.N * p.Ktot;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int kcol = (kh * 3 + (j >> 1)) * p.C + c0 + 16 * (2 * wk + (j & 1)) + lc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * 64 + 16 * i + lr + r;
                sp[(size_t)n * p.Ktot + kcol] = acc[i][j][r];
            }
    }
    if (do_bias && lc == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) p.bslab[(size_t)split * p.N + n0 + wm * 64 + 16 * i + lr + r] = accb[i][r];
    }
}

// dW[i] += slab[0][i] + slab[1][i] + ... + slab[S-1][i] with a FIXED association: the S splits are cut into SL consecutive ranges, every
// range is summed in order by one "split lane" (SL lanes x 256 / SL float4 columns per block), and the SL partial sums are added in
// lane order.  Same bits on every run; SL only spreads the loads of a many-split, few-weights layer (res2: 256 splits x 16 K weights)
// over enough threads.  Blocks past `main_blocks` do the same for the bias slabs (scalar columns).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(float* __restrict__ dW, const float* __restrict__ slab, int S, long long NK,
                                                           float* __restrict__ db, const float* __restrict__ bslab, int N, int db_n, int main_blocks,
                                                           int bS, rtn_wgrad_frag_t fr) {
    constexpr int CB = 256 / SL;
    __shared__ float4 part[SL][CB];
    const int c = threadIdx.x % CB, sl = threadIdx.x / CB;
    const bool is_main = (int)blockIdx.x < main_blocks;
    const int Sx = is_main ? S : bS;                      // the bias slabs may come in a different number of parts (rtn_wgrad_win.hip)
    const int per = (Sx + SL - 1) / SL;
    const int s0 = sl * per < Sx ? sl * per : Sx, s1 = s0 + per < Sx ? s0 + per : Sx;
    if (is_main) {
        const long long i = (long long)blockIdx.x * CB + c;          // float4 column
        const bool live = i < NK / 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
#pragma unroll 8
            for (int s = s0; s < s1; ++s) {                               // (unrolled: the loads of eight splits in flight, the sums in the same order)
                const float4 a = reinterpret_cast<const float4*>(slab + (size_t)s * NK)[i];
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
        part[sl][c] = v;
        __syncthreads();
        if (sl == 0 && live) {
#pragma unroll
            for (int l = 1; l < SL; ++l) { const float4 a = part[l][c]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
            if (fr.ncb > 0) {
                // slabs in accumulator-fragment order (rtn_wgrad_win.hip): float4 i = [tile][wave][tap j][i4][lane] holds rows
                // n .. n + 3 of ONE column: n = co_tile tn + 64 wm + 16 i4 + 4 (lane / 16), k = j C + 64 cb + 16 wk + lane % 16
                const int lane = (int)(i & 63);
                const unsigned f = (unsigned)(i >> 6);                  // NK < 2^32
                const int i4 = (int)(f & 3);
                const unsigned t2 = f >> 2, t3 = t2 / 9u;
                const int j = (int)(t2 - t3 * 9u);
                const int tile = (int)(t3 / (unsigned)fr.wpt), wave = (int)t3 - tile * fr.wpt;
                const int tn = tile / fr.ncb, cb = tile - tn * fr.ncb;
                const int n = fr.co_tile * tn + 64 * (wave >> 2) + 16 * i4 + 4 * (lane >> 4);
                const int k = j * fr.C + 64 * cb + 16 * (wave & 3) + (lane & 15);
                float* o = dW + (size_t)n * fr.Ktot + k;
                o[0] += v.x; o[fr.Ktot] += v.y; o[2 * (size_t)fr.Ktot] += v.z; o[3 * (size_t)fr.Ktot] += v.w;
            } else {
                float4 w = reinterpret_cast<float4*>(dW)[i];
                w.x += v.x; w.y += v.y; w.z += v.z; w.w += v.w;
                reinterpret_cast<float4*>(dW)[i] = w;
            }
        }
    } else {
        const int n = ((int)blockIdx.x - main_blocks) * CB + c;
        const bool live = n < db_n;
        float v = 0.f;
        if (live)
            for (int s = s0; s < s1; ++s) v += bslab[(size_t)s * N + n];
        part[sl][c].x = v;
        __syncthreads();
        if (sl == 0 && live) {
#pragma unroll
            for (int l = 1; l < SL; ++l) v += part[l][c].x;
            db[n] += v;
        }
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
bool plan(const rtn_conv_desc_t* d, int* S_out, long long* stages_out) {
    if (d->dtype != RTN_BF16 || d->KH != 3 || d->KW != 3 || d->sy != 1 || d->sx != 1) return false;
    if (d->N < 128 || d->N % 128 || d->Crun < 128 || d->Crun % 128 || d->pix_stride != d->Crun || d->out_ld < d->N || d->out_ld % 8) return false;
    if (d->pad_t < 0 || d->pad_t > 2 || d->pad_l < 0 || d->pad_l > 2) return false;
    long long stages = 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (s.Hin != s.Hout || s.Win != s.Wout || s.in_row_stride != (long long)s.Win * d->pix_stride ||
            s.in_img_stride != (long long)s.Hin * s.in_row_stride) return false;
        const long long cells = (long long)s.Hout * s.Wout, M = cells * d->batch;
        if (s.out_off != 0 || s.out_img_stride != cells * d->out_ld || s.out_step > 1) return false;          // dense dY
        if (M < 1 || M >= (1ll << 24)) return false;
        if (s.in_elems * 2 >= (long long)WH_OOB || s.out_elems * 2 >= (long long)WH_OOB) return false;
        const long long Mp = (long long)d->batch * s.Hout * (s.Wout + 1);      // the padded stream (one zero pixel per image row)
        if (Mp >= (1ll << 24)) return false;
        stages += (Mp + WH_PX - 1) / WH_PX;
    }
    if (d->pad_l != 1) return false;                   // one padding pixel per row = one column of left / right padding
    const long long ntiles = (long long)(d->N / 128) * 3 * (d->Crun / 128);
    long long S = 8 * (32 / ntiles > 1 ? 32 / ntiles : 1);
    while (S > 8 && stages / S < 6) S -= 8;            // short pixel ranges: fewer, longer splits
    if (stages < 8) return false;
    *S_out = (int)S;
    *stages_out = stages;
    return true;
}

}  // namespace

// dW[0..NK) += slab[0] + slab[1] + ... + slab[S-1] (in that order), db[0..db_n) likewise from bslab[S][N]: the ordered reduction
// of the pixel splits of every weight-gradient kernel
int rtn_wgrad_finish(rtn_handle_t h, float* dW, const float* slab, int S, long long NK, float* db, const float* bslab, int N, int db_n, int bS,
                     const rtn_wgrad_frag_t* frag) {
    if (!dW || !slab || S < 1 || NK < 4 || NK % 4) return rtn_fail(h, RTN_EINVAL, "wgrad finish: bad argument");
    const int nb = db ? db_n : 0;
    if (bS < 1) bS = S;
    rtn_wgrad_frag_t fr = {0, 0, 0, 8, 128};
    if (frag) fr = *frag;
#define RTN_WF(SL_)                                                                                               \
    do {                                                                                                          \
        const long long mb = (NK / 4 + 256 / SL_ - 1) / (256 / SL_);                                              \
        const long long bb = (nb + 256 / SL_ - 1) / (256 / SL_);                                                  \
        hipLaunchKernelGGL((wgrad_finish_kernel<SL_>), dim3((unsigned)(mb + bb)), dim3(256), 0, h->stream, dW, slab, S, NK, db, bslab, N, nb, (int)mb, bS, fr); \
    } while (0)
    if (S >= 64) RTN_WF(16); else if (S >= 16) RTN_WF(8); else if (S >= 8) RTN_WF(4); else RTN_WF(1);
#undef RTN_WF
    RTN_CHECK_LAUNCH(h, "wgrad_finish_kernel");
    return RTN_OK;
}

size_t rtn_wgrad_halo_workspace_bytes(const rtn_conv_desc_t* d) {
    int S;
    long long stages;
    if (!plan(d, &S, &stages)) return 0;
    return (size_t)S * d->N * ((size_t)9 * d->Crun + 1) * sizeof(float);
}

// RTN_OK after the launches, 1 when the layer is not one this kernel takes, < 0 on error.
int rtn_wgrad_halo_try(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, void* workspace, size_t workspace_bytes) {
    int S;
    long long stages;
    if (!plan(d, &S, &stages)) return 1;
    if (!dW || ((uintptr_t)dW & 15) || !workspace || ((uintptr_t)workspace & 15)) return 1;
    const size_t need = rtn_wgrad_halo_workspace_bytes(d);
    if (workspace_bytes < need) return 1;
    WHParams p;
    memset(&p, 0, sizeof(p));
    long long sb = 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (!s.in || !s.out || ((uintptr_t)s.in & 15) || ((uintptr_t)s.out & 15)) return 1;
        const long long cells = (long long)s.Hout * s.Wout, M = cells * d->batch;
        if (s.in_elems < M * d->Crun || s.out_elems < (M - 1) * d->out_ld + d->N) return 1;
        WHGroup& g = p.g[i];
        g.x = (const char*)s.in;
        g.dy = (const char*)s.out;
        g.x_bytes = (unsigned)(s.in_elems * 2);
        g.dy_bytes = (unsigned)(s.out_elems * 2);
        const long long Mp = (long long)d->batch * s.Hin * (s.Win + 1);
        g.H = s.Hin; g.W = s.Win; g.Mp = (int)Mp;
        g.stage_begin = (int)sb;
        g.x_row_b = (int)(s.in_row_stride * 2);
        g.cells_p = (unsigned)(s.Hin * (s.Win + 1));
        magic24(g.cells_p, &g.mg_cells, &g.sh_cells);
        magic24((unsigned)s.Win + 1u, &g.mg_w1, &g.sh_w1);
        sb += (Mp + WH_PX - 1) / WH_PX;
    }
    const int Ktot = 9 * d->Crun;
    p.slab = (float*)workspace;
    p.bslab = db ? p.slab + (size_t)S * d->N * Ktot : nullptr;
    p.ngroups = d->ngroups;
    p.total_stages = (int)stages;
    p.stages_per_split = (int)((stages + S - 1) / S);
    p.S = S;
    p.ncb = d->Crun / 128;
    p.ntiles = (d->N / 128) * 3 * p.ncb;
    p.N = d->N; p.C = d->Crun; p.Ktot = Ktot;
    p.pix_b = d->pix_stride * 2;
    p.dy_ld_b = d->out_ld * 2;
    p.pad_t = d->pad_t;
    p.dbg = rtn_env_int("RTN_WGRAD_HALO_DBG", 0);
    // every split must own at least one stage: the finish adds all S slabs
    const int S_used = (int)((stages + p.stages_per_split - 1) / p.stages_per_split);
    static bool attr_set = false;
    if (!attr_set) {
        RTN_HIP(h, hipFuncSetAttribute((const void*)conv_wgrad_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WH_LDS));
        attr_set = true;
    }
    const unsigned grid = (unsigned)(p.ntiles * ((S + 7) / 8) * 8);
    hipLaunchKernelGGL(conv_wgrad_halo_kernel, dim3(grid), dim3(WH_THREADS), WH_LDS, h->stream, p);
    RTN_CHECK_LAUNCH(h, "conv_wgrad_halo_kernel");
    return rtn_wgrad_finish(h, dW, p.slab, S_used, (long long)d->N * Ktot, db, p.bslab, d->N, db ? db_n : 0);
}
