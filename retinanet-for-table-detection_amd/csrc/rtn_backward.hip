// rtn_backward.hip — the backward / optimizer half of the training step (RetinaNet.py:125-131,280: Keras compile +
// fit_generator => TF autodiff of every conv, Adam(lr, clipnorm=0.001)).
//
//   rtn_conv2d_wgrad      dW[n][(kh,kw,c)] += sum_pixels dY[pix][n] * X[tap(pix,kh,kw)][c]   (Conv2DBackpropFilter)
//   rtn_bias_grad         db[n] += sum_pixels dY[pix][n]
//   rtn_zero_insert2      dY -> zero-inserted dY for the data gradient of the stride-2 3x3 convs (P6, P7)
//   rtn_upsample_add_bwd  adjoint of UpsampleLike (legacy-TF nearest) + Add  (model/layers.py:89-98)
//   rtn_maxpool3x3s2_tfsame_bwd
//   rtn_sumsq / rtn_adam_clipnorm_step   global-norm clip + Adam on the flat parameter buffer
//
// wgrad on MFMA: the reduction runs over PIXELS, so both operands are needed pixel-major per lane while memory is
// channel-major.  Tiles [64 pixels][16 x 16-byte chunks] are staged as they lie in memory (range-checked buffer loads:
// padding taps and tail pixels read zeros) and the fragments are read TRANSPOSED with ds_read_b64_tr_b16 (gfx950): one
// read hands a lane 4 consecutive pixels of its channel.  Rows are padded to 288 B and rows 8..15 of every 16 swap
// their 128-byte halves, which makes the transposed reads of a 32-lane half conflict-free.  The pixel range is split
// over gridDim.y; partial tiles are added into the f32 gradient with global_atomic_add_f32.
#include "rtn_internal.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;
constexpr int WG_ROWB = 288;                 // LDS bytes per pixel row (256 B of channels + 32 B pad)
constexpr int WG_TILE_B = 64 * WG_ROWB;      // one operand tile

__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct WGroup {
    const char* x;
    const char* dy;
    unsigned x_bytes, dy_bytes;
    long long x_img_stride_b, dy_img_stride_b, dy_off_b;
    int x_row_stride_b;
    int Hin, Win, Hout, Wout;
    int M;
    int tile_begin;      // first 64-pixel tile of the group
};

struct WParams {
    WGroup g[RTN_MAX_GROUPS];
    float* dW;
    float* db;           // optional bias gradient (BiasAddGrad fused: column sums of dY), accumulated by the tile_k == 0 blocks
    float* slab;         // not null: every pixel split stores its tile into slab[split][N][Ktot] (plain stores) and wgrad_finish adds the
    float* bslab;        // splits in order into dW / db - no atomics, the same bits on every run; bslab[split][N] for the bias sums
    int db_n;            // valid bias entries
    const uint4* rowinfo;
    int ngroups, total_tiles, tiles_per_split, ntiles_k;
    int out_tiles, xcd_map;   // xcd_map: 1-D grid, the output tiles of one pixel split share an XCD (its L2)
    int N, Ktot;
    int cshift, crun_mask, kw_inv, KW;
    int pix_stride_b, sy, sx, pad_t, pad_l, dy_ld_b;
};

// per output pixel: {byte offset of tap (0,0) in X (mod 2^32), byte offset of the dY row, iy0 | ix0 << 16, -}
template <int ES>
__global__ __launch_bounds__(256) void wgrad_rowinfo_kernel(const WParams p, uint4* __restrict__ info) {
    const long long total = (long long)p.total_tiles * 64;
    for (long long R = (long long)blockIdx.x * blockDim.x + threadIdx.x; R < total; R += (long long)gridDim.x * blockDim.x) {
        const int tile = (int)(R >> 6);
        int gi = 0;
        for (int i = 1; i < RTN_MAX_GROUPS; ++i)
            if (i < p.ngroups && tile >= p.g[i].tile_begin) gi = i;
        const WGroup& G = p.g[gi];
        const int m = (int)(R - (long long)G.tile_begin * 64);
        uint4 o = make_uint4(0u, OOB_OFFSET, 0x00008000u, 0u);          // iy0 = -32768: every tap out of range
        if (m < G.M) {
            const int cells = G.Hout * G.Wout;
            const int b = m / cells;
            const int cell = m - b * cells;
            const int oy = cell / G.Wout, ox = cell - oy * G.Wout;
            const int iy0 = oy * p.sy - p.pad_t, ix0 = ox * p.sx - p.pad_l;
            o.x = (unsigned)((long long)b * G.x_img_stride_b + (long long)iy0 * G.x_row_stride_b + (long long)ix0 * p.pix_stride_b);
            o.y = (unsigned)((long long)b * G.dy_img_stride_b + G.dy_off_b + (long long)cell * p.dy_ld_b);
            o.z = ((unsigned)iy0 & 0xffffu) | ((unsigned)ix0 << 16);
        }
        info[R] = o;
    }
}

// One 64-pixel step of the MFMA loop: transposed fragment reads of the dY tile (A) and the X tile (B) of one LDS buffer.
template <int ES, int FI, int WT>
__device__ __forceinline__ void wgrad_step(const char* A, const char* B, int lane, int wm, int wn, f32x4 (&acc)[FI][FI]) {
    if constexpr (ES == 2) {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int swz = (g & 1) << 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = 32 * s + 8 * g + q;
            s16x8 af[FI], bf[FI];
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int ca = (wm * WT + 16 * i + 4 * pp) * 2;       // byte column inside the 256-byte row
                const int aoff = row * WG_ROWB + ((((ca >> 4) ^ swz)) << 4) + (ca & 15);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(A + aoff));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(A + aoff + 4 * WG_ROWB));
                af[i] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const int cb = (wn * WT + 16 * i + 4 * pp) * 2;
                const int boff = row * WG_ROWB + ((((cb >> 4) ^ swz)) << 4) + (cb & 15);
                const s16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(B + boff));
                const s16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(B + boff + 4 * WG_ROWB));
                bf[i] = (s16x8){lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
            }
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                        __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
        }
    } else {
        const int li = lane & 15, kq = lane >> 4;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int row = 4 * s + kq;
            const int swz = ((row >> 3) & 1) << 3;
            float af[FI], bf[FI];
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int ca = (wm * WT + 16 * i + li) * 4;
                af[i] = *reinterpret_cast<const float*>(A + row * WG_ROWB + (((ca >> 4) ^ swz) << 4) + (ca & 15));
                const int cb = (wn * WT + 16 * i + li) * 4;
                bf[i] = *reinterpret_cast<const float*>(B + row * WG_ROWB + (((cb >> 4) ^ swz) << 4) + (cb & 15));
            }
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
}

// Loads run TWO pixel tiles ahead of the MFMAs (two register sets: the tile staged into LDS at the end of a step was
// requested a whole step earlier).  Workgroups that read the same pixel range (all output tiles of one split) are placed
// on one XCD (p.xcd_map) so they share its L2.  Knob-ablated on the batch-16 training step (wgrad ~13.5 ms of 38.7):
// operand loads ~5-9 ms, MFMA + transposed reads ~0.7-4 ms (hidden behind the loads), float atomics 1.9 ms - every one of the
// (N/128)*(K/128) output tiles re-reads all pixels (6.4 GB of L2->LDS traffic per head layer at 65 FLOP/B), which is what
// bounds this kernel.
template <int ES>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WParams p) {
    constexpr int CE = 16 / ES;                // elements per 16-byte chunk
    constexpr int CH = 16 * CE;                // channels per tile side: 128 (bf16) / 64 (f32)
    constexpr int WT = CH / 2;                 // per-wave extent
    constexpr int FI = WT / 16;                // 16x16 MFMA tiles per wave side: 4 / 2
    __shared__ __attribute__((aligned(16))) char lds[4 * WG_TILE_B];   // [buf][dY | X]

    int otile, split;
    if (p.xcd_map) {                           // 1-D grid; gridDim.x = out_tiles * nsplit, nsplit a multiple of 8
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int sl = j / p.out_tiles;
        otile = j - sl * p.out_tiles;
        split = sl * 8 + xcd;
    } else {
        otile = blockIdx.x;
        split = blockIdx.y;
    }
    const int tile_n = otile / p.ntiles_k;
    const int tile_k = otile - tile_n * p.ntiles_k;
    const int n0 = tile_n * CH, k0 = tile_k * CH;
    const int tlo = split * p.tiles_per_split;
    int thi = tlo + p.tiles_per_split;
    thi = thi < p.total_tiles ? thi : p.total_tiles;
    if (tlo >= thi) return;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ch = t & 15, r0 = t >> 4;
    const int st_off = r0 * WG_ROWB + ((ch ^ (((r0 >> 3) & 1) << 3)) << 4);

    // this thread's X chunk: fixed tap and channel offset for the whole kernel
    const int kk = k0 + ch * CE;
    const bool kvalid = kk < p.Ktot;
    const int kpos = kk >> p.cshift;
    const int coff = kk & p.crun_mask;
    const int kh = (kpos * p.kw_inv) >> 16;
    const int kw = kpos - kh * p.KW;
    const int nn = n0 + ch * CE;
    const bool nvalid = nn < p.N;
    const unsigned dyo = (unsigned)(nn * ES);

    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[FI][FI];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool do_bias = p.db != nullptr && tile_k == 0;
    float bsum[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) bsum[j] = 0.f;
    uint4 ra0[4], rb0[4], ra1[4], rb1[4];
    uint4 ri[4];                                  // row info of the NEXT tile to load (prefetched one load ahead)
#define WG_ROWINFO(TILE)                                                                                            \
    {                                                                                                               \
        const int tt_ = (TILE) < thi ? (TILE) : thi - 1;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) ri[i_] = p.rowinfo[(long long)tt_ * 64 + r0 + 16 * i_];    \
    }
#define WG_LOAD(TILE, RA, RB)                                                                                       \
    {                                                                                                               \
        int gi_ = 0;                                                                                                \
        _Pragma("unroll") for (int i_ = 1; i_ < RTN_MAX_GROUPS; ++i_)                                               \
            if (i_ < p.ngroups && (TILE) >= p.g[i_].tile_begin) gi_ = i_;                                           \
        const WGroup& G_ = p.g[gi_];                                                                                \
        const __amdgpu_buffer_rsrc_t xs_ = __builtin_amdgcn_make_buffer_rsrc(                                       \
            (void*)G_.x, 0, (int)__builtin_amdgcn_readfirstlane((int)G_.x_bytes), 0x00020000);                      \
        const __amdgpu_buffer_rsrc_t ys_ = __builtin_amdgcn_make_buffer_rsrc(                                       \
            (void*)G_.dy, 0, (int)__builtin_amdgcn_readfirstlane((int)G_.dy_bytes), 0x00020000);                    \
        const unsigned delta_ = (unsigned)(kh * G_.x_row_stride_b + kw * p.pix_stride_b + coff * ES);               \
        const int Hin_ = G_.Hin, Win_ = G_.Win;                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
            const int iy_ = (int)(short)(ri[i_].z & 0xffffu) + kh, ix_ = (int)(short)(ri[i_].z >> 16) + kw;         \
            const bool ok_ = kvalid && (unsigned)iy_ < (unsigned)Hin_ && (unsigned)ix_ < (unsigned)Win_;            \
            RB[i_] = buffer_load16(xs_, ok_ ? ri[i_].x + delta_ : OOB_OFFSET);                                      \
            RA[i_] = buffer_load16(ys_, (nvalid && ri[i_].y != OOB_OFFSET) ? ri[i_].y + dyo : OOB_OFFSET);          \
        }                                                                                                           \
        WG_ROWINFO((TILE) + 1);                                                                                     \
    }
#define WG_STORE(BUF, RA, RB)                                                                                       \
    {                                                                                                               \
        if (do_bias) {                                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                      \
                const unsigned w4_[4] = {RA[i_].x, RA[i_].y, RA[i_].z, RA[i_].w};                                   \
                if constexpr (ES == 2) {                                                                            \
                    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                              \
                        bsum[2 * j_] += __uint_as_float(w4_[j_] << 16);                                             \
                        bsum[2 * j_ + 1] += __uint_as_float(w4_[j_] & 0xffff0000u);                                 \
                    }                                                                                               \
                } else {                                                                                            \
                    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) bsum[j_] += __uint_as_float(w4_[j_]);          \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
        char* A_ = lds + (BUF) * 2 * WG_TILE_B;                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
            *reinterpret_cast<uint4*>(A_ + st_off + i_ * 16 * WG_ROWB) = RA[i_];                                    \
            *reinterpret_cast<uint4*>(A_ + WG_TILE_B + st_off + i_ * 16 * WG_ROWB) = RB[i_];                        \
        }                                                                                                           \
    }

    WG_ROWINFO(tlo);
    WG_LOAD(tlo, ra0, rb0);
    WG_STORE(0, ra0, rb0);
    if (tlo + 1 < thi) WG_LOAD(tlo + 1, ra1, rb1);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int tile = tlo; tile < thi; tile += 2) {
        // LDS[cur] = tile, set 1 = tile+1 (requested a step ago); request tile+2 into set 0
        if (tile + 2 < thi) WG_LOAD(tile + 2, ra0, rb0);
        wgrad_step<ES, FI, WT>(lds + cur * 2 * WG_TILE_B, lds + cur * 2 * WG_TILE_B + WG_TILE_B, lane, wm, wn, acc);
        if (tile + 1 < thi) WG_STORE(cur ^ 1, ra1, rb1);
        __syncthreads();
        cur ^= 1;
        if (tile + 1 >= thi) break;
        // LDS[cur] = tile+1, set 0 = tile+2; request tile+3 into set 1
        if (tile + 3 < thi) WG_LOAD(tile + 3, ra1, rb1);
        wgrad_step<ES, FI, WT>(lds + cur * 2 * WG_TILE_B, lds + cur * 2 * WG_TILE_B + WG_TILE_B, lane, wm, wn, acc);
        if (tile + 2 < thi) WG_STORE(cur ^ 1, ra0, rb0);
        __syncthreads();
        cur ^= 1;
    }
#undef WG_LOAD
#undef WG_ROWINFO
#undef WG_STORE

    if (do_bias) {       // threads t = ch + 16*r0 share chunk column ch: sum their partials through LDS, one atomic per channel
        float* sb = reinterpret_cast<float*>(lds);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CE; ++j) sb[t * CE + j] = bsum[j];
        __syncthreads();
        if (t < 16) {
#pragma unroll
            for (int j = 0; j < CE; ++j) {
                float v = 0.f;
                for (int rr = 0; rr < 16; ++rr) v += sb[(t + 16 * rr) * CE + j];
                const int n = n0 + t * CE + j;
                if (n < p.db_n) { if (p.slab) p.bslab[(long long)split * p.N + n] = v; else unsafeAtomicAdd(p.db + n, v); }
            }
        }
    }
    // D[row = n][col = k]: lane holds rows 4*(lane>>4)+r, column lane&15 of each 16x16 tile.  The tile goes through this wave's
    // LDS slice so that every atomic wave-instruction adds WT consecutive k of ONE filter row (256 contiguous bytes for bf16:
    // the shape the memory-side atomic units run at full rate; straight from the accumulators it was 4 x 64 B).
    const int lr = (lane >> 4) * 4, lc = lane & 15;
    constexpr int SP = WT + 1;                                   // padded row of the staging slice (floats)
    __syncthreads();                                             // every wave is done with the operand tiles
    float* S = reinterpret_cast<float*>(lds) + wave * (WT * SP);
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(16 * i + lr + r) * SP + 16 * j + lc] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);                          // this wave's LDS writes have landed (slice is wave-private)
    if (lane < WT) {
        const int kc = k0 + wn * WT + lane;
        if (kc < p.Ktot) {
#pragma unroll 4
            for (int nl = 0; nl < WT; ++nl) {
                const int n = n0 + wm * WT + nl;
                if (n < p.N) {
                    if (p.slab) p.slab[((long long)split * p.N + n) * p.Ktot + kc] = S[nl * SP + lane];
                    else unsafeAtomicAdd(p.dW + (long long)n * p.Ktot + kc, S[nl * SP + lane]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 256 x 256, LDS-DMA (bf16)
// The loop above is bound by the L2->LDS volume of its operand loads (65 FLOP/B).  This variant owns a 256 (n) x 256 (k) tile
// with 8 waves (4 x 2, wave tile 64 x 128): 64 KB per 64-pixel step for 8.4 MFLOP, half the volume per FLOP, and stages it by
// LDS-DMA (no staging registers beside 128 accumulator registers; the next step's 64 KB is in flight while this one multiplies).
// LDS image: [64 px][512 B] per operand, UNPADDED (a DMA piece is 1 KiB = 2 rows, lane-linear); the transposed reads stay
// conflict-free through an XOR on the 32-byte granule index with key(row) = (row & 3) | ((row >> 3) & 1) << 2: the 32 lanes
// of a half-wave read 8 rows (q = 0..3, two values of g) x one granule each, 8 distinct keys -> 8 x 8 = 64 banks.  The key
// of the rows a lane stages does not change from piece to piece, so each lane still has ONE source chunk (tap, channel offset).
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ i32x4 wd_make_srd(const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void wd_dma16(const i32x4& srd, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd)
                 : "memory");
}

// NWN x NWK waves, wave tile 64 (n) x 16*FJ (k); square tiles only: TS = 64*NWN = 16*FJ*NWK channels per side.
//   <4, 2, 8>: 256 x 256, 512 threads, 128 KB of LDS  (many-pixel layers)
//   <2, 2, 4>: 128 x 128, 256 threads,  64 KB of LDS  (two workgroups per CU)
template <int NWN, int NWK, int FJ>
__global__ __launch_bounds__(64 * NWN * NWK, 2) void conv_wgrad_dma_kernel(const WParams p) {
    constexpr int NW = NWN * NWK;
    constexpr int TS = 64 * NWN;
    static_assert(TS == 16 * FJ * NWK, "square output tile");
    constexpr int WD_ROWB = TS * 2;               // bytes per pixel row of an operand tile
    constexpr int WD_TILE_B = 64 * WD_ROWB;
    constexpr int CPR = WD_ROWB / 16;             // 16-byte chunks per row
    constexpr int RPP = 1024 / WD_ROWB;           // rows per 1-KiB staging piece
    static_assert(RPP * NW == 16 && 64 / (RPP * NW) == 4, "four pieces per wave and operand, 16 rows apart");
    __shared__ __attribute__((aligned(16))) char lds[4 * WD_TILE_B];   // [stage][dY | X]

    int otile, split;
    if (p.xcd_map) {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int sl = j / p.out_tiles;
        otile = j - sl * p.out_tiles;
        split = sl * 8 + xcd;
    } else {
        otile = blockIdx.x;
        split = blockIdx.y;
    }
    const int tile_n = otile / p.ntiles_k;
    const int tile_k = otile - tile_n * p.ntiles_k;
    const int n0 = tile_n * TS, k0 = tile_k * TS;
    const int tlo = split * p.tiles_per_split;
    int thi = tlo + p.tiles_per_split;
    thi = thi < p.total_tiles ? thi : p.total_tiles;
    if (tlo >= thi) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // ---- staging role: piece i of this wave = rows RPP*(wave + NW*i) + lane / CPR, LDS chunk position lane % CPR
    const int rsub = lane / CPR;
    const int srow0 = RPP * wave + rsub;                       // + 16 i
    const int skey = (srow0 & 3) | (((srow0 >> 3) & 1) << 2);  // the same for every i (+16 keeps row & 3 and bit 3)
    const int sc = (lane % CPR) ^ (skey << 1);                 // source chunk (8 elements) this lane fetches
    const int kk = k0 + sc * 8;
    const bool kvalid = kk < p.Ktot;
    const int kpos = kk >> p.cshift;
    const int coff = kk & p.crun_mask;
    const int kh = (kpos * p.kw_inv) >> 16;
    const int kw = kpos - kh * p.KW;
    const int nn = n0 + sc * 8;
    const bool nvalid = nn < p.N;
    const unsigned dyo = (unsigned)(nn * 2);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    const int wm = wave / NWK, wn = wave % NWK;
    f32x4 acc[4][FJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.db != nullptr && tile_k == 0 && wn == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    uint4 ri[4];
#define WD_ROWINFO(TILE)                                                                                            \
    {                                                                                                               \
        const int tt_ = (TILE) < thi ? (TILE) : thi - 1;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) ri[i_] = p.rowinfo[(long long)tt_ * 64 + srow0 + 16 * i_]; \
    }
#define WD_STAGE(TILE, STG)                                                                                         \
    {                                                                                                               \
        int gi_ = 0;                                                                                                \
        _Pragma("unroll") for (int i_ = 1; i_ < RTN_MAX_GROUPS; ++i_)                                               \
            if (i_ < p.ngroups && (TILE) >= p.g[i_].tile_begin) gi_ = i_;                                           \
        const WGroup& G_ = p.g[gi_];                                                                                \
        const i32x4 xs_ = wd_make_srd(G_.x, G_.x_bytes);                                                            \
        const i32x4 ys_ = wd_make_srd(G_.dy, G_.dy_bytes);                                                          \
        const unsigned delta_ = (unsigned)(kh * G_.x_row_stride_b + kw * p.pix_stride_b + coff * 2);                \
        const int Hin_ = G_.Hin, Win_ = G_.Win;                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                          \
            const int iy_ = (int)(short)(ri[i_].z & 0xffffu) + kh, ix_ = (int)(short)(ri[i_].z >> 16) + kw;         \
            const bool ok_ = kvalid && (unsigned)iy_ < (unsigned)Hin_ && (unsigned)ix_ < (unsigned)Win_;            \
            const unsigned piece_ = (unsigned)((wave + NW * i_) * 1024);                                             \
            wd_dma16(ys_, (nvalid && ri[i_].y != OOB_OFFSET) ? ri[i_].y + dyo : OOB_OFFSET,                         \
                     lds_base + (unsigned)((STG) * 2 * WD_TILE_B) + piece_);                                        \
            wd_dma16(xs_, ok_ ? ri[i_].x + delta_ : OOB_OFFSET,                                                     \
                     lds_base + (unsigned)((STG) * 2 * WD_TILE_B + WD_TILE_B) + piece_);                            \
        }                                                                                                           \
    }

    WD_ROWINFO(tlo);
    WD_STAGE(tlo, 0);
    WD_ROWINFO(tlo + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int rkey = q | ((g & 1) << 2);
    int cur = 0;
#pragma unroll 1
    for (int tile = tlo; tile < thi; ++tile) {
        const char* A = lds + cur * 2 * WD_TILE_B;     // dY tile
        const char* B = A + WD_TILE_B;                 // X tile
        // the tile's first A fragments are requested before the next tile's staging is issued (as in the forward kernels)
        s16x8 af0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ca = (wm * 64 + 16 * i + 4 * pp) * 2;
            const int aoff = (8 * g + q) * WD_ROWB + (((ca >> 5) ^ rkey) << 5) + (ca & 31);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(A + aoff));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(A + aoff + 4 * WD_ROWB));
            af0[i] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        if (tile + 1 < thi) {
            WD_STAGE(tile + 1, cur ^ 1);
            WD_ROWINFO(tile + 2);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = 32 * s + 8 * g + q;
            s16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (s == 0) { af[i] = af0[i]; continue; }
                const int ca = (wm * 64 + 16 * i + 4 * pp) * 2;            // byte column inside the 512-byte row
                const int aoff = row * WD_ROWB + (((ca >> 5) ^ rkey) << 5) + (ca & 31);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(A + aoff));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(A + aoff + 4 * WD_ROWB));
                af[i] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (do_bias) {                              // BiasAddGrad: the A fragment of lane (m, kq) holds 8 pixels of channel m
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[i] += __uint_as_float(((unsigned)(unsigned short)af[i][e]) << 16);
            }
#pragma unroll
            for (int jh = 0; jh < FJ; jh += 4) {
                s16x8 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cb = (wn * 16 * FJ + 16 * (jh + j) + 4 * pp) * 2;
                    const int boff = row * WD_ROWB + (((cb >> 5) ^ rkey) << 5) + (cb & 31);
                    const s16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(B + boff));
                    const s16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(B + boff + 4 * WD_ROWB));
                    bf[j] = (s16x8){lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                                 __builtin_bit_cast(bf16x8, bf[j]), acc[i][jh + j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next stage (and the row info behind it) has landed
        __syncthreads();
        cur ^= 1;
    }
#undef WD_STAGE
#undef WD_ROWINFO

    const int lr = (lane >> 4) * 4, lc = lane & 15;
    if (do_bias) {       // lane (m = lane & 15, kq = lane >> 4): fold the four pixel groups, lanes 0..15 add channel m
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = n0 + wm * 64 + 16 * i + lc;
            if (lane < 16 && n < p.db_n) { if (p.slab) p.bslab[(long long)split * p.N + n] = v; else unsafeAtomicAdd(p.db + n, v); }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int kc = k0 + wn * 16 * FJ + 16 * j + lc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * 64 + 16 * i + lr + r;
                if (n < p.N && kc < p.Ktot) {
                    if (p.slab) p.slab[((long long)split * p.N + n) * p.Ktot + kc] = acc[i][j][r];
                    else unsafeAtomicAdd(p.dW + (long long)n * p.Ktot + kc, acc[i][j][r]);
                }
            }
        }
}

// (A ring-of-four variant of the 256 x 256 kernel - 32-pixel stages, one counted wait per iteration, row info through scalar loads -
// was built in round 3, bit-identical and 1-16 % slower on every layer (profiles/r3_wgrad_ring_ab.txt, r3_wgrad_ring128_ab.txt: the loop
// is bound by instruction issue, not by the latency its vmcnt(0) exposes); removed in round 4.)

// ------------------------------------------------------------------------------------------------ ordered reduction of the pixel splits
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

// ------------------------------------------------------------------------------------------------ small kernels
template <int ES>
__global__ __launch_bounds__(256) void bias_grad_kernel(const char* __restrict__ dy, long long rows, int N, long long ld_b,
                                                        float* __restrict__ db) {
    // thread = (16-byte chunk column c, row lane r); rows are strided over row lanes and blocks; the row lanes of a block are
    // summed through LDS so that each block issues ONE atomic per channel (same-address atomics serialise).
    constexpr int CE = 16 / ES;
    __shared__ float sh[256 * CE];
    const int nch = (N + CE - 1) / CE;
    const int RP = 256 / nch;                          // row lanes per block (host guarantees nch <= 256)
    const int c = threadIdx.x % nch, r = threadIdx.x / nch;
    float s[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) s[j] = 0.f;
    if (r < RP) {
        const long long step = (long long)gridDim.x * RP;
        for (long long row = (long long)blockIdx.x * RP + r; row < rows; row += 8 * step) {
            uint4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                 // 8 independent row loads in flight (rows past the end: zeros)
                const long long rr = row + u * step;
                q[u] = rr < rows ? *reinterpret_cast<const uint4*>(dy + rr * ld_b + (long long)c * 16) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned w4[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                if constexpr (ES == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s[2 * j] += __uint_as_float(w4[j] << 16);
                        s[2 * j + 1] += __uint_as_float(w4[j] & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += __uint_as_float(w4[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CE; ++j) sh[threadIdx.x * CE + j] = s[j];
    __syncthreads();
    if (threadIdx.x < nch) {
#pragma unroll
        for (int j = 0; j < CE; ++j) {
            float v = 0.f;
            for (int rr = 0; rr < RP; ++rr) v += sh[(threadIdx.x + rr * nch) * CE + j];
            if (threadIdx.x * CE + j < N) unsafeAtomicAdd(db + threadIdx.x * CE + j, v);
        }
    }
}

// out[b][2*oy][2*ox][:] = in[b][oy][ox][:], zeros elsewhere; out is [B][Hu][Wu][C]
__global__ __launch_bounds__(256) void zero_insert2_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int B, int H, int W,
                                                           int Hu, int Wu, int cv) {
    const long long total = (long long)B * Hu * Wu * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int x = (int)(r % Wu);
        r /= Wu;
        const int y = (int)(r % Hu);
        const int b = (int)(r / Hu);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(x & 1) && !(y & 1) && (y >> 1) < H && (x >> 1) < W) v = in[(((long long)b * H + (y >> 1)) * W + (x >> 1)) * cv + c];
        out[i] = v;
    }
}

// d_src[b][sy][sx][:] (+)= sum over destination pixels (y,x) with floor(y*rh)==sy (clamped) && floor(x*rw)==sx of d_dst
template <int ES>
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const char* __restrict__ d_dst, char* __restrict__ d_src, int B,
                                                               int Hd, int Wd, int Hs, int Ws, int C, float rh, float rw, int accumulate) {
    constexpr int CE = 16 / ES;
    const int cv = C / CE;
    const long long total = (long long)B * Hs * Ws * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int sx = (int)(r % Ws);
        r /= Ws;
        const int sy = (int)(r % Hs);
        const int b = (int)(r / Hs);
        float s[CE];
#pragma unroll
        for (int j = 0; j < CE; ++j) s[j] = 0.f;
        // candidate destination rows: the preimage of sy under y -> min(floor(y*rh), Hs-1) is a short contiguous run
        int y0 = (int)floorf((float)sy / rh) - 1; if (y0 < 0) y0 = 0;
        int x0 = (int)floorf((float)sx / rw) - 1; if (x0 < 0) x0 = 0;
        for (int y = y0; y < Hd; ++y) {
            int my = (int)floorf((float)y * rh); my = my < Hs - 1 ? my : Hs - 1;
            if (my < sy) continue;
            if (my > sy) break;
            for (int x = x0; x < Wd; ++x) {
                int mx = (int)floorf((float)x * rw); mx = mx < Ws - 1 ? mx : Ws - 1;
                if (mx < sx) continue;
                if (mx > sx) break;
                const uint4 q = *reinterpret_cast<const uint4*>(d_dst + ((((long long)b * Hd + y) * Wd + x) * C + c * CE) * ES);
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
                if constexpr (ES == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s[2 * j] += __uint_as_float(w4[j] << 16); s[2 * j + 1] += __uint_as_float(w4[j] & 0xffff0000u); }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += __uint_as_float(w4[j]);
                }
            }
        }
        char* op = d_src + i * 16;
        if (accumulate) {
            const uint4 q = *reinterpret_cast<const uint4*>(op);
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
            if constexpr (ES == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[2 * j] += __uint_as_float(w4[j] << 16); s[2 * j + 1] += __uint_as_float(w4[j] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += __uint_as_float(w4[j]);
            }
        }
        uint4 o;
        if constexpr (ES == 2) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { bf16x2 v = {(__bf16)s[2 * j], (__bf16)s[2 * j + 1]}; w[j] = __builtin_bit_cast(unsigned, v); }
            o = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            o = make_uint4(__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3]));
        }
        *reinterpret_cast<uint4*>(op) = o;
    }
}

// Training-mode max-pool: also records which of the 9 window taps (kh*3+kw, first maximum in scan order — TF MaxPoolGrad's
// choice) won, one byte per output element, so that the backward pass is a gather with no atomics.
template <int ES>
__global__ __launch_bounds__(256) void maxpool_fwd_idx_kernel(const char* __restrict__ in, char* __restrict__ out,
                                                              unsigned char* __restrict__ idx, int B, int Hin, int Win, int C, int Hout,
                                                              int Wout, int pad_t, int pad_l) {
    constexpr int CE = 16 / ES;
    const int cv = C / CE;
    const long long total = (long long)B * Hout * Wout * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv);
        long long r = i / cv;
        const int ox = (int)(r % Wout);
        r /= Wout;
        const int oy = (int)(r % Hout);
        const int b = (int)(r / Hout);
        float mx[CE];
        unsigned char am[CE];
#pragma unroll
        for (int j = 0; j < CE; ++j) { mx[j] = -INFINITY; am[j] = 255; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy * 2 - pad_t + kh;
            if ((unsigned)iy >= (unsigned)Hin) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = ox * 2 - pad_l + kw;
                if ((unsigned)ix >= (unsigned)Win) continue;
                const uint4 q = *reinterpret_cast<const uint4*>(in + ((((long long)b * Hin + iy) * Win + ix) * C + cc * CE) * ES);
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
                float v[CE];
                if constexpr (ES == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w4[j] << 16); v[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u); }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(w4[j]);
                }
#pragma unroll
                for (int j = 0; j < CE; ++j)
                    if (v[j] > mx[j]) { mx[j] = v[j]; am[j] = (unsigned char)(kh * 3 + kw); }
            }
        }
        uint4 o;
        if constexpr (ES == 2) {
            o.x = (__float_as_uint(mx[0]) >> 16) | (__float_as_uint(mx[1]) & 0xffff0000u);
            o.y = (__float_as_uint(mx[2]) >> 16) | (__float_as_uint(mx[3]) & 0xffff0000u);
            o.z = (__float_as_uint(mx[4]) >> 16) | (__float_as_uint(mx[5]) & 0xffff0000u);
            o.w = (__float_as_uint(mx[6]) >> 16) | (__float_as_uint(mx[7]) & 0xffff0000u);
        } else {
            o = make_uint4(__float_as_uint(mx[0]), __float_as_uint(mx[1]), __float_as_uint(mx[2]), __float_as_uint(mx[3]));
        }
        *reinterpret_cast<uint4*>(out + i * 16) = o;
#pragma unroll
        for (int j = 0; j < CE; ++j) idx[i * CE + j] = am[j];
    }
}

// dx[b][iy][ix][c] = sum over the (at most 4) windows containing (iy,ix) whose recorded winner is this tap of dy,
// zeroed where x <= 0 when relu_mask (x is a ReLU output).
template <int ES>
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const char* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                              const char* __restrict__ x, char* __restrict__ dx, int B, int Hin, int Win,
                                                              int C, int Hout, int Wout, int pad_t, int pad_l, int relu_mask) {
    constexpr int CE = 16 / ES;
    const int cv = C / CE;
    // grid: x = 16-byte items of an image row, y = image rows of the batch (strided).  One 32-bit division per item: the three 64-bit
    // ones of the flat index were a third of the kernel's time (0.32 -> 0.2x ms at batch 16, last kernel but one of the backward pass).
    const unsigned jrow = blockIdx.x * blockDim.x + threadIdx.x;
    if (jrow >= (unsigned)(Win * cv)) return;
    const int ix = (int)(jrow / (unsigned)cv), cc = (int)(jrow - (unsigned)ix * (unsigned)cv);
    for (int row = blockIdx.y; row < B * Hin; row += gridDim.y) {
        const int b = row / Hin, iy = row - b * Hin;
        const long long i = (long long)row * Win * cv + jrow;
        float s[CE];
#pragma unroll
        for (int j = 0; j < CE; ++j) s[j] = 0.f;
        // windows oy with oy*2 - pad_t <= iy <= oy*2 - pad_t + 2
        const int ty = iy + pad_t, tx = ix + pad_l;
        for (int oy = (ty - 2 + 1) >> 1; oy <= (ty >> 1); ++oy) {
            if (oy < 0 || oy >= Hout) continue;
            const int kh = ty - 2 * oy;
            for (int ox = (tx - 2 + 1) >> 1; ox <= (tx >> 1); ++ox) {
                if (ox < 0 || ox >= Wout) continue;
                const int kw = tx - 2 * ox;
                const unsigned char tap = (unsigned char)(kh * 3 + kw);
                const long long o = (((long long)b * Hout + oy) * Wout + ox) * cv + cc;
                const uint4 gq = *reinterpret_cast<const uint4*>(dy + o * 16);
                const unsigned g4[4] = {gq.x, gq.y, gq.z, gq.w};
                const unsigned char* ip = idx + o * CE;
                // relu_mask == 2: x is the POOLED tensor: the winner's value IS the window's maximum, so "x > 0 at the winner" is
                // "pooled > 0" (the fused stem kernel never writes the pre-pool tensor)
                uint4 yq = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
                if (relu_mask == 2) yq = *reinterpret_cast<const uint4*>(x + o * 16);
                const unsigned y4[4] = {yq.x, yq.y, yq.z, yq.w};
                if constexpr (ES == 2) {
                    const uint2 iq = *reinterpret_cast<const uint2*>(ip);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned char a = (unsigned char)(((j < 4 ? iq.x : iq.y) >> (8 * (j & 3))) & 0xffu);
                        const float g = (j & 1) ? __uint_as_float(g4[j >> 1] & 0xffff0000u) : __uint_as_float(g4[j >> 1] << 16);
                        const float yv = (j & 1) ? __uint_as_float(y4[j >> 1] & 0xffff0000u) : __uint_as_float(y4[j >> 1] << 16);
                        if (a == tap && yv > 0.f) s[j] += g;
                    }
                } else {
                    const unsigned iq = *reinterpret_cast<const unsigned*>(ip);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((unsigned char)((iq >> (8 * j)) & 0xffu) == tap && (relu_mask != 2 || __uint_as_float(y4[j]) > 0.f)) s[j] += __uint_as_float(g4[j]);
                }
            }
        }
        if (relu_mask == 1) {
            const uint4 xq = *reinterpret_cast<const uint4*>(x + i * 16);
            const unsigned x4[4] = {xq.x, xq.y, xq.z, xq.w};
            if constexpr (ES == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = (j & 1) ? __uint_as_float(x4[j >> 1] & 0xffff0000u) : __uint_as_float(x4[j >> 1] << 16);
                    if (!(xv > 0.f)) s[j] = 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!(__uint_as_float(x4[j]) > 0.f)) s[j] = 0.f;
            }
        }
        uint4 o;
        if constexpr (ES == 2) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { bf16x2 v = {(__bf16)s[2 * j], (__bf16)s[2 * j + 1]}; w[j] = __builtin_bit_cast(unsigned, v); }
            o = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            o = make_uint4(__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3]));
        }
        *reinterpret_cast<uint4*>(dx + i * 16) = o;
    }
}

// gradient goes to the first window position (row-major scan) holding the maximum — TF MaxPoolGrad's choice
template <int ES>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const char* __restrict__ x, const char* __restrict__ dy, float* __restrict__ dx32,
                                                          int B, int Hin, int Win, int C, int Hout, int Wout, int pad_t, int pad_l) {
    constexpr int CE = 16 / ES;
    const int cv = C / CE;
    const long long total = (long long)B * Hout * Wout * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv);
        long long r = i / cv;
        const int ox = (int)(r % Wout);
        r /= Wout;
        const int oy = (int)(r % Hout);
        const int b = (int)(r / Hout);
        float mx[CE];
        int arg[CE];
#pragma unroll
        for (int j = 0; j < CE; ++j) { mx[j] = -INFINITY; arg[j] = -1; }
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy * 2 - pad_t + kh;
            if ((unsigned)iy >= (unsigned)Hin) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = ox * 2 - pad_l + kw;
                if ((unsigned)ix >= (unsigned)Win) continue;
                const int pix = iy * Win + ix;
                const uint4 q = *reinterpret_cast<const uint4*>(x + (((long long)b * Hin * Win + pix) * C + cc * CE) * ES);
                const unsigned w4[4] = {q.x, q.y, q.z, q.w};
                float v[CE];
                if constexpr (ES == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w4[j] << 16); v[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u); }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(w4[j]);
                }
#pragma unroll
                for (int j = 0; j < CE; ++j)
                    if (v[j] > mx[j]) { mx[j] = v[j]; arg[j] = pix; }
            }
        }
        const uint4 gq = *reinterpret_cast<const uint4*>(dy + i * 16);
        const unsigned g4[4] = {gq.x, gq.y, gq.z, gq.w};
        float g[CE];
        if constexpr (ES == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[2 * j] = __uint_as_float(g4[j] << 16); g[2 * j + 1] = __uint_as_float(g4[j] & 0xffff0000u); }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = __uint_as_float(g4[j]);
        }
#pragma unroll
        for (int j = 0; j < CE; ++j)
            if (arg[j] >= 0 && g[j] != 0.f)
                unsafeAtomicAdd(dx32 + ((long long)b * Hin * Win + arg[j]) * C + cc * CE + j, g[j]);
    }
}

template <int ES>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ in, char* __restrict__ out, long long n,
                                                            const char* __restrict__ relu_src) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float val = in[i];
        if (relu_src) {
            float xv;
            if constexpr (ES == 2) xv = __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(relu_src)[i]) << 16);
            else xv = reinterpret_cast<const float*>(relu_src)[i];
            if (!(xv > 0.f)) val = 0.f;
        }
        if constexpr (ES == 2) {
            const __bf16 hb = (__bf16)val;
            reinterpret_cast<unsigned short*>(out)[i] = __builtin_bit_cast(unsigned short, hb);
        } else {
            reinterpret_cast<float*>(out)[i] = val;
        }
    }
}

// out[r][0..cout) = cast(in[r][0..cin)), zero for c >= cin   (f32 loss gradients -> 64-channel dY of the head outputs)
template <int ES>
__global__ __launch_bounds__(256) void pad_cast_rows_kernel(const float* __restrict__ in, char* __restrict__ out, long long rows,
                                                            int cin, int cout) {
    const long long total = rows * cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cout;
        const int c = (int)(i - r * cout);
        const float v = c < cin ? in[r * cin + c] : 0.f;
        if constexpr (ES == 2) {
            const __bf16 hb = (__bf16)v;
            reinterpret_cast<unsigned short*>(out)[i] = __builtin_bit_cast(unsigned short, hb);
        } else {
            reinterpret_cast<float*>(out)[i] = v;
        }
    }
}

// bf16, cout a multiple of 8, 16-byte aligned out: one thread = 8 consecutive columns of a row (one 16-byte store instead of eight
// 2-byte ones; 32-bit index arithmetic)
__global__ __launch_bounds__(256) void pad_cast_rows8_kernel(const float* __restrict__ in, uint4* __restrict__ out, unsigned items,
                                                             int cin, int cg) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
        const unsigned r = i / (unsigned)cg;
        const int c0 = (int)(i - r * (unsigned)cg) * 8;
        const float* src = in + (long long)r * cin;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = c0 + 2 * j < cin ? src[c0 + 2 * j] : 0.f, b = c0 + 2 * j + 1 < cin ? src[c0 + 2 * j + 1] : 0.f;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            const bf16x2 v = {(__bf16)a, (__bf16)b};
            w[j] = __builtin_bit_cast(unsigned, v);
        }
        out[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over i of (g[i]*scale[i])^2 -> partial per block (fixed order), finished by sumsq_final
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, const float* __restrict__ scale, long long n,
                                                    double* __restrict__ partial) {
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i] * (scale ? scale[i] : 1.f);
        s += (double)v * (double)v;
    }
    __shared__ double sh[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ partial, int nb, double* __restrict__ out) {
    __shared__ double sh[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) v += partial[i];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// Keras-2 Adam (RetinaNet.py:130: lr, beta 0.9/0.999, epsilon 1e-7, no decay) with global-norm clipping:
//   g = grad * scale * min(1, clipnorm / norm);  lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  w -= lr_t * m / (sqrt(v) + eps)
// and re-emission of the forward weights  w_fwd = cast(w * fold)  (fold = frozen-BN scale per output channel, 1 elsewhere).
template <int ES>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, const float* __restrict__ gscale,
                                                   const float* __restrict__ fold, char* __restrict__ w_fwd, long long n, float lr_t,
                                                   float b1, float b2, float eps, const double* __restrict__ sumsq, float clipnorm,
                                                   float grad_mul) {
    float clip = 1.f;
    if (clipnorm > 0.f && sumsq) {
        const float norm = sqrtf((float)sumsq[0]) * fabsf(grad_mul);
        if (norm > clipnorm) clip = clipnorm / norm;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * (gscale ? gscale[i] : 1.f) * grad_mul * clip;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float wi = w[i] - lr_t * mi / (sqrtf(vi) + eps);
        m[i] = mi; v[i] = vi; w[i] = wi;
        if (w_fwd) {
            const float wf = wi * (fold ? fold[i] : 1.f);
            if constexpr (ES == 2) {
                const __bf16 hb = (__bf16)wf;
                reinterpret_cast<unsigned short*>(w_fwd)[i] = __builtin_bit_cast(unsigned short, hb);
            } else {
                reinterpret_cast<float*>(w_fwd)[i] = wf;
            }
        }
    }
}

// Per-tensor clipping (tf.keras / Keras >= 2.4 semantics of Adam(clipnorm=c)): every gradient tensor t is scaled by
// c / max(||g_t||, c).  The tensors are contiguous segments [seg_begin[s], seg_begin[s + 1]) of the flat parameter vector.
__global__ __launch_bounds__(256) void sumsq_segments_kernel(const float* __restrict__ g, const float* __restrict__ scale,
                                                             const long long* __restrict__ seg_begin, double* __restrict__ out) {
    const long long lo = seg_begin[blockIdx.x], hi = seg_begin[blockIdx.x + 1];
    double s = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {               // fixed order: the same bits on every run
        const float v = g[i] * (scale ? scale[i] : 1.f);
        s += (double)v * (double)v;
    }
    __shared__ double sh[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

constexpr int ADAM_MAX_SEGS = 2048;
template <int ES>
__global__ __launch_bounds__(256) void adam_segments_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ g, const float* __restrict__ gscale,
                                                            const float* __restrict__ fold, char* __restrict__ w_fwd, long long n, float lr_t,
                                                            float b1, float b2, float eps, const long long* __restrict__ seg_begin, int nseg,
                                                            const double* __restrict__ sumsq_seg, long long elem_offset, float clipnorm,
                                                            float grad_mul) {
    __shared__ long long sb[ADAM_MAX_SEGS + 1];
    __shared__ float sclip[ADAM_MAX_SEGS];
    for (int i = threadIdx.x; i <= nseg; i += 256) sb[i] = seg_begin[i];
    for (int i = threadIdx.x; i < nseg; i += 256) {
        const float norm = sqrtf((float)sumsq_seg[i]) * fabsf(grad_mul);
        sclip[i] = (clipnorm > 0.f && norm > clipnorm) ? clipnorm / norm : 1.f;
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long gi_ = i + elem_offset;
        int lo = 0, hi = nseg;                          // last segment whose begin <= gi_
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sb[mid] <= gi_) lo = mid; else hi = mid;
        }
        const float gi = g[i] * (gscale ? gscale[i] : 1.f) * grad_mul * sclip[lo];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float wi = w[i] - lr_t * mi / (sqrtf(vi) + eps);
        m[i] = mi; v[i] = vi; w[i] = wi;
        if (w_fwd) {
            const float wf = wi * (fold ? fold[i] : 1.f);
            if constexpr (ES == 2) {
                const __bf16 hb = (__bf16)wf;
                reinterpret_cast<unsigned short*>(w_fwd)[i] = __builtin_bit_cast(unsigned short, hb);
            } else {
                reinterpret_cast<float*>(w_fwd)[i] = wf;
            }
        }
    }
}

inline unsigned grid_for(long long work, int cap = 4096) {
    long long g = (work + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace

// Which general kernel runs a layer and how its pixels are split (shared by the launcher and the workspace size; sized for the 256 CUs
// of an MI355X when there is no handle).
struct WgradPlan {
    bool dma, dma_small, xcd_map;
    int CH;
    long long tiles, out_tiles, nsplit, nsplit_used;
    int tiles_per_split;
};
static WgradPlan wgrad_plan(const rtn_conv_desc_t* d, int cus) {
    WgradPlan w;
    const int es = rtn_dtype_size(d->dtype);
    const long long Ktot = (long long)d->KH * d->KW * d->Crun;
    w.tiles = 0;
    for (int i = 0; i < d->ngroups; ++i) w.tiles += ((long long)d->g[i].Hout * d->g[i].Wout * d->batch + 63) / 64;
    // bf16 layers with more than 128 filters and at least one 256-wide K tile: the 256 x 256 LDS-DMA kernel (RTN_WGRAD_DMA=0: off)
    // Measured per layer (tools/profile_train.py, batch 16): -27 % on the head layers (5575 pixel tiles, 0.71 -> 0.52 ms) and
    // -21 % on P3, but +25..40 % on the layers with ~1000 pixel tiles or fewer (res4/res5, C4/C5, P5, P6: short loops, and each
    // workgroup ends with 128 accumulator registers of output), hence the pixel-count condition.
    w.dma = es == 2 && d->N > 128 && Ktot >= 256 && w.tiles >= 2048;
    const int dma_env = rtn_env_int("RTN_WGRAD_DMA", -1);       // 0: never; 2: wherever the shape allows (tests, A/B); -1: unset
    if (dma_env >= 0) w.dma = dma_env == 2 ? (es == 2 && d->N > 128 && Ktot >= 256) : (w.dma && dma_env != 0);
    // the same staging on the 128 x 128 tile for the other bf16 layers with whole 128-wide tiles (RTN_WGRAD_DMA_SMALL=0: register-staged kernel)
    w.dma_small = !w.dma && es == 2;                       // measured: train step 37.4 -> 36.8 ms against the register-staged kernel
    if (rtn_env_int("RTN_WGRAD_DMA_SMALL", 1) == 0 || dma_env == 0) w.dma_small = false;
    w.CH = w.dma ? 256 : (es == 2 ? 128 : 64);
    const int ntn = (d->N + w.CH - 1) / w.CH;
    const long long ntk = (Ktot + w.CH - 1) / w.CH;
    w.out_tiles = (long long)ntn * ntk;
    // Workgroup count: whole rounds of the resident slots (two 256-thread workgroups per CU), rounded DOWN - 1044 workgroups
    // on 1024 slots cost a third round (training step 40.8 -> 38.8 ms).  RTN_WGRAD_BLOCKS overrides the target.
    // (A 256 x 256-tile, 512-thread variant of the kernel was built and measured: no faster at equal rounds - the loop is
    // bound by the latency of its register-staged loads, not by MFMA work per byte.)
    const long long slots = (long long)(cus > 0 ? cus : 256) * (w.dma ? 1 : 2);
    // measured (RTN_WGRAD_BLOCKS sweep): one round for both LDS-DMA kernels.  Round 3, 256 x 256 kernel with the slab reduction
    // (profiles/r3_wgrad_blocks_ab.txt): 256 workgroups 0.268 ms against 0.282 with 512 on a head-tower layer, 0.209 / 0.223 on P3 -
    // every split writes a 2.4 MB slab that the finish kernel reads back (PMC: 132 MB written per layer with 56 splits)
    long long target = slots * ((w.dma_small || w.dma) ? 1 : 2);
    { const long long v = rtn_env_int("RTN_WGRAD_BLOCKS", 0); if (v >= 64 && v <= 65536) target = v; }
    long long nsplit = target / w.out_tiles;
    if (nsplit > w.tiles) nsplit = w.tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    w.xcd_map = nsplit >= 8;
    if (w.xcd_map) nsplit &= ~7ll;
    w.tiles_per_split = (int)((w.tiles + nsplit - 1) / nsplit);
    w.nsplit_used = (w.tiles + w.tiles_per_split - 1) / w.tiles_per_split;      // splits that own at least one pixel tile
    w.nsplit = w.nsplit_used;
    if (w.xcd_map && (w.nsplit & 7)) w.nsplit = (w.nsplit + 7) & ~7ll;          // the XCD map wants a multiple of 8: the extra splits are empty
    return w;
}

// (rtn_wgrad_halo.hip, the one-kernel-row-per-tile 3x3 kernel of round 2, lost every layer of the benchmark shapes to the window
// kernel below and left the library in round 4: profiles/r2_v3_ab_wgrad_halo.txt, r3_wgrad_win_ab.txt.)

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

// rtn_wgrad_win.hip (all nine taps per output tile over a sliding window of the input; stride-1 3x3 layers with whole blocks of 128
// filters and 64 channels, image rows of up to 254 pixels) against the kernels above, measured per shape at batch 8 (tools/ab_wgrad.py,
// profiles/r3_wgrad_win_ab.txt): head towers 0.197 vs 0.285 ms, P3 0.157 vs 0.216, res4 branch2b / P4 0.068 vs 0.076, res3 branch2b
// 0.065 vs 0.068, res5 branch2b 0.071 vs 0.087, but P5 (8,400 pixels x 8 output tiles = 64 workgroups) 0.048 vs 0.042: it takes the
// layers with enough work for a chip-wide grid, 64-pixel tiles x output tiles >= 3000 (res5 4,192, res4 4,200, P5 1,048).
// RTN_WGRAD_WIN=1: wherever the shape allows (tests, A/B), 0: never.
static bool wgrad_takes_win(const rtn_conv_desc_t* d, const WgradPlan& w) {
    const int knob = rtn_env_int("RTN_WGRAD_WIN", -1);
    if (knob == 0 || rtn_wgrad_win_workspace_bytes(d) == 0) return false;
    if (knob > 0) return true;
    // (the 64-filter form, res2 branch2b and the head outputs: half the work per tile)
    return w.tiles * (long long)(d->N == 64 ? 1 : 2 * (d->N / 128)) * (d->Crun / 64) >= 6000;
}

// workspace = the row-info table, then (unless RTN_WGRAD_SLAB=0) the per-split slabs of the ordered reduction; the nine-tap window
// kernel (rtn_wgrad_win.hip) uses the same bytes for its own slabs
extern "C" size_t rtn_conv2d_wgrad_workspace_bytes(const rtn_conv_desc_t* d) {
    rtn_env_sync();
    if (!d || d->ngroups < 1 || d->ngroups > RTN_MAX_GROUPS || d->N < 1 || d->Crun < 1 || d->KH < 1 || d->KW < 1) return 0;
    if (d->dtype != RTN_BF16 && d->dtype != RTN_F32) return 0;
    const WgradPlan w = wgrad_plan(d, 256);
    const size_t table = ((size_t)w.tiles * 64 * sizeof(uint4) + 255) & ~(size_t)255;
    size_t slabs = 0;
    if (rtn_env_int("RTN_WGRAD_SLAB", 1) != 0)
        slabs = (size_t)w.nsplit_used * (size_t)d->N * ((size_t)d->KH * d->KW * d->Crun + 1) * sizeof(float);
    // the window kernel's slabs also sit BEHIND the table (16 B per pixel): a table built once by rtn_conv2d_wgrad_rowinfo stays valid
    // whichever kernel later prepared calls select
    const size_t win = wgrad_takes_win(d, w) ? rtn_wgrad_win_workspace_bytes(d) : 0;
    return table + (slabs > win ? slabs : win);
}

static int wgrad_launch(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, void* workspace, size_t workspace_bytes,
                        int mode = 0 /* 0: build the row-info table and run; 1: build it only; 2: run on a table built earlier */);

extern "C" int rtn_debug_last_wgrad_impl(rtn_handle_t h) { return h ? h->last_wgrad_impl : RTN_EINVAL; }

extern "C" int rtn_conv2d_wgrad(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, void* workspace, size_t workspace_bytes) {
    return wgrad_launch(h, d, dW, nullptr, 0, workspace, workspace_bytes);
}

/* The per-pixel row-info table alone (workspace), and the weight gradient on a table built earlier for the SAME descriptor. */
extern "C" int rtn_conv2d_wgrad_rowinfo(rtn_handle_t h, const rtn_conv_desc_t* d, void* workspace, size_t workspace_bytes) {
    return wgrad_launch(h, d, nullptr, nullptr, 0, workspace, workspace_bytes, 1);
}
extern "C" int rtn_conv2d_wgrad_prepared(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, const void* workspace,
                                         size_t workspace_bytes) {
    if (h && db && (db_n < 1 || (d && db_n > d->N))) return rtn_fail(h, RTN_EINVAL, "wgrad_prepared: bad db_n");
    return wgrad_launch(h, d, dW, db, db ? db_n : 0, const_cast<void*>(workspace), workspace_bytes, 2);
}

/* wgrad with the bias gradient fused: db[0..db_n) += column sums of dY (BiasAddGrad) */
extern "C" int rtn_conv2d_wgrad_bias(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, void* workspace,
                                     size_t workspace_bytes) {
    if (h && (!db || db_n < 1 || (d && db_n > d->N))) return rtn_fail(h, RTN_EINVAL, "wgrad_bias: bad db / db_n");
    return wgrad_launch(h, d, dW, db, db_n, workspace, workspace_bytes);
}

static int wgrad_launch(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, void* workspace, size_t workspace_bytes, int mode) {
    if (!h) return RTN_EINVAL;
    rtn_env_sync();
    if (!d || (!dW && mode != 1) || !workspace) return rtn_fail(h, RTN_EINVAL, "wgrad: null argument");
    if (d->dtype != RTN_BF16 && d->dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "wgrad: bad dtype");
    const int es = rtn_dtype_size(d->dtype);
    if (d->ngroups < 1 || d->ngroups > RTN_MAX_GROUPS || d->batch < 1 || d->N < 1) return rtn_fail(h, RTN_EINVAL, "wgrad: bad group/batch/N");
    const int cshift = ilog2_exact(d->Crun);
    if (cshift < 0 || (d->Crun * es) % 16) return rtn_fail(h, RTN_EINVAL, "wgrad: Crun %d", d->Crun);
    const long long Ktot = (long long)d->KH * d->KW * d->Crun;
    if ((Ktot * es) % 16 || Ktot > (1 << 24)) return rtn_fail(h, RTN_EINVAL, "wgrad: K %lld", Ktot);
    if ((d->N * es) % 16 || (d->out_ld * es) % 16) return rtn_fail(h, RTN_EINVAL, "wgrad: N %d / out_ld %d must span whole 16-byte chunks (pad dY)", d->N, d->out_ld);
    if (d->w_rows < d->N) return rtn_fail(h, RTN_EINVAL, "wgrad: w_rows < N");
    if (((uintptr_t)dW & 15) || ((uintptr_t)workspace & 15)) return rtn_fail(h, RTN_EINVAL, "wgrad: dW/workspace alignment");
    const long long pix_b = (long long)d->pix_stride * es;
    if (pix_b % 16 && (d->KW != 1 || (d->sx * pix_b) % 16 || (d->pad_l * pix_b) % 16)) return rtn_fail(h, RTN_EINVAL, "wgrad: unaligned taps");
    {   // required: the row-info table; the slabs of the ordered reduction behind it are used when there is room (else float atomics)
        long long t64 = 0;
        for (int i = 0; i < d->ngroups; ++i) t64 += ((long long)d->g[i].Hout * d->g[i].Wout * d->batch + 63) / 64;
        const size_t need = (size_t)t64 * 64 * sizeof(uint4);
        if (workspace_bytes < need) return rtn_fail(h, RTN_ENOMEM, "wgrad: workspace %zu < %zu", workspace_bytes, need);
    }

    // the stride-1 3x3 layers the nine-tap window kernel takes (wgrad_takes_win).  Mode 1 (table only) never stops here: the row-info
    // table is built for every layer, so that a later prepared call that selects the general kernels (a knob flipped in between, a
    // shape the window kernel turns down) never runs on an unprepared table.
    if (mode != 1 && wgrad_takes_win(d, wgrad_plan(d, h->num_cus > 0 ? h->num_cus : 256))) {
        bool ok = true;
        for (int i = 0; i < d->ngroups && ok; ++i) {
            const rtn_conv_group_t& s = d->g[i];
            const long long cells = (long long)s.Hout * s.Wout, M = cells * d->batch;
            ok = s.in && s.out && !((uintptr_t)s.in & 15) && !((uintptr_t)s.out & 15) && s.in_elems >= M * d->Crun &&
                 s.out_elems >= (M - 1) * d->out_ld + d->N;
        }
        long long t64 = 0;
        for (int i = 0; i < d->ngroups; ++i) t64 += ((long long)d->g[i].Hout * d->g[i].Wout * d->batch + 63) / 64;
        const size_t table_b = ((size_t)t64 * 64 * sizeof(uint4) + 255) & ~(size_t)255;       // its slabs start behind the row-info table
        if (ok && workspace_bytes > table_b) {
            const int rc = rtn_wgrad_win_try(h, d, dW, db, db_n, (char*)workspace + table_b, workspace_bytes - table_b);
            if (rc == RTN_OK) h->last_wgrad_impl = 4;
            if (rc <= 0) return rc;
        }
    }

    WParams p;
    memset(&p, 0, sizeof(p));
    long long tiles = 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        WGroup& g = p.g[i];
        if (!s.in || !s.out) return rtn_fail(h, RTN_EINVAL, "wgrad: group %d null x/dY", i);
        if (((uintptr_t)s.in & 15) || ((uintptr_t)s.out & 15)) return rtn_fail(h, RTN_EINVAL, "wgrad: group %d alignment", i);
        if (s.Hin < 1 || s.Win < 1 || s.Hout < 1 || s.Wout < 1 || s.Hout > 32000 || s.Wout > 32000) return rtn_fail(h, RTN_EINVAL, "wgrad: group %d extent", i);
        if ((s.in_img_stride * es) % 16 || ((long long)s.in_row_stride * es) % 16 || (s.out_img_stride * es) % 16 || (s.out_off * es) % 16)
            return rtn_fail(h, RTN_EINVAL, "wgrad: group %d strides not 16-byte multiples", i);
        const long long in_max = (long long)(d->batch - 1) * s.in_img_stride + (long long)(s.Hin - 1) * s.in_row_stride + (long long)(s.Win - 1) * d->pix_stride + d->Crun;
        if (in_max > s.in_elems) return rtn_fail(h, RTN_EBOUNDS, "wgrad: group %d x taps reach %lld of %lld", i, in_max, (long long)s.in_elems);
        const long long cells = (long long)s.Hout * s.Wout;
        const long long dy_max = (long long)(d->batch - 1) * s.out_img_stride + s.out_off + (cells - 1) * d->out_ld + d->N;
        if (s.out_off < 0 || dy_max > s.out_elems) return rtn_fail(h, RTN_EBOUNDS, "wgrad: group %d dY reads reach %lld of %lld", i, dy_max, (long long)s.out_elems);
        if (s.in_elems * es >= (long long)OOB_OFFSET || s.out_elems * es >= (long long)OOB_OFFSET) return rtn_fail(h, RTN_EINVAL, "wgrad: group %d tensor exceeds the 4 GiB descriptor range", i);
        g.x = (const char*)s.in; g.dy = (const char*)s.out;
        g.x_bytes = (unsigned)(s.in_elems * es); g.dy_bytes = (unsigned)(s.out_elems * es);
        g.x_img_stride_b = s.in_img_stride * es;
        g.dy_img_stride_b = s.out_img_stride * es;
        g.dy_off_b = s.out_off * es;
        g.x_row_stride_b = (int)((long long)s.in_row_stride * es);
        g.Hin = s.Hin; g.Win = s.Win; g.Hout = s.Hout; g.Wout = s.Wout;
        g.M = (int)(cells * d->batch);
        g.tile_begin = (int)tiles;
        tiles += (cells * d->batch + 63) / 64;
    }
    if (tiles > (1 << 24)) return rtn_fail(h, RTN_EINVAL, "wgrad: too many pixels");
    const WgradPlan w = wgrad_plan(d, h->num_cus > 0 ? h->num_cus : 256);
    const bool dma = w.dma, dma_small = w.dma_small;
    const int CH = w.CH;
    p.dW = dW;
    p.db = db;
    p.db_n = db_n;
    p.rowinfo = (const uint4*)workspace;
    p.ngroups = d->ngroups;
    p.total_tiles = (int)tiles;
    p.N = d->N;
    p.Ktot = (int)Ktot;
    p.cshift = cshift; p.crun_mask = d->Crun - 1; p.KW = d->KW;
    p.kw_inv = (65536 + d->KW - 1) / d->KW;
    for (int kp = 0; kp < d->KH * d->KW; ++kp)
        if (((kp * p.kw_inv) >> 16) != kp / d->KW) return rtn_fail(h, RTN_EINVAL, "wgrad: KW %d unsupported", d->KW);
    p.pix_stride_b = (int)pix_b;
    p.sy = d->sy; p.sx = d->sx; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.dy_ld_b = d->out_ld * es;
    p.ntiles_k = (int)((Ktot + CH - 1) / CH);
    const long long out_tiles = w.out_tiles;
    long long nsplit = w.nsplit;
    const bool xcd_map = w.xcd_map;
    p.tiles_per_split = w.tiles_per_split;
    // ordered reduction: slabs behind the row-info table when the workspace has room for them (rtn_conv2d_wgrad_workspace_bytes asks
    // for it); otherwise float atomics into dW (the same values, run-to-run differences in the last bits)
    const size_t table_b = ((size_t)tiles * 64 * sizeof(uint4) + 255) & ~(size_t)255;
    const size_t slab_b = (size_t)w.nsplit_used * (size_t)d->N * ((size_t)Ktot + 1) * sizeof(float);
    const bool use_slab = mode != 1 && rtn_env_int("RTN_WGRAD_SLAB", 1) != 0 && workspace_bytes >= table_b + slab_b && (((long long)d->N * Ktot) % 4 == 0);
    if (use_slab) {
        p.slab = (float*)((char*)workspace + table_b);
        p.bslab = p.slab + (size_t)w.nsplit_used * d->N * Ktot;
    }

    if (mode != 2) {         // the table depends on the layer's geometry and the dY / X base offsets only: a caller may build it once
        if (es == 2) hipLaunchKernelGGL((wgrad_rowinfo_kernel<2>), dim3(grid_for(tiles * 64)), dim3(256), 0, h->stream, p, (uint4*)workspace);
        else         hipLaunchKernelGGL((wgrad_rowinfo_kernel<4>), dim3(grid_for(tiles * 64)), dim3(256), 0, h->stream, p, (uint4*)workspace);
        RTN_CHECK_LAUNCH(h, "wgrad_rowinfo_kernel");
        if (mode == 1) return RTN_OK;
    }
    p.out_tiles = (int)out_tiles;
    p.xcd_map = xcd_map ? 1 : 0;
    dim3 grid = xcd_map ? dim3((unsigned)(out_tiles * nsplit)) : dim3((unsigned)out_tiles, (unsigned)nsplit);
    if (dma)            hipLaunchKernelGGL((conv_wgrad_dma_kernel<4, 2, 8>), grid, dim3(512), 0, h->stream, p);
    else if (dma_small) hipLaunchKernelGGL((conv_wgrad_dma_kernel<2, 2, 4>), grid, dim3(256), 0, h->stream, p);
    else if (es == 2) hipLaunchKernelGGL((conv_wgrad_kernel<2>), grid, dim3(256), 0, h->stream, p);
    else              hipLaunchKernelGGL((conv_wgrad_kernel<4>), grid, dim3(256), 0, h->stream, p);
    RTN_CHECK_LAUNCH(h, "conv_wgrad_kernel");
    h->last_wgrad_impl = dma ? 2 : dma_small ? 3 : 0;
    if (use_slab) return rtn_wgrad_finish(h, dW, p.slab, (int)w.nsplit_used, (long long)d->N * Ktot, db, p.bslab, d->N, db ? db_n : 0);
    return RTN_OK;
}

extern "C" int rtn_bias_grad(rtn_handle_t h, const void* dy, int dtype, int64_t rows, int N, int64_t ld, float* db) {
    if (!h) return RTN_EINVAL;
    if (!dy || !db || rows < 1 || N < 1 || ld < N) return rtn_fail(h, RTN_EINVAL, "bias_grad: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "bias_grad: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((ld * es) % 16 || ((uintptr_t)dy & 15)) return rtn_fail(h, RTN_EINVAL, "bias_grad: ld/pointer not 16-byte aligned");
    const int ce = 16 / es, nch = (N + ce - 1) / ce;
    if ((long long)nch * ce > ld) return rtn_fail(h, RTN_EINVAL, "bias_grad: last chunk leaves the row");
    if (nch > 256) return rtn_fail(h, RTN_EINVAL, "bias_grad: N %d too wide", N);
    const int rp = 256 / nch;
    long long blocks = (rows + (long long)rp * 32 - 1) / ((long long)rp * 32);     // ~32 rows per row lane
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    if (es == 2) hipLaunchKernelGGL((bias_grad_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, h->stream, (const char*)dy, (long long)rows, N, (long long)ld * es, db);
    else         hipLaunchKernelGGL((bias_grad_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, h->stream, (const char*)dy, (long long)rows, N, (long long)ld * es, db);
    RTN_CHECK_LAUNCH(h, "bias_grad_kernel");
    return RTN_OK;
}

extern "C" int rtn_zero_insert2(rtn_handle_t h, const void* in, void* out, int dtype, int B, int H, int W, int C, int Hu, int Wu) {
    if (!h) return RTN_EINVAL;
    if (!in || !out || B < 1 || H < 1 || W < 1 || C < 1) return rtn_fail(h, RTN_EINVAL, "zero_insert2: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "zero_insert2: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((C * es) % 16 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return rtn_fail(h, RTN_EINVAL, "zero_insert2: alignment");
    if (Hu < 2 * H - 1 || Wu < 2 * W - 1) return rtn_fail(h, RTN_EINVAL, "zero_insert2: %dx%d too small for %dx%d", Hu, Wu, H, W);
    const int cv = C * es / 16;
    hipLaunchKernelGGL(zero_insert2_kernel, dim3(grid_for((long long)B * Hu * Wu * cv)), dim3(256), 0, h->stream, (const uint4*)in, (uint4*)out, B, H, W, Hu, Wu, cv);
    RTN_CHECK_LAUNCH(h, "zero_insert2_kernel");
    return RTN_OK;
}

extern "C" int rtn_upsample_add_bwd(rtn_handle_t h, const void* d_dst, void* d_src, int dtype, int B, int Hd, int Wd, int Hs, int Ws,
                                    int C, int accumulate) {
    if (!h) return RTN_EINVAL;
    if (!d_dst || !d_src || B < 1 || Hd < 1 || Wd < 1 || Hs < 1 || Ws < 1 || C < 1) return rtn_fail(h, RTN_EINVAL, "upsample_add_bwd: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "upsample_add_bwd: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((C * es) % 16 || ((uintptr_t)d_dst & 15) || ((uintptr_t)d_src & 15)) return rtn_fail(h, RTN_EINVAL, "upsample_add_bwd: alignment");
    const float rh = (float)Hs / (float)Hd, rw = (float)Ws / (float)Wd;
    const long long total = (long long)B * Hs * Ws * (C * es / 16);
    if (es == 2) hipLaunchKernelGGL((upsample_add_bwd_kernel<2>), dim3(grid_for(total)), dim3(256), 0, h->stream, (const char*)d_dst, (char*)d_src, B, Hd, Wd, Hs, Ws, C, rh, rw, accumulate);
    else         hipLaunchKernelGGL((upsample_add_bwd_kernel<4>), dim3(grid_for(total)), dim3(256), 0, h->stream, (const char*)d_dst, (char*)d_src, B, Hd, Wd, Hs, Ws, C, rh, rw, accumulate);
    RTN_CHECK_LAUNCH(h, "upsample_add_bwd_kernel");
    return RTN_OK;
}

extern "C" int rtn_maxpool3x3s2_tfsame_bwd(rtn_handle_t h, const void* x, const void* dy, void* dx, int dtype, int B, int Hin, int Win,
                                           int C, float* scratch_f32, int relu_mask) {
    if (!h) return RTN_EINVAL;
    if (!x || !dy || !dx || !scratch_f32 || B < 1 || Hin < 1 || Win < 1 || C < 1) return rtn_fail(h, RTN_EINVAL, "maxpool_bwd: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "maxpool_bwd: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((C * es) % 16 || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15)) return rtn_fail(h, RTN_EINVAL, "maxpool_bwd: alignment");
    const int Hout = (Hin + 1) / 2, Wout = (Win + 1) / 2;
    int pth = (Hout - 1) * 2 + 3 - Hin; if (pth < 0) pth = 0;
    int ptw = (Wout - 1) * 2 + 3 - Win; if (ptw < 0) ptw = 0;
    const long long nin = (long long)B * Hin * Win * C;
    RTN_HIP(h, hipMemsetAsync(scratch_f32, 0, (size_t)nin * 4, h->stream));
    const long long total = (long long)B * Hout * Wout * (C * es / 16);
    if (es == 2) hipLaunchKernelGGL((maxpool_bwd_kernel<2>), dim3(grid_for(total)), dim3(256), 0, h->stream, (const char*)x, (const char*)dy, scratch_f32, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2);
    else         hipLaunchKernelGGL((maxpool_bwd_kernel<4>), dim3(grid_for(total)), dim3(256), 0, h->stream, (const char*)x, (const char*)dy, scratch_f32, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2);
    RTN_CHECK_LAUNCH(h, "maxpool_bwd_kernel");
    if (es == 2) hipLaunchKernelGGL((cast_from_f32_kernel<2>), dim3(grid_for(nin)), dim3(256), 0, h->stream, (const float*)scratch_f32, (char*)dx, nin, relu_mask ? (const char*)x : (const char*)nullptr);
    else         hipLaunchKernelGGL((cast_from_f32_kernel<4>), dim3(grid_for(nin)), dim3(256), 0, h->stream, (const float*)scratch_f32, (char*)dx, nin, relu_mask ? (const char*)x : (const char*)nullptr);
    RTN_CHECK_LAUNCH(h, "cast_from_f32_kernel");
    return RTN_OK;
}

extern "C" size_t rtn_sumsq_workspace_bytes(void) { return 2048 * sizeof(double); }

extern "C" int rtn_sumsq(rtn_handle_t h, const float* g, const float* scale, int64_t n, double* out, void* workspace, size_t workspace_bytes) {
    if (!h) return RTN_EINVAL;
    if (!g || !out || !workspace || n < 1) return rtn_fail(h, RTN_EINVAL, "sumsq: bad argument");
    if (workspace_bytes < rtn_sumsq_workspace_bytes()) return rtn_fail(h, RTN_ENOMEM, "sumsq: workspace too small");
    const unsigned nb = grid_for(n, 2048);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, h->stream, g, scale, (long long)n, (double*)workspace);
    RTN_CHECK_LAUNCH(h, "sumsq_kernel");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, h->stream, (const double*)workspace, (int)nb, out);
    RTN_CHECK_LAUNCH(h, "sumsq_final_kernel");
    return RTN_OK;
}

extern "C" int rtn_adam_clipnorm_step(rtn_handle_t h, float* w, float* m, float* v, const float* g, const float* gscale,
                                      const float* fold, void* w_fwd, int fwd_dtype, int64_t n, int64_t step, float lr, float beta1,
                                      float beta2, float eps, const double* sumsq, float clipnorm, float grad_mul) {
    if (!h) return RTN_EINVAL;
    if (!w || !m || !v || !g || n < 1 || step < 1) return rtn_fail(h, RTN_EINVAL, "adam: bad argument");
    if (w_fwd && fwd_dtype != RTN_BF16 && fwd_dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "adam: bad forward dtype");
    // Keras 2: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    const unsigned nb = grid_for(n, 4096);
    if (fwd_dtype == RTN_BF16)
        hipLaunchKernelGGL((adam_kernel<2>), dim3(nb), dim3(256), 0, h->stream, w, m, v, g, gscale, fold, (char*)w_fwd, (long long)n, (float)lr_t, beta1, beta2, eps, sumsq, clipnorm, grad_mul);
    else
        hipLaunchKernelGGL((adam_kernel<4>), dim3(nb), dim3(256), 0, h->stream, w, m, v, g, gscale, fold, (char*)w_fwd, (long long)n, (float)lr_t, beta1, beta2, eps, sumsq, clipnorm, grad_mul);
    RTN_CHECK_LAUNCH(h, "adam_kernel");
    return RTN_OK;
}

extern "C" int rtn_sumsq_segments(rtn_handle_t h, const float* g, const float* scale, const int64_t* seg_begin_dev, int nseg, double* out_dev) {
    if (!h) return RTN_EINVAL;
    if (!g || !seg_begin_dev || !out_dev || nseg < 1 || nseg > ADAM_MAX_SEGS) return rtn_fail(h, RTN_EINVAL, "sumsq_segments: bad argument (1..%d segments)", ADAM_MAX_SEGS);
    hipLaunchKernelGGL(sumsq_segments_kernel, dim3((unsigned)nseg), dim3(256), 0, h->stream, g, scale, (const long long*)seg_begin_dev, out_dev);
    RTN_CHECK_LAUNCH(h, "sumsq_segments_kernel");
    return RTN_OK;
}

extern "C" int rtn_adam_clipnorm_step_segments(rtn_handle_t h, float* w, float* m, float* v, const float* g, const float* gscale,
                                               const float* fold, void* w_fwd, int fwd_dtype, int64_t n, int64_t step, float lr, float beta1,
                                               float beta2, float eps, const int64_t* seg_begin_dev, int nseg, const double* sumsq_seg_dev,
                                               int64_t elem_offset, float clipnorm, float grad_mul) {
    if (!h) return RTN_EINVAL;
    if (!w || !m || !v || !g || n < 1 || step < 1) return rtn_fail(h, RTN_EINVAL, "adam_segments: bad argument");
    if (!seg_begin_dev || !sumsq_seg_dev || nseg < 1 || nseg > ADAM_MAX_SEGS || elem_offset < 0) return rtn_fail(h, RTN_EINVAL, "adam_segments: bad segment table");
    if (w_fwd && fwd_dtype != RTN_BF16 && fwd_dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "adam_segments: bad forward dtype");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    const unsigned nb = grid_for(n, 4096);
    if (fwd_dtype == RTN_BF16)
        hipLaunchKernelGGL((adam_segments_kernel<2>), dim3(nb), dim3(256), 0, h->stream, w, m, v, g, gscale, fold, (char*)w_fwd, (long long)n, (float)lr_t, beta1, beta2, eps,
                           (const long long*)seg_begin_dev, nseg, sumsq_seg_dev, (long long)elem_offset, clipnorm, grad_mul);
    else
        hipLaunchKernelGGL((adam_segments_kernel<4>), dim3(nb), dim3(256), 0, h->stream, w, m, v, g, gscale, fold, (char*)w_fwd, (long long)n, (float)lr_t, beta1, beta2, eps,
                           (const long long*)seg_begin_dev, nseg, sumsq_seg_dev, (long long)elem_offset, clipnorm, grad_mul);
    RTN_CHECK_LAUNCH(h, "adam_segments_kernel");
    return RTN_OK;
}

extern "C" int rtn_pad_cast_rows(rtn_handle_t h, const float* in, void* out, int dtype, int64_t rows, int cin, int cout) {
    if (!h) return RTN_EINVAL;
    if (!in || !out || rows < 1 || cin < 1 || cout < cin) return rtn_fail(h, RTN_EINVAL, "pad_cast_rows: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "pad_cast_rows: bad dtype");
    const unsigned nb = grid_for((long long)rows * cout);
    if (dtype == RTN_BF16 && cout % 8 == 0 && !((uintptr_t)out & 15) && (long long)rows * (cout / 8) < (1ll << 31)) {
        const long long items = (long long)rows * (cout / 8);
        hipLaunchKernelGGL(pad_cast_rows8_kernel, dim3(grid_for(items)), dim3(256), 0, h->stream, in, (uint4*)out, (unsigned)items, cin, cout / 8);
        RTN_CHECK_LAUNCH(h, "pad_cast_rows8_kernel");
        return RTN_OK;
    }
    if (dtype == RTN_BF16) hipLaunchKernelGGL((pad_cast_rows_kernel<2>), dim3(nb), dim3(256), 0, h->stream, in, (char*)out, (long long)rows, cin, cout);
    else                   hipLaunchKernelGGL((pad_cast_rows_kernel<4>), dim3(nb), dim3(256), 0, h->stream, in, (char*)out, (long long)rows, cin, cout);
    RTN_CHECK_LAUNCH(h, "pad_cast_rows_kernel");
    return RTN_OK;
}

extern "C" int rtn_maxpool3x3s2_tfsame_fwd_idx(rtn_handle_t h, const void* in, void* out, uint8_t* idx, int dtype, int B, int Hin,
                                               int Win, int C) {
    if (!h) return RTN_EINVAL;
    if (!in || !out || !idx || B < 1 || Hin < 1 || Win < 1 || C < 1) return rtn_fail(h, RTN_EINVAL, "maxpool_fwd_idx: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "maxpool_fwd_idx: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((C * es) % 16 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || ((uintptr_t)idx & 7)) return rtn_fail(h, RTN_EINVAL, "maxpool_fwd_idx: alignment");
    const int Hout = (Hin + 1) / 2, Wout = (Win + 1) / 2;
    int pth = (Hout - 1) * 2 + 3 - Hin; if (pth < 0) pth = 0;
    int ptw = (Wout - 1) * 2 + 3 - Win; if (ptw < 0) ptw = 0;
    const long long total = (long long)B * Hout * Wout * (C * es / 16);
    if (es == 2) hipLaunchKernelGGL((maxpool_fwd_idx_kernel<2>), dim3(grid_for(total)), dim3(256), 0, h->stream, (const char*)in, (char*)out, idx, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2);
    else         hipLaunchKernelGGL((maxpool_fwd_idx_kernel<4>), dim3(grid_for(total)), dim3(256), 0, h->stream, (const char*)in, (char*)out, idx, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2);
    RTN_CHECK_LAUNCH(h, "maxpool_fwd_idx_kernel");
    return RTN_OK;
}

extern "C" int rtn_maxpool3x3s2_tfsame_bwd_idx(rtn_handle_t h, const void* dy, const uint8_t* idx, const void* x, void* dx, int dtype,
                                               int B, int Hin, int Win, int C, int relu_mask) {
    if (!h) return RTN_EINVAL;
    if (!dy || !idx || !dx || (relu_mask && !x) || B < 1 || Hin < 1 || Win < 1 || C < 1) return rtn_fail(h, RTN_EINVAL, "maxpool_bwd_idx: bad argument");
    if (dtype != RTN_BF16 && dtype != RTN_F32) return rtn_fail(h, RTN_EINVAL, "maxpool_bwd_idx: bad dtype");
    const int es = rtn_dtype_size(dtype);
    if ((C * es) % 16 || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)x & 15) || ((uintptr_t)idx & 7)) return rtn_fail(h, RTN_EINVAL, "maxpool_bwd_idx: alignment");
    const int Hout = (Hin + 1) / 2, Wout = (Win + 1) / 2;
    int pth = (Hout - 1) * 2 + 3 - Hin; if (pth < 0) pth = 0;
    int ptw = (Wout - 1) * 2 + 3 - Win; if (ptw < 0) ptw = 0;
    const long long row_items = (long long)Win * (C * es / 16), rows = (long long)B * Hin;
    if (row_items >= (1ll << 31) || rows >= (1ll << 31)) return rtn_fail(h, RTN_EINVAL, "maxpool_bwd_idx: extent");
    const dim3 grid((unsigned)((row_items + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535));
    if (es == 2) hipLaunchKernelGGL((maxpool_bwd_idx_kernel<2>), grid, dim3(256), 0, h->stream, (const char*)dy, idx, (const char*)x, (char*)dx, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2, relu_mask);
    else         hipLaunchKernelGGL((maxpool_bwd_idx_kernel<4>), grid, dim3(256), 0, h->stream, (const char*)dy, idx, (const char*)x, (char*)dx, B, Hin, Win, C, Hout, Wout, pth / 2, ptw / 2, relu_mask);
    RTN_CHECK_LAUNCH(h, "maxpool_bwd_idx_kernel");
    return RTN_OK;
}
