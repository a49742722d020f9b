// rtn_conv_halo8r.hip — generation 7: the head-tower convolution with its FILTERS IN REGISTERS.
//
// Same layers, tile and halo staging as generation 4 (rtn_conv_halo8.hip: stride-1 'same' 3x3, 129..256 output channels,
// persistent 256 x 256 tiles, two wave groups one barrier apart), for
//   model/defineModel.py:101-117,155-163  the 4 x [3x3 conv 256 + ReLU] towers of both heads over P3..P7 (one grouped launch),
//   model/defineModel.py:183-203           P3 / P4 (3x3, 256 -> 256), keras_resnet's res4 branch2b (3x3, 256 -> 256),
// but the weights never pass through LDS.  What generation 4's time line showed (profiles/r2_v3_halo8_phase_timeline.txt,
// r2_v3_halo8_ablation.txt): a barrier interval lasts 430-520 clocks for 256 clocks of MFMAs because the READING wave group
// (6 ds_read_b128 + 1.3 LDS-DMA issues per phase, then the wait for the last fragment) is slower than the multiplying one, and
// two thirds of those reads and three quarters of the LDS-DMA issues are the B (weight) tile.  Here:
//   * the weights come from a copy in MFMA-FRAGMENT ORDER (rtn_pack_frag_weights): fragment (K block kb of 64, k half h, column
//     quarter wn, fragment jw) is 1 KiB = lane (kq, rho) x 16 bytes, so one buffer_load_dwordx4 per wave and fragment is a fully
//     coalesced 1-KiB read straight into the MFMA operand registers (L2-resident: every workgroup streams the same 1.2 MB);
//     three register buffers of four fragments rotate over the half steps, loaded one whole K step ahead.
//   * wave tile 128 pixels x 64 channels (2 x 4 waves) instead of 64 x 128: a phase reads FOUR pixel fragments from LDS (half the
//     LDS bytes per MFMA), issues two weight loads and runs the same 16 MFMAs; the only LDS-DMA left is the halo (1.3 pieces per
//     wave and K step), waited for once per (kh, chunk) group.
//   * transposed products (weights = MFMA A operand, pixels = B operand) with weight row rho of fragment (wn, jw) = channel
//     64 wn + 16 (rho >> 2) + 4 jw + (rho & 3): lane (kq, c) ends up with 16 CONSECUTIVE channels 64 wn + 16 kq .. of pixel c and
//     stores 2 x 16 bytes per pixel fragment straight from registers.  Bias table in LDS (no VMEM load at a tile start).
// The accumulation order per output element (K steps, k halves, the MFMA's own k order) is generation 4's: results are bit-identical.
//
// LDS: halo images 2 x 32 KiB at 0 (256 rows x 128 B; row R-1 = zeros), bias table 1 KiB at 64 KiB + 2 KiB.
#include "rtn_internal.h"

#ifndef RTN_R7_ABLATE                                  // timing ablations (wrong results): 1 = no weight loads inside the loop, 2 = no halo staging
#define RTN_R7_ABLATE 0
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned OOB = 0xFFFFFF00u;                 // beyond every descriptor: loads return zeros, stores are dropped
constexpr int R7_THREADS = 512;
constexpr unsigned A_TOGGLE = 0x8000u;                // halo buffers at LDS 0 and 32 KiB
constexpr unsigned BIAS_OFF = 0x10000u + 2048u;       // behind the slack the discarded rows' centre taps may read
constexpr int R7_LDS = 0x10000 + 2048 + 1024;

struct R7Group {
    const char* in;
    char* out;
    const char* res;              // RTN_CONV_RES_SAME source (dense [M][res_ld]) or null
    const char* mask;             // RTN_CONV_RELU_MASK source (dense [M][mask_ld]) or null
    unsigned in_bytes, out_bytes, res_bytes, mask_bytes;
    int Hin, Win, M, tile_begin;
    int in_row_stride_b;
    float inv_cells, inv_w;
};

struct R7Params {
    R7Group g[RTN_MAX_GROUPS];
    const char* wf;               // fragment-order filters (rtn_pack_frag_weights)
    const float* bias;
    unsigned wf_bytes;
    int ngroups, ntiles;
    int N, KH, nchunk, pad_t, relu, out_ld, pix_b;
    int res_ld, mask_ld, mask_pre;
    int xcd, kh_fast;
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

// 64 lanes x 16 B from (descriptor, per-lane byte offset) to LDS bytes [lds_addr, lds_addr + 1024): asm, so that hipcc neither
// counts nor drains it; the kernel's own counted wait covers it (once per group).
__device__ __forceinline__ void dma16(const i32x4& srd, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd)
                 : "memory");
}

__device__ __forceinline__ void divmod24(int f, int d, float inv, int& q, int& r) {
    q = (int)((float)f * inv);
    r = f - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

// see rtn_conv_halo8.hip / profiles/r3_store_hazard_isa.txt: four wait states between a 16-byte store and the rewrite of its data
#define RTN_STORE_GUARD(V) asm volatile("s_nop 3" :: "v"(V.x), "v"(V.y), "v"(V.z), "v"(V.w));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// NP = pixel fragments (16 rows) per wave: halo images of R = 32 NP rows (256 or 192), TM = R - 3 output rows per tile.
// EPI: bit 0 = residual add, bit 1 = ReLU mask (the data-gradient launches of the training step), as in generation 4.
template <int NP, int EPI>
__global__ __launch_bounds__(R7_THREADS, 2) void conv_halo8r_kernel(const R7Params p) {
    constexpr int KW = 3;
    constexpr int NPH = NP / 2;                        // pixel fragments per phase
    constexpr int R = 32 * NP;
    constexpr int TM = R - KW;
    constexpr int NPC = R / 64;                        // halo pieces (8 rows) per wave and group
    constexpr unsigned ZERO_ROW = (unsigned)(R - 1) * 128u;
    static_assert(NP == 8 || NP == 6, "256- or 192-row tiles");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;           // wave tile: pixels [16 NP wm, + 16 NP) x channels [64 wn, + 64); SIMD partners
                                                       // (w, w + 4) sit in different groups (wm) and read the SAME weight fragments
    const int lr = lane >> 3, sc = (lane & 7) ^ lr;    // halo staging: row inside an 8-row piece, SOURCE chunk (swizzle on the source)
    const int lrow_c = lane & 15, kq = lane >> 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    const int G = p.KH * p.nchunk;                     // (kh, chunk) groups per tile
    const int nchunk = p.nchunk;
    const __amdgpu_buffer_rsrc_t wf_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, (int)__builtin_amdgcn_readfirstlane((int)p.wf_bytes), 0x00020000);
    const int lane16 = lane * 16;

    if (t < 256) reinterpret_cast<float*>(lds + BIAS_OFF)[t] = (p.bias && t < p.N) ? p.bias[t] : 0.f;

    // ---- staging cursor (the tile whose halos are being staged)
    unsigned hbase[NPC];
    int hiy[NPC];
    i32x4 in_srd = make_srd(p.wf, 0u);
    int st_Hin = 1, st_row_b = 0;
    auto stage_tile = [&](int T) {                     // T uniform; T >= ntiles: nothing to stage (zeros)
        if (T >= p.ntiles) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) { hiy[i] = -(1 << 28); hbase[i] = 0; }
            return;
        }
        int gi = 0;
#pragma unroll
        for (int i = 1; i < RTN_MAX_GROUPS; ++i)
            if (i < p.ngroups && T >= p.g[i].tile_begin) gi = i;
        const R7Group& Gs = p.g[gi];
        const int m0 = (T - Gs.tile_begin) * TM;
        const int cells = Gs.Hin * Gs.Win;
        in_srd = make_srd(Gs.in, Gs.in_bytes);
        st_Hin = Gs.Hin;
        st_row_b = Gs.in_row_stride_b;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int h = (i * 8 + wave) * 8 + lr;
            const int f = m0 + h - 1;                  // pad_l = 1
            if (h < R - 1 && f >= 0 && f < Gs.M) {
                int b, rem, y, x;
                divmod24(f, cells, Gs.inv_cells, b, rem);
                divmod24(rem, Gs.Win, Gs.inv_w, y, x);
                hiy[i] = y;
                hbase[i] = (unsigned)f * (unsigned)p.pix_b + (unsigned)sc * 16u;
            } else {
                hiy[i] = -(1 << 28);
                hbase[i] = 0;
            }
        }
    };
    auto stage_a = [&](int i, int kh, int cc, unsigned abuf_addr) {
        const int dy = kh - p.pad_t;
        const unsigned delta = (unsigned)(dy * st_row_b + cc * 128);
        const bool ok = (unsigned)(hiy[i] + dy) < (unsigned)st_Hin;
#if !(RTN_R7_ABLATE & 2)
        dma16(in_srd, ok ? hbase[i] + delta : OOB, lds_base + abuf_addr + (unsigned)(wave * 1024 + i * 8192));
#endif
    };

    // ---- compute tile: fragment read offsets.  Centre tap: one register (fragment ip = + 2048 ip: 16 rows further, the same
    // swizzle); side taps: one per fragment, the zero row where the tap leaves the image or the row is past the tile.
    unsigned a_cur = 0u;
    unsigned arow0[NP], arow2[NP], arow1 = 0u;
    auto compute_tile = [&](int T, int& gi_out, int& m0_out) {
        int gi = 0;
#pragma unroll
        for (int i = 1; i < RTN_MAX_GROUPS; ++i)
            if (i < p.ngroups && T >= p.g[i].tile_begin) gi = i;
        const R7Group& Gc = p.g[gi];
        const int m0 = (T - Gc.tile_begin) * TM;
        const int cells = Gc.Hin * Gc.Win;
        gi_out = gi;
        m0_out = m0;
        int lrow = lrow_c;                             // opaque per tile: the compiler would otherwise hoist 2 NP per-fragment constants
        asm volatile("" : "+v"(lrow));                 // out of the tile loop and spill them
        {
            const int rr = wm * (16 * NP) + lrow + 1;
            arow1 = a_cur + (unsigned)(rr * 128 + ((kq ^ (rr & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int rloc = wm * (16 * NP) + i * 16 + lrow;
            const int m = m0 + rloc;
            const int mc = m < Gc.M ? m : Gc.M - 1;
            int b, rem, y, x;
            divmod24(mc, cells, Gc.inv_cells, b, rem);
            divmod24(rem, Gc.Win, Gc.inv_w, y, x);
            const bool off0 = rloc >= TM || x == 0;
            const bool off2 = rloc >= TM || x == Gc.Win - 1;
            const int r2 = rloc + 2;
            arow0[i] = a_cur + (off0 ? ZERO_ROW : (unsigned)(rloc * 128 + ((kq ^ (rloc & 7)) << 4)));
            arow2[i] = a_cur + (off2 ? ZERO_ROW : (unsigned)(r2 * 128 + ((kq ^ (r2 & 7)) << 4)));
        }
    };

    f32x4 acc[4][NP];

    auto item_of = [&](int v) {
        if (v >= p.ntiles || !p.xcd) return v;
        const int x = v & 7, j = v >> 3, base = p.ntiles >> 3, rem = p.ntiles & 7;
        return x * base + (x < rem ? x : rem) + j;
    };

    // weight fragments: three buffers of four (jw), rotating over the half steps
    u32x4 fb[3][4];
#define R7_LDW(BUF, J, KCOL, H)                                                                       \
    fb[BUF][J] = __builtin_amdgcn_raw_buffer_load_b128(wf_rsrc, lane16 + (J) * 1024,                  \
        (int)__builtin_amdgcn_readfirstlane((int)((KCOL) * 256u + (unsigned)((H) * 16384 + wn * 4096))), 0);

#if RTN_R7_ABLATE & 1
#define R7_LDWL(BUF, J, KCOL, H)
#else
#define R7_LDWL(BUF, J, KCOL, H) R7_LDW(BUF, J, KCOL, H)
#endif

    // ---- prologue: halo of the first group, weights of step 0
    int vidx = blockIdx.x;
    int rt = item_of(vidx);
    if (rt >= p.ntiles) rt = p.ntiles;
    stage_tile(rt);
#pragma unroll
    for (int i = 0; i < NPC; ++i) stage_a(i, 0, 0, 0u);
    {
        const unsigned kc0 = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) { R7_LDW(0, j, kc0, 0) }
#pragma unroll
        for (int j = 0; j < 4; ++j) { R7_LDW(1, j, kc0, 1) }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // once: the halo pieces (and the weights of step 0)
    if (wm == 1) __builtin_amdgcn_s_barrier();         // group 1 runs one barrier behind group 0
    __builtin_amdgcn_s_barrier();

#define R7_MFMA(BUF, HALF)                                                                           \
    __builtin_amdgcn_s_barrier();                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < NPH; ++i_)                                           \
            acc[j_][(HALF) * NPH + i_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                    \
                __builtin_bit_cast(bf16x8, fb[BUF][j_]), __builtin_bit_cast(bf16x8, fa[i_]), acc[j_][(HALF) * NPH + i_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_barrier();
#define R7_RD(KWI, H, HALF)                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < NPH; ++i_) {                                             \
        const int ip_ = (HALF) * NPH + i_;                                                           \
        const unsigned a_ = (KWI) == 1 ? ((arow1 ^ ((H) * 64u)) + (unsigned)ip_ * 2048u)             \
                                       : (((KWI) == 0 ? arow0[ip_] : arow2[ip_]) ^ ((H) * 64u));     \
        fa[i_] = *reinterpret_cast<const uint4*>(lds + a_);                                          \
    }
    // One K step = tap KWI of the current group: four phases (k half h, pixel half) of {NPH fragment reads + two weight loads of
    // the NEXT step's same k half [+ a halo piece of the next group] | barrier | 4 NPH MFMAs | barrier}.  Half step t = 2 KWI + h
    // multiplies from buffer t % 3 and loads into (t + 2) % 3, which half step t - 1 has finished with.
#define R7_STEP(KWI)                                                                                 \
    {                                                                                                \
        const unsigned kc_n1 = (KWI) + 1 < KW ? kcol_g + ((KWI) + 1) * kw_stride : kcol_g1;          \
        uint4 fa[NPH];                                                                               \
        R7_RD(KWI, 0, 0)                                                                             \
        R7_LDWL((2 * (KWI) + 2) % 3, 0, kc_n1, 0) R7_LDWL((2 * (KWI) + 2) % 3, 1, kc_n1, 0)          \
        R7_MFMA((2 * (KWI)) % 3, 0)                                                                  \
        R7_RD(KWI, 0, 1)                                                                             \
        R7_LDWL((2 * (KWI) + 2) % 3, 2, kc_n1, 0) R7_LDWL((2 * (KWI) + 2) % 3, 3, kc_n1, 0)          \
        if (2 * (KWI) < NPC && (KWI) < 2) stage_a(2 * (KWI), kh1, cc1, a_cur ^ A_TOGGLE);            \
        R7_MFMA((2 * (KWI)) % 3, 1)                                                                  \
        R7_RD(KWI, 1, 0)                                                                             \
        R7_LDWL((2 * (KWI) + 3) % 3, 0, kc_n1, 1) R7_LDWL((2 * (KWI) + 3) % 3, 1, kc_n1, 1)          \
        R7_MFMA((2 * (KWI) + 1) % 3, 0)                                                              \
        R7_RD(KWI, 1, 1)                                                                             \
        R7_LDWL((2 * (KWI) + 3) % 3, 2, kc_n1, 1) R7_LDWL((2 * (KWI) + 3) % 3, 3, kc_n1, 1)          \
        if (2 * (KWI) + 1 < NPC && (KWI) < 2) stage_a(2 * (KWI) + 1, kh1, cc1, a_cur ^ A_TOGGLE);    \
        /* the next group's halo must have landed before its first read (the next phase): every halo piece is older than the */ \
        /* eight weight loads of this step; 4 instead of 8 also covers the loads the next phase multiplies from anyway */     \
        if ((KWI) == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                             \
        R7_MFMA((2 * (KWI) + 1) % 3, 1)                                                              \
    }

    const unsigned kw_stride = (unsigned)(nchunk * 128);           // K bytes (of the [N][K] layout) between the taps of a kernel row
    while (rt < p.ntiles) {
        int gi, m0;
        compute_tile(rt, gi, m0);
        {
            // accumulators start at the bias of the lane's 16 channels (64 wn + 16 kq + 4 jw + r)
            const float* bt = reinterpret_cast<const float*>(lds + BIAS_OFF) + wn * 64 + kq * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt + 4 * j);
#pragma unroll
                for (int i = 0; i < NP; ++i) acc[j][i] = b4;
            }
        }
        int kh = 0, cc = 0;
        int rt1 = rt;
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            int kh1 = kh, cc1 = cc;
            if (p.kh_fast) { if (++kh1 == p.KH) { kh1 = 0; ++cc1; } }
            else if (++cc1 == nchunk) { cc1 = 0; ++kh1; }
            if (g + 1 == G) {
                vidx += (int)gridDim.x;
                rt1 = item_of(vidx);
                if (rt1 >= p.ntiles) rt1 = p.ntiles;
                kh1 = 0; cc1 = 0;
                stage_tile(rt1);
            }
            const unsigned kcol_g = (unsigned)((kh * KW * nchunk + cc) * 128);
            const unsigned kcol_g1 = (unsigned)((kh1 * KW * nchunk + cc1) * 128);
            R7_STEP(0)
            R7_STEP(1)
            R7_STEP(2)
#pragma unroll
            for (int i = 0; i < NP; ++i) { arow0[i] ^= A_TOGGLE; arow2[i] ^= A_TOGGLE; }
            arow1 ^= A_TOGGLE;
            a_cur ^= A_TOGGLE;
            kh = kh1; cc = cc1;
        }
        // ---- epilogue: [mask] [+ residual] [mask] ReLU, bf16, 2 NP stores of 16 B per lane
        {
            const R7Group& Gc = p.g[gi];
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)Gc.out, 0, (int)__builtin_amdgcn_readfirstlane((int)Gc.out_bytes), 0x00020000);
            const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 1) ? Gc.res : Gc.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 1) ? Gc.res_bytes : 0u)), 0x00020000);
            const __amdgpu_buffer_rsrc_t mask_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((EPI & 2) ? Gc.mask : Gc.out), 0, (int)__builtin_amdgcn_readfirstlane((int)((EPI & 2) ? Gc.mask_bytes : 0u)), 0x00020000);
            const int ncol = wn * 64 + kq * 16;
            const bool ok0 = ncol < p.N, ok1 = ncol + 8 < p.N;
            u32x4 rq[2][2], mq[2][2];               // residual / mask of fragment i (ping-pong: fragment i + 1 is in flight)
            auto fetch = [&](int i, int par) {
                const int rloc = wm * (16 * NP) + i * 16 + lrow_c;
                const int m = m0 + rloc;
                const bool ok = rloc < TM && m < Gc.M;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const bool okh = ok && (hh ? ok1 : ok0);
                    if (EPI & 1) rq[par][hh] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, (int)(okh ? ((unsigned)m * (unsigned)p.res_ld + (unsigned)(ncol + 8 * hh)) * 2u : OOB), 0, 0);
                    if (EPI & 2) mq[par][hh] = __builtin_amdgcn_raw_buffer_load_b128(mask_rsrc, (int)(okh ? ((unsigned)m * (unsigned)p.mask_ld + (unsigned)(ncol + 8 * hh)) * 2u : OOB), 0, 0);
                }
            };
            if (EPI) fetch(0, 0);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (EPI && i + 1 < NP) fetch(i + 1, (i + 1) & 1);
                const int rloc = wm * (16 * NP) + i * 16 + lrow_c;
                const int m = m0 + rloc;
                const bool ok = rloc < TM && m < Gc.M;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[2 * hh + (e >> 2)][i][e & 3];
                    if (EPI) {
                        const u32x4 rw = rq[i & 1][hh], mw = mq[i & 1][hh];
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
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    u32x4 o;
                    o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
                    const bool okh = ok && (hh ? ok1 : ok0);
                    const unsigned off = okh ? ((unsigned)m * (unsigned)p.out_ld + (unsigned)(ncol + 8 * hh)) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, (int)off, 0, 0);
                    RTN_STORE_GUARD(o)
                }
            }
        }
        rt = rt1;
    }
#undef R7_STEP
#undef R7_RD
#undef R7_MFMA
#undef R7_LDWL
#undef R7_LDW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may land after the workgroup has released its LDS
    if (wm == 0) __builtin_amdgcn_s_barrier();
}

// [w_rows][Kbytes] (K contiguous) -> fragment order: 16-byte unit u = ((((kb * 2 + h) * 4 + wn) * 4 + jw) * 64 + 16 kq + rho) holds
// filter 64 wn + 16 (rho >> 2) + 4 jw + (rho & 3), k bytes [128 kb + 64 h + 16 kq, + 16); filters >= w_rows are zeros.
__global__ void pack_frag_kernel(const uint4* __restrict__ w, uint4* __restrict__ wf, int w_rows, int kunits /* Kbytes / 16 */, long long units) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const int l = (int)(u & 63), jw = (int)((u >> 6) & 3), wn = (int)((u >> 8) & 3), h = (int)((u >> 10) & 1);
    const long long kb = u >> 11;
    const int rho = l & 15, kq = l >> 4;
    const int n = 64 * wn + 16 * (rho >> 2) + 4 * jw + (rho & 3);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (n < w_rows) v = w[(long long)n * kunits + kb * 8 + h * 4 + kq];
    wf[u] = v;
}

}  // namespace

extern "C" int rtn_pack_frag_weights(rtn_handle_t h, const void* w, void* wf, int w_rows, int N, int64_t Ktot) {
    if (!h || !w || !wf || w_rows < 1 || N < 1 || N > 256 || w_rows < N || Ktot < 64 || Ktot % 64 || ((uintptr_t)w & 15) || ((uintptr_t)wf & 15))
        return rtn_fail(h, RTN_EINVAL, "rtn_pack_frag_weights: bf16 [w_rows >= N][Ktot] with N <= 256, Ktot a multiple of 64, 16-byte aligned");
    const long long units = Ktot * 2 / 128 * 2048;     // 256 filters x Ktot x 2 bytes / 16
    const int kunits = (int)(Ktot * 2 / 16);
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, h->stream,
                       (const uint4*)w, (uint4*)wf, w_rows < 256 ? w_rows : 256, kunits, units);
    RTN_CHECK_LAUNCH(h, "pack_frag_kernel");
    return RTN_OK;
}

// Launcher.  Returns RTN_OK after a launch, 1 when the layer is not one this kernel takes (the caller falls through to
// generation 4), < 0 on a launch error.
int rtn_conv_halo8r_try(rtn_handle_t h, const rtn_conv_desc_t* d, int grid_limit, bool forced, int mi_force) {
    if (d->dtype != RTN_BF16 || !d->w_frag || ((uintptr_t)d->w_frag & 15)) return 1;
    if (d->KW != 3 || d->KH < 1 || d->KH > 7 || d->sy != 1 || d->sx != 1 || d->pad_l != 1) return 1;
    if (d->flags & ~(RTN_CONV_RELU | RTN_CONV_RES_SAME | RTN_CONV_RELU_MASK | RTN_CONV_MASK_PRE)) return 1;
    if ((d->flags & RTN_CONV_MASK_PRE) && !(d->flags & RTN_CONV_RELU_MASK)) return 1;
    const int epi = ((d->flags & RTN_CONV_RES_SAME) ? 1 : 0) | ((d->flags & RTN_CONV_RELU_MASK) ? 2 : 0);
    if (d->N <= 128 || d->N > 256 || d->w_rows < d->N || d->N % 8 || d->out_ld % 8) return 1;
    if (d->Crun != d->pix_stride || (d->Crun * 2) % 128 || d->Crun <= 0) return 1;
    if (d->pad_t < 0 || d->pad_t >= d->KH) return 1;
    if ((uintptr_t)d->bias & 15) return 1;
    const int nchunk = d->Crun * 2 / 128;
    const long long Kbytes = (long long)d->KH * d->KW * d->Crun * 2;
    if (Kbytes * 256 >= 0xFFFFFF00ll) return 1;
    R7Params p;
    memset(&p, 0, sizeof(p));
    const int cus = h->num_cus > 0 ? h->num_cus : 256;
    // tile height by rounds of workgroups x rows, as generation 4 (RTN_CONV_H8_MI pins it: 4 = 256 rows, 3 = 192)
    int mi = 0;
    {
        double best = 0;
        for (int cand = 4; cand >= 3; --cand) {
            if (mi_force >= 3 && mi_force <= 4 && cand != mi_force) continue;
            long long t = 0;
            for (int i = 0; i < d->ngroups; ++i) t += ((long long)d->g[i].Hout * d->g[i].Wout * d->batch + 64 * cand - 4) / (64 * cand - 3);
            const double cost = (double)((t + cus - 1) / cus) * (cand + 0.3);
            if (mi == 0 || cost < best * 0.97) { best = cost; mi = cand; }
        }
        if (mi == 0) return 1;
    }
    const int TM = 64 * mi - 3;
    long long tiles = 0;
    for (int i = 0; i < d->ngroups; ++i) {
        const rtn_conv_group_t& s = d->g[i];
        if (s.Hin != s.Hout || s.Win != s.Wout || s.in_row_stride != (long long)s.Win * d->pix_stride ||
            s.in_img_stride != (long long)s.Hin * s.in_row_stride) return 1;
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
        R7Group& g = p.g[i];
        g.res = (epi & 1) ? (const char*)s.res : nullptr;
        g.mask = (epi & 2) ? (const char*)s.mask : nullptr;
        g.res_bytes = (epi & 1) ? (unsigned)(s.res_elems * 2) : 0u;
        g.mask_bytes = (epi & 2) ? (unsigned)(s.mask_elems * 2) : 0u;
        g.in = (const char*)s.in;
        g.out = (char*)s.out;
        g.in_bytes = (unsigned)(s.in_elems * 2);
        g.out_bytes = (unsigned)(s.out_elems * 2);
        g.Hin = s.Hin; g.Win = s.Win; g.M = (int)M;
        g.tile_begin = (int)tiles;
        g.in_row_stride_b = (int)(s.in_row_stride * 2);
        g.inv_cells = 1.0f / (float)cells;
        g.inv_w = 1.0f / (float)s.Win;
        tiles += (M + TM - 1) / TM;
    }
    if (tiles < 1 || tiles > 0x3fffffff) return 1;
    if (!forced && tiles * 2 < cus) return 1;          // small grids: generation 4's K slices / generation 2's narrow tiles
    p.wf = (const char*)d->w_frag;
    p.bias = d->bias;
    p.wf_bytes = (unsigned)(Kbytes * 256);
    p.ngroups = d->ngroups;
    p.ntiles = (int)tiles;
    p.N = d->N;
    p.KH = d->KH;
    p.nchunk = nchunk;
    p.pad_t = d->pad_t;
    p.relu = (d->flags & RTN_CONV_RELU) ? 1 : 0;
    p.out_ld = d->out_ld;
    p.pix_b = d->pix_stride * 2;
    p.res_ld = (epi & 1) ? d->g[0].res_ld : 0;
    p.mask_ld = (epi & 2) ? d->g[0].mask_ld : 0;
    p.mask_pre = (d->flags & RTN_CONV_MASK_PRE) ? 1 : 0;
    p.xcd = rtn_env_int("RTN_CONV_XCD", 1) != 0;
    p.kh_fast = rtn_env_int("RTN_CONV_H8_KHFAST", 1) != 0;
    int grid = cus;
    if (grid_limit > 0 && grid_limit < grid) grid = grid_limit;
    if (grid > p.ntiles) grid = p.ntiles;
#define RTN_R7_LAUNCH(NP_, EP)                                                                           \
    do {                                                                                                 \
        static bool attr_set = false;                                                                    \
        if (!attr_set) {                                                                                 \
            RTN_HIP(h, hipFuncSetAttribute((const void*)conv_halo8r_kernel<NP_, EP>,                     \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, R7_LDS));         \
            attr_set = true;                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((conv_halo8r_kernel<NP_, EP>), dim3((unsigned)grid), dim3(R7_THREADS), R7_LDS, h->stream, p); \
    } while (0)
#define RTN_R7_PICK(NP_)                                                                                 \
    do {                                                                                                 \
        if (epi == 0) RTN_R7_LAUNCH(NP_, 0); else if (epi == 1) RTN_R7_LAUNCH(NP_, 1);                   \
        else if (epi == 2) RTN_R7_LAUNCH(NP_, 2); else RTN_R7_LAUNCH(NP_, 3);                            \
    } while (0)
    if (mi == 4) RTN_R7_PICK(8); else RTN_R7_PICK(6);
#undef RTN_R7_PICK
#undef RTN_R7_LAUNCH
    RTN_CHECK_LAUNCH(h, "conv_halo8r_kernel");
    return RTN_OK;
}
