// rtn_bottleneck.hip — a keras_resnet identity bottleneck block of the 64-channel stage as ONE kernel:
//   branch2b (3x3, 64 -> 64, BN, ReLU) -> branch2c (1x1, 64 -> 256, BN) + shortcut + ReLU -> [next block's branch2a (1x1, 256 -> 64, BN, ReLU)]
// (keras_resnet bottleneck_2d, called from model/defineModel.py:376-380; the reference executes these as three Conv2D ops, three
// BatchNormalization ops, an Add and three ReLUs.)  res2 is HBM-bound: unfused, a block moves 1.09 GB at batch 8 (the 64-channel
// tensors twice each, the 256-channel tensor four times); fused it reads the 64-channel input and the shortcut once and writes the
// 256-channel output and the next block's 64-channel input once: 0.68 GB.
//
// How the three GEMMs chain without LDS or barriers: every product is computed TRANSPOSED, D^T[channel][pixel] = W[channel][k] .
// X^T[k][pixel] — the weights are the MFMA's A operand (LDS-resident for the whole kernel), the pixels its columns.  In the
// 16x16x32 layout a lane (q, c) then owns, per 16-channel fragment, 4 consecutive channel ROWS of pixel c — and the B operand of the
// next product wants, from lane (kq = q, column c), 8 consecutive k of pixel c.  With the weight rows of each product stored
// permuted (row 16 f + 4 q + r  <->  channel 32 (f >> 1) + 8 q + 4 (f & 1) + r) two fragments of a lane ARE those 8 values: bias,
// ReLU, pack to bf16, and the accumulators of one product are the next product's B operand in place (cdna_hip_programming.md §3,
// "An accumulator tile as the next MFMA's operand").  The same registers are what is stored: 16 bytes = 8 consecutive channels.
//
// Work decomposition: a wave owns a strip of 32 consecutive pixels (two 16-pixel column fragments) and runs the whole chain on it;
// waves never synchronise (the only shared data, the weights, is read-only).  The 3x3's input fragments come straight from global
// memory (16 B per lane: pixel x tap x 8 channels; padding taps get an out-of-range buffer offset = zeros; only the centre column
// of the taps is loaded, the left / right ones are lane shifts of it - "shifted taps" below), the shortcut likewise.
// LDS: 17 weight images of [64 rows][128 B] (9 taps of W2b, 4 output chunks of W2c, 4 k-chunks of W2a') + biases = 137.5 KiB.
#include "rtn_internal.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned BK_OOB = 0xFFFF0000u;       // beyond every descriptor (< 2 GiB) even with the largest uniform offset added
constexpr int IMG = 64 * 128;                          // one weight image: 64 rows x 128 B
constexpr int W2B_OFF = 0, W2C_OFF = 9 * IMG, W2A_OFF = 13 * IMG, BIAS_OFF = 17 * IMG;
constexpr int RING_OFF = 9 * IMG;                      // PROJ + TAIL: two slots of [branch2c | projection | next branch2a] images of one 64-channel chunk
constexpr int XST_OFF = BIAS_OFF + (64 + 256 + 64) * 4;      // per wave 2 KB: the transpose buffer of the line-shaped stores
constexpr int BK_LDS = XST_OFF + 8 * 2048;

struct BkParams {
    const char* ain;        // [M][64]  bf16: this block's branch2a output (after BN + ReLU)
    const char* xin;        // [M][256] bf16: the block input (identity shortcut)
    char* xout;             // [M][256] bf16
    char* aout;             // [M][64]  bf16 or nullptr: the next block's branch2a output
    char* h1out;            // [M][64]  bf16 or nullptr: branch2b's output (training keeps it for the backward pass)
    const char* w2b;        // [64][3*3*64] bf16, BN folded
    const char* w2c;        // [256][64]
    const char* w2a;        // [64][256] (next block), or nullptr
    const char* pin;        // PROJ: [M][64] bf16, the block INPUT (pool1): the shortcut is wproj x pin, computed here
    const char* wproj;      // PROJ: [256][64] rows w2c_ld elements apart (the K-concatenated [branch2c | branch1] filters: wproj = w2c + 64)
    int w2c_ld;             // elements between rows of w2c / wproj (64, or 128 for the K-concatenated filters)
    const float* b2b;       // [64], [256], [64] f32 (folded BN shifts)
    const float* b2c;
    const float* b2a;
    int M, H, W, nstrips;
    float inv_cells, inv_w;
    int dbg;                // always 0 (the round-2 timing ablations: 1 drop the 3x3's loads, 8 the a_out stores; profiles/r2_v2_bottleneck_fused.txt)
};

__device__ __forceinline__ int perm_row(int rho) {     // MFMA row (16 f + 4 q + r) -> channel 32 (f >> 1) + 8 q + 4 (f & 1) + r
    const int f = rho >> 4, q = (rho >> 2) & 3, r = rho & 3;
    return 32 * (f >> 1) + 8 * q + 4 * (f & 1) + r;
}

__device__ __forceinline__ void divmod24(int f, int d, float inv, int& q, int& r) {
    q = (int)((float)f * inv);
    r = f - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float relu(float v) { return v > 0.f ? v : 0.f; }
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// Store-data guard (root cause and machine-code excerpts: profiles/r3_store_hazard_isa.txt).  gfx940+ needs two wait states between a
// VMEM store of more than 64 bits and a VALU write of its data VGPRs; LLVM pads them (s_nop 1) EXCEPT for buffer stores whose soffset
// is an SGPR, which its hazard table treats as immune (GCNHazardRecognizer::createsVALUHazard).  hipcc keeps this kernel's chunk
// offsets in SGPRs, so the x_out stores of the no-TAIL variant (whose data registers are dead after the store) were followed after
// ZERO or one wait state by the VALU that re-used dword 0: a few hundred corrupted pixels per launch at full size, different on every
// run.  The guard names the data registers as inputs of an `s_nop RTN_BK_STORE_NOPS`, so they stay live and unmodified for
// RTN_BK_STORE_NOPS + 1 wait states whatever the compiler schedules; tools/scan_store_hazard.py (a CPU test) checks every 16-byte
// store of the built library for that distance.
#ifndef RTN_BK_STORE_NOPS
#define RTN_BK_STORE_NOPS 3
#endif
#if RTN_BK_STORE_NOPS < 0
#define BK_STORE_GUARD(V)
#else
#define BK_STORE_GUARD(V) asm volatile("s_nop %4" :: "v"(V.x), "v"(V.y), "v"(V.z), "v"(V.w), "n"(RTN_BK_STORE_NOPS));
#endif

// PROJ: the stage's FIRST block (res2a).  Its shortcut is not the block input but branch1 = conv1x1(block input; wproj): four more
// weight images sit where TAIL keeps the next branch2a's, the two 16-byte fragments of the input pixel (64 channels) are loaded once
// per strip, and every output chunk gets 16 more MFMAs instead of 4 shortcut loads.  b2c then holds b2c + b1.
// Shifted taps (round 4; profiles/r4_bottleneck_shifted_taps.txt): the 3x3's left / right taps are not loaded.  In the flattened pixel order the left neighbour of the pixel in lane c is the
// pixel in lane c - 1, so tap (kh, kw = 0) of a fragment IS tap (kh, 1) of the same fragment moved up one lane (DPP row_shr:1 - a DPP
// row is the 16 lanes of one k quarter, i.e. the 16 pixels of the fragment), lane 0 taking lane 15 of the fragment before it or,
// for the strip's first fragment, one extra "edge" load that fetches just the two pixels beside the strip; where the neighbour lies
// across an image edge (x = 0 / x = W - 1) the tap is zero padding and the lane is cleared.  12 + 6 loads per strip instead of 36:
// the kernel is bound by the texture addresser (~60 cycles per fragment-shaped load, header), not by VALU issue.
template <bool TAIL, int BK_THREADS, bool ROWPP, bool PROJ = false>
__global__ __launch_bounds__(BK_THREADS, BK_THREADS / 256) void bottleneck64_kernel(const BkParams p) {
    // PROJ && TAIL (res2a with res2b's branch2a appended): 21 images do not fit the LDS, so the three per-chunk filter sets are
    // STREAMED - chunk g of an output strip needs image g of branch2c, of the projection and of the next branch2a only: a ring of
    // two 3-image slots, the next chunk's images loaded into registers at the start of a chunk and written to the other slot at its
    // end, ONE workgroup barrier per chunk (the other forms have none: their waves never synchronise).  Every wave then runs the same
    // number of strips (a wave without a strip computes on out-of-range offsets: zeros in, nothing out).
    constexpr bool STREAM = TAIL && PROJ;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;            // column (pixel) / k quarter = row quarter of this lane

    // ---- weights -> LDS, once: image row rho holds source row perm_row(rho); 16-byte chunk s of a row sits at slot s ^ (rho & 7)
    {
        const int rho = (t >> 3) & 63, slot = t & 7, chunk = slot ^ (rho & 7), src = perm_row(rho);
#pragma unroll 1
        for (int im = 0; im < (STREAM ? 12 : 17) && t < 512; ++im) {
            const char* g = nullptr;
            if (im < 9)       g = p.w2b + ((long long)src * 576 + im * 64) * 2 + chunk * 16;
            else if (STREAM)  g = im == 9 ? p.w2c + ((long long)src * p.w2c_ld) * 2 + chunk * 16         // chunk 0 into ring slot 0
                                : im == 10 ? p.wproj + ((long long)src * p.w2c_ld) * 2 + chunk * 16
                                : p.w2a + ((long long)src * 256) * 2 + chunk * 16;
            else if (im < 13) g = p.w2c + ((long long)((im - 9) * 64 + src) * p.w2c_ld) * 2 + chunk * 16;
            else if (TAIL)    g = p.w2a + ((long long)src * 256 + (im - 13) * 64) * 2 + chunk * 16;
            else if (PROJ)    g = p.wproj + ((long long)((im - 13) * 64 + src) * p.w2c_ld) * 2 + chunk * 16;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (g) v = *reinterpret_cast<const uint4*>(g);
            *reinterpret_cast<uint4*>(lds + im * IMG + rho * 128 + slot * 16) = v;
        }
        float* bl = reinterpret_cast<float*>(lds + BIAS_OFF);
        if (t < 64) bl[t] = p.b2b[t];
        if (t < 256) bl[64 + t] = p.b2c[t];
        if (t < 64) bl[320 + t] = TAIL ? p.b2a[t] : 0.f;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.ain, 0, p.M * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(PROJ ? p.pin : p.xin), 0, PROJ ? p.M * 128 : p.M * 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.xout, 0, p.M * 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.h1out ? p.h1out : p.xout), 0, p.h1out ? p.M * 128 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t n_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(TAIL ? p.aout : p.xout), 0, TAIL ? p.M * 128 : 0, 0x00020000);

    // A-operand (weight) fragment of image `im`, row fragment f, k half ks: lane (kq = q, row c) reads row 16 f + c
    const unsigned w_lane = (unsigned)(c * 128 + ((q ^ (c & 7)) << 4));
    char* const xl = lds + XST_OFF + wave * 2048;         // this wave's [16 pixels][128 B] transpose buffer (same swizzle)
    const unsigned xr_lane = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
#define BK_WFRAG(IMOFF, F, KS) (*reinterpret_cast<const uint4*>(lds + (IMOFF) + (F) * 2048 + (w_lane ^ ((KS) * 64u))))
    // where chunk G's images sit: resident (image G of the block) or, STREAM, in ring slot G & 1
#define BK_IMG_2C(G) (STREAM ? RING_OFF + ((G) & 1) * 3 * IMG : W2C_OFF + (G) * IMG)
#define BK_IMG_PJ(G) (STREAM ? RING_OFF + ((G) & 1) * 3 * IMG + IMG : W2A_OFF + (G) * IMG)
#define BK_IMG_2A(G) (STREAM ? RING_OFF + ((G) & 1) * 3 * IMG + 2 * IMG : W2A_OFF + (G) * IMG)
    // STREAM: this thread's piece of the next chunk's three images (row rho of an image <- filter row perm_row(rho), slot swizzled)
    const int st_rho = (t >> 3) & 63, st_slot = t & 7, st_src = perm_row(st_rho);
    const unsigned st_loff = (unsigned)(st_rho * 128 + st_slot * 16);
    const unsigned st_vc = STREAM ? (unsigned)(st_src * p.w2c_ld * 2 + (st_slot ^ (st_rho & 7)) * 16) : 0u;       // row st_src of a [64][w2c_ld] block
    const unsigned st_va = (unsigned)(st_src * 512 + (st_slot ^ (st_rho & 7)) * 16);                              // row st_src of [64][256]
    const __amdgpu_buffer_rsrc_t wc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2c, 0, STREAM ? (255 * p.w2c_ld + 64) * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(STREAM ? p.wproj : p.w2c), 0, STREAM ? (255 * p.w2c_ld + 64) * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wa_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(STREAM ? p.w2a : p.w2c), 0, STREAM ? 64 * 256 * 2 : 0, 0x00020000);
    uint4 wreg[3];
    auto ring_load = [&](int gn) {
        const u32x4 a_ = __builtin_amdgcn_raw_buffer_load_b128(wc_rsrc, (int)st_vc, gn * 64 * p.w2c_ld * 2, 0);
        const u32x4 b_ = __builtin_amdgcn_raw_buffer_load_b128(wp_rsrc, (int)st_vc, gn * 64 * p.w2c_ld * 2, 0);
        const u32x4 c_ = __builtin_amdgcn_raw_buffer_load_b128(wa_rsrc, (int)st_va, gn * 128, 0);
        wreg[0] = make_uint4(a_.x, a_.y, a_.z, a_.w); wreg[1] = make_uint4(b_.x, b_.y, b_.z, b_.w); wreg[2] = make_uint4(c_.x, c_.y, c_.z, c_.w);
    };
    auto ring_store = [&](int slot_r) {
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4*>(lds + RING_OFF + (slot_r * 3 + k) * IMG + st_loff) = wreg[k];
    };
    const float* bias_l = reinterpret_cast<const float*>(lds + BIAS_OFF);
    // bias of this lane's rows of fragment f (channels 32 (f >> 1) + 8 q + 4 (f & 1) + r, r = 0..3: one float4)
#define BK_BIAS(BASE, F) (*reinterpret_cast<const f32x4*>(bias_l + (BASE) + 32 * ((F) >> 1) + 8 * q + 4 * ((F) & 1)))
#define BK_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)

    const int cells = p.H * p.W;
    const int stride = (int)gridDim.x * (BK_THREADS / 64);
    // strips are dealt wave-major: the one or two extra strips of a launch go to the low wave indices of every workgroup
    // ---- geometry of a strip for this lane: its two pixels, the byte offsets of their taps, the shortcut / output offsets
    struct Geo {
        unsigned pbase[2];          // byte offset of the pixel in a 64-channel tensor (+ this lane's 8-channel group)
        unsigned okmask[2];         // bit (kh * 3 + kw): the tap lies inside the image
        unsigned xoff[2];           // byte offset in a 256-channel tensor (+ this lane's 8-channel group), out of range past M
        unsigned aoff[2];           // byte offset in the 64-channel output
        unsigned xs[2][2];          // byte offset in the 256-channel output of pixel 16 u + 8 j + (lane >> 3), + 16 (lane & 7)
    };
    auto geometry = [&](int strip, Geo& g) {
        const int p0 = strip * 32;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = p0 + 16 * u + c;
            const bool live = strip < p.nstrips && pix < p.M;
            const int pc = live ? pix : 0;
            int b, rem, y, x;
            divmod24(pc, cells, p.inv_cells, b, rem);
            divmod24(rem, p.W, p.inv_w, y, x);
            unsigned m = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if (live && (unsigned)(y + kh - 1) < (unsigned)p.H && (unsigned)(x + kw - 1) < (unsigned)p.W) m |= 1u << (kh * 3 + kw);
            g.okmask[u] = (p.dbg & 1) ? 0u : m;
            g.pbase[u] = (unsigned)pc * 128u + (unsigned)q * 16u;
            g.xoff[u] = live ? (unsigned)pix * 512u + (unsigned)q * 16u : BK_OOB;
            g.aoff[u] = (live && !(p.dbg & 8)) ? (unsigned)pix * 128u + (unsigned)q * 16u : BK_OOB;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int px = p0 + 16 * u + 8 * j + (lane >> 3);
                g.xs[u][j] = (strip < p.nstrips && px < p.M) ? (unsigned)px * 512u + (unsigned)(lane & 7) * 16u : BK_OOB;
            }
        }
    };
    // a tap row in registers: the centre tap + the edge pixels
    struct Row { uint4 v[1][2][2]; uint4 e[2]; };       // v[0][k half][u]; e[k half]: lane 0 = pixel left of the strip, lane 15 = right of it
#define BK_LOAD_ROW(KH, G, DST)                                                                      \
    {                                                                                                \
        const int delta_ = ((KH) - 1) * p.W * 128;                                                   \
        _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                           \
            const unsigned off_ = ((G.okmask[u_] >> ((KH) * 3 + 1)) & 1u) ? G.pbase[u_] + (unsigned)delta_ : BK_OOB; \
            _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                    \
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off_, ks_ * 64, 0); \
                DST.v[0][ks_][u_] = make_uint4(v_.x, v_.y, v_.z, v_.w);                              \
            }                                                                                        \
        }                                                                                            \
        const unsigned eoff_ = c == 0 ? (((G.okmask[0] >> ((KH) * 3 + 0)) & 1u) ? G.pbase[0] + (unsigned)(delta_ - 128) : BK_OOB)  \
                             : c == 15 ? (((G.okmask[1] >> ((KH) * 3 + 2)) & 1u) ? G.pbase[1] + (unsigned)(delta_ + 128) : BK_OOB) : BK_OOB; \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                        \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)eoff_, ks_ * 64, 0); \
            DST.e[ks_] = make_uint4(v_.x, v_.y, v_.z, v_.w);                                         \
        }                                                                                            \
    }
    // lane c <- lane c - 1 of `cur` within each 16-lane row; lane 0 <- `first` as it stands (left) / lane c <- lane c + 1, lane 15 <- `last`
#define BK_DPP(OLD, SRC, CTRL) (unsigned)__builtin_amdgcn_update_dpp((int)(OLD), (int)(SRC), (CTRL), 0xf, 0xf, false)
    auto shift_right = [&](const uint4& first, const uint4& cur) {     // result lane c = cur lane c - 1; lane 0 = first lane 0
        return make_uint4(BK_DPP(first.x, cur.x, 0x111), BK_DPP(first.y, cur.y, 0x111), BK_DPP(first.z, cur.z, 0x111), BK_DPP(first.w, cur.w, 0x111));
    };
    auto shift_left = [&](const uint4& last, const uint4& cur) {       // result lane c = cur lane c + 1; lane 15 = last lane 15
        return make_uint4(BK_DPP(last.x, cur.x, 0x101), BK_DPP(last.y, cur.y, 0x101), BK_DPP(last.z, cur.z, 0x101), BK_DPP(last.w, cur.w, 0x101));
    };
    auto rotate = [&](const uint4& v, int ctrl) {                      // 0x121: lane 0 <- lane 15; 0x12f: lane 15 <- lane 0
        return ctrl == 0x121 ? make_uint4(BK_DPP(0, v.x, 0x121), BK_DPP(0, v.y, 0x121), BK_DPP(0, v.z, 0x121), BK_DPP(0, v.w, 0x121))
                             : make_uint4(BK_DPP(0, v.x, 0x12f), BK_DPP(0, v.y, 0x12f), BK_DPP(0, v.z, 0x12f), BK_DPP(0, v.w, 0x12f));
    };
    auto keep_if = [&](const uint4& v, unsigned ok) { return ok ? v : make_uint4(0u, 0u, 0u, 0u); };
#define BK_MUL_TAP(KH, KW, OPND)                                                                     \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                          \
            _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) {                                       \
                const uint4 wf_ = BK_WFRAG(W2B_OFF + ((KH) * 3 + (KW)) * IMG, f_, ks_);              \
                BK_MFMA(acc1[f_][0], wf_, OPND[ks_][0]);                                             \
                BK_MFMA(acc1[f_][1], wf_, OPND[ks_][1]);                                             \
            }
#define BK_MUL_ROW(KH, SRC, G)                                                                       \
    {                                                                                                \
        uint4 nb_[2][2];                                                                             \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                        \
            nb_[ks_][0] = keep_if(shift_right(SRC.e[ks_], SRC.v[0][ks_][0]), (G.okmask[0] >> ((KH) * 3 + 0)) & 1u);                        \
            nb_[ks_][1] = keep_if(shift_right(rotate(SRC.v[0][ks_][0], 0x121), SRC.v[0][ks_][1]), (G.okmask[1] >> ((KH) * 3 + 0)) & 1u);   \
        }                                                                                            \
        BK_MUL_TAP(KH, 0, nb_)                                                                       \
        BK_MUL_TAP(KH, 1, SRC.v[0])                                                                  \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                        \
            nb_[ks_][0] = keep_if(shift_left(rotate(SRC.v[0][ks_][1], 0x12f), SRC.v[0][ks_][0]), (G.okmask[0] >> ((KH) * 3 + 2)) & 1u);    \
            nb_[ks_][1] = keep_if(shift_left(SRC.e[ks_], SRC.v[0][ks_][1]), (G.okmask[1] >> ((KH) * 3 + 2)) & 1u);                         \
        }                                                                                            \
        BK_MUL_TAP(KH, 2, nb_)                                                                       \
    }
#define BK_LOAD_RES(GI, G, DST)                                                                      \
    if (!PROJ)                                                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                 \
        _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                           \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)G.xoff[u_], (GI) * 128 + s_ * 64, 0); \
            DST[s_][u_] = make_uint4(v_.x, v_.y, v_.z, v_.w);                                        \
        }
    // Software pipeline over the strips of this wave.  vmcnt retires in order, so a load can only be waited for together with
    // every store issued before it: the first tap row and the first shortcut chunk of the NEXT strip are therefore requested in
    // the middle of the current strip's output chunks (before the stores of chunks 2 and 3), not after them - otherwise every
    // strip begins by waiting for its predecessor's 20 stores to be acknowledged (measured: memory and compute times added up).
    Geo gc, gn;
    Row row0;                       // tap row 0 of the current strip (requested during the previous strip)
    uint4 res0[2][2];               // shortcut chunk 0 of the current strip (likewise)
    int strip = wave * (int)gridDim.x + (int)blockIdx.x;
    geometry(strip, gc);
    BK_LOAD_ROW(0, gc, row0)
    BK_LOAD_RES(0, gc, res0)
#pragma unroll 1
    for (; STREAM ? (strip - wave * (int)gridDim.x) < p.nstrips : strip < p.nstrips; strip += stride) {      // STREAM: as long as wave 0 has a strip
        // ---- G1: branch2b.  acc1[f][u] = sum over taps, k of W2b[chan(f)][tap][k] * Ain[pixel(u) + tap][k]
        f32x4 acc1[4][2];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4 bv = BK_BIAS(0, f);
            acc1[f][0] = bv; acc1[f][1] = bv;
        }
        uint4 pfrag[2][2];          // PROJ: [k half][u] the block input at this strip's pixels (B operand of the projection shortcut)
        if (PROJ) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)gc.pbase[u], ks * 64, 0);
                    pfrag[ks][u] = make_uint4(v.x, v.y, v.z, v.w);
                }
        }
        {
            Row row1;
            BK_LOAD_ROW(1, gc, row1)
            __builtin_amdgcn_sched_barrier(0);
            BK_MUL_ROW(0, row0, gc)
            __builtin_amdgcn_sched_barrier(0);
            BK_LOAD_ROW(2, gc, row0)
            __builtin_amdgcn_sched_barrier(0);
            BK_MUL_ROW(1, row1, gc)
            __builtin_amdgcn_sched_barrier(0);
            BK_MUL_ROW(2, row0, gc)
            __builtin_amdgcn_sched_barrier(0);
        }
        // ReLU + bf16: the B operand of G2, k half s = fragments 2 s and 2 s + 1
        uint4 h1[2][2];             // [s][u]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 lo = acc1[2 * s][u], hi = acc1[2 * s + 1][u];
                h1[s][u] = make_uint4(pack2(relu(lo[0]), relu(lo[1])), pack2(relu(lo[2]), relu(lo[3])),
                                      pack2(relu(hi[0]), relu(hi[1])), pack2(relu(hi[2]), relu(hi[3])));
            }
        if (p.h1out) {              // uniform: the training forward keeps branch2b's activation (lane: 8 consecutive channels of a pixel)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const u32x4 ov = {h1[s][u].x, h1[s][u].y, h1[s][u].z, h1[s][u].w};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, h_rsrc, (int)gc.aoff[u], s * 64, 0);
                    BK_STORE_GUARD(ov)
                }
        }
        // ---- G2 (branch2c + shortcut + ReLU) in four 64-channel chunks, each feeding G3 (next branch2a) as one k chunk
        f32x4 acc3[4][2];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4 bv = BK_BIAS(320, f);
            acc3[f][0] = bv; acc3[f][1] = bv;
        }
        uint4 resq[2][2][2];        // [parity][s][u]: shortcut, channels 64 g + 32 s + 8 q .. + 8 of pixel (u, c)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) resq[0][s][u] = res0[s][u];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (ROWPP && g == 2) {  // the next strip's first loads go out ahead of this strip's last stores
                geometry(strip + stride, gn);
                BK_LOAD_ROW(0, gn, row0)
                BK_LOAD_RES(0, gn, res0)
            }
            if (g + 1 < 4) BK_LOAD_RES(g + 1, gc, resq[(g + 1) & 1])
            if (STREAM) ring_load((g + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
            uint4 (&res)[2][2] = resq[g & 1];
            f32x4 acc2[4][2];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const f32x4 bv = BK_BIAS(64 + 64 * g, f);
                acc2[f][0] = bv; acc2[f][1] = bv;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const uint4 wf = BK_WFRAG(BK_IMG_2C(g), f, ks);
                    BK_MFMA(acc2[f][0], wf, h1[ks][0]);
                    BK_MFMA(acc2[f][1], wf, h1[ks][1]);
                }
            if (PROJ) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const uint4 wf = BK_WFRAG(BK_IMG_PJ(g), f, ks);
                        BK_MFMA(acc2[f][0], wf, pfrag[ks][0]);
                        BK_MFMA(acc2[f][1], wf, pfrag[ks][1]);
                    }
            }
            uint4 xo[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f32x4 lo = acc2[2 * s][u], hi = acc2[2 * s + 1][u];
                    const uint4 r = PROJ ? make_uint4(0u, 0u, 0u, 0u) : res[s][u];
                    const uint4 o = make_uint4(pack2(relu(lo[0] + bf_lo(r.x)), relu(lo[1] + bf_hi(r.x))),
                                               pack2(relu(lo[2] + bf_lo(r.y)), relu(lo[3] + bf_hi(r.y))),
                                               pack2(relu(hi[0] + bf_lo(r.z)), relu(hi[1] + bf_hi(r.z))),
                                               pack2(relu(hi[2] + bf_lo(r.w)), relu(hi[3] + bf_hi(r.w))));
                    xo[s][u] = o;
                    *reinterpret_cast<uint4*>(xl + (w_lane ^ (s * 64u))) = o;           // [pixel c][128 B], slot (4 s + q) ^ (c & 7)
                }
                {
                    // the fragment's 16 pixels x 128 B leave as two stores of 8 whole lines each (8 lanes per pixel) instead of two
                    // stores of 16 half lines: the texture addresser spends ~60 cycles on the latter (header)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint4 o = *reinterpret_cast<const uint4*>(xl + j * 1024 + xr_lane);
                        const u32x4 ov = {o.x, o.y, o.z, o.w};
                        // aux 2 = nt: a streaming store.  x_out (273 MB at batch 8) is many times the L2 and is read back by the next
                        // launch only; without the hint its lines push the tap rows and shortcut lines out of the L2 on their way to
                        // memory.  Launch alone 0.122 / 0.147 / 0.133 -> 0.102 / 0.140 / 0.123 ms, the step -0.4 % (tools/ab_engine.py);
                        // the same hint on the shortcut loads changed nothing.
                        __builtin_amdgcn_raw_buffer_store_b128(ov, o_rsrc, (int)gc.xs[u][j], g * 128, 2);
                        BK_STORE_GUARD(ov)
                    }
                }
            }
            if (TAIL) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const uint4 wf = BK_WFRAG(BK_IMG_2A(g), f, ks);
                        BK_MFMA(acc3[f][0], wf, xo[ks][0]);
                        BK_MFMA(acc3[f][1], wf, xo[ks][1]);
                    }
            }
            if (STREAM) {           // the next chunk's images are in registers: into the slot the chunk before read, then everyone meets
                ring_store((g & 1) ^ 1);
                __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TAIL) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 lo = acc3[2 * s][u], hi = acc3[2 * s + 1][u];
                    const u32x4 ov = {pack2(relu(lo[0]), relu(lo[1])), pack2(relu(lo[2]), relu(lo[3])),
                                      pack2(relu(hi[0]), relu(hi[1])), pack2(relu(hi[2]), relu(hi[3]))};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, n_rsrc, (int)gc.aoff[u], s * 64, 0);
                    BK_STORE_GUARD(ov)
                }
        }
        if (!ROWPP) {               // A/B: the unpipelined order (next strip's loads behind this strip's stores)
            geometry(strip + stride, gn);
            BK_LOAD_ROW(0, gn, row0)
            BK_LOAD_RES(0, gn, res0)
        }
        gc = gn;
    }
#undef BK_LOAD_RES
#undef BK_MUL_ROW
#undef BK_MUL_TAP
#undef BK_DPP
#undef BK_LOAD_ROW
#undef BK_MFMA
#undef BK_BIAS
#undef BK_IMG_2C
#undef BK_IMG_PJ
#undef BK_IMG_2A
#undef BK_WFRAG
}

constexpr int BK_NT = 512;          // (the 768-thread instances left in round 4: slower with the shifted taps, and the LDS now holds the store buffers)
bool rtn_bneck_rowpp(int nt) {                           // cross-strip software pipeline (RTN_BNECK_ROWPP=0: off, for the A/B)
    return rtn_env_int("RTN_BNECK_ROWPP", 1) != 0;
}

}  // namespace

extern "C" int rtn_bottleneck64_fwd(rtn_handle_t h, const rtn_bottleneck_desc_t* d) {
    if (!h) return RTN_EINVAL;
    rtn_env_sync();
    if (!d) return rtn_fail(h, RTN_EINVAL, "bottleneck64: null descriptor");
    if (d->dtype != RTN_BF16) return rtn_fail(h, RTN_EINVAL, "bottleneck64: bf16 only");
    if (d->mid != 64) return rtn_fail(h, RTN_EINVAL, "bottleneck64: the fused block exists for 64 bottleneck channels (res2), got %d", d->mid);
    if (d->batch < 1 || d->H < 1 || d->W < 1) return rtn_fail(h, RTN_EINVAL, "bottleneck64: empty extent");
    const long long M = (long long)d->batch * d->H * d->W;
    if (M >= (1ll << 22)) return rtn_fail(h, RTN_EINVAL, "bottleneck64: %lld pixels exceed the 4 Mi-pixel (2 GiB tensor) range", M);
    const bool proj = d->wproj != nullptr;
    const void* need[] = {d->a_in, proj ? d->p_in : d->x_in, d->x_out, d->w2b, d->w2c, d->b2b, d->b2c};
    for (const void* q : need)
        if (!q || ((uintptr_t)q & 15)) return rtn_fail(h, RTN_EINVAL, "bottleneck64: null / misaligned pointer");
    const bool tail = d->a_out != nullptr;
    const int w2c_ld = d->w2c_ld > 0 ? d->w2c_ld : 64;
    if (w2c_ld < 64 || w2c_ld % 8) return rtn_fail(h, RTN_EINVAL, "bottleneck64: w2c_ld %d", d->w2c_ld);
    if (proj && (((uintptr_t)d->wproj & 15) || d->p_in_elems < M * 64))
        return rtn_fail(h, RTN_EINVAL, "bottleneck64: the projection-shortcut form takes p_in [M][64] and an aligned wproj");
    if (tail && (!d->w2a || !d->b2a || ((uintptr_t)d->a_out & 15) || ((uintptr_t)d->w2a & 15) || ((uintptr_t)d->b2a & 15)))
        return rtn_fail(h, RTN_EINVAL, "bottleneck64: a_out needs aligned w2a / b2a");
    if (d->a_in_elems < M * 64 || (!proj && d->x_in_elems < M * 256) || d->x_out_elems < M * 256 || (tail && d->a_out_elems < M * 64))
        return rtn_fail(h, RTN_EBOUNDS, "bottleneck64: a tensor is smaller than batch x H x W x channels");
    if (d->h1_out && (((uintptr_t)d->h1_out & 15) || d->h1_out_elems < M * 64 || d->h1_out == d->a_in))
        return rtn_fail(h, RTN_EINVAL, "bottleneck64: h1_out must be an aligned [M][64] tensor other than a_in");
    {   // no output may alias an input or another output: strips of other waves read a_in / x_in / p_in taps while this one stores
        const void* ins[] = {d->a_in, proj ? d->p_in : d->x_in};
        const void* outs[] = {d->x_out, d->a_out, d->h1_out};
        for (int i = 0; i < 3; ++i) {
            if (!outs[i]) continue;
            for (const void* in : ins)
                if (outs[i] == in) return rtn_fail(h, RTN_EINVAL, "bottleneck64: an output may not alias an input (taps of neighbouring strips)");
            for (int j = i + 1; j < 3; ++j)
                if (outs[i] == outs[j]) return rtn_fail(h, RTN_EINVAL, "bottleneck64: two outputs share a buffer");
        }
    }
    BkParams p;
    memset(&p, 0, sizeof(p));
    p.ain = (const char*)d->a_in; p.xin = (const char*)d->x_in; p.xout = (char*)d->x_out; p.aout = (char*)d->a_out;
    p.h1out = (char*)d->h1_out;
    p.w2b = (const char*)d->w2b; p.w2c = (const char*)d->w2c; p.w2a = (const char*)d->w2a;
    p.pin = (const char*)d->p_in; p.wproj = (const char*)d->wproj; p.w2c_ld = w2c_ld;
    p.b2b = d->b2b; p.b2c = d->b2c; p.b2a = d->b2a;
    p.M = (int)M; p.H = d->H; p.W = d->W;
    p.nstrips = (int)((M + 31) / 32);
    p.inv_cells = 1.0f / (float)((long long)d->H * d->W);
    p.inv_w = 1.0f / (float)d->W;
    p.dbg = 0;                                         // (timing ablations of round 2: profiles/r2_v2_bottleneck_fused.txt)
    const int nt = BK_NT;
    int grid = h->num_cus > 0 ? h->num_cus : 256;
    const int wgs_needed = (p.nstrips + nt / 64 - 1) / (nt / 64);
    if (grid > wgs_needed) grid = wgs_needed;
    { const int gl = rtn_env_int("RTN_BNECK_GRID", 0); if (gl > 0 && gl < grid) grid = gl; }     // tests: several strips per wave on small inputs
    const bool rowpp = rtn_bneck_rowpp(nt);
#define RTN_BK_LAUNCH_S(T, RP, PJ)                                                                       \
    do {                                                                                                 \
        static std::atomic<unsigned long long> attr_set{0ull};  /* one bit per device */                 \
        if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {                  \
            RTN_HIP(h, hipFuncSetAttribute((const void*)bottleneck64_kernel<T, 512, RP, PJ>,         \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, BK_LDS));         \
            attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);                      \
        }                                                                                                \
        hipLaunchKernelGGL((bottleneck64_kernel<T, 512, RP, PJ>), dim3((unsigned)grid), dim3(512), BK_LDS, h->stream, p); \
    } while (0)
    if (proj) {
        if (tail) RTN_BK_LAUNCH_S(true, true, true); else RTN_BK_LAUNCH_S(false, true, true);
    } else {
        if (rowpp) { if (tail) RTN_BK_LAUNCH_S(true, true, false); else RTN_BK_LAUNCH_S(false, true, false); }
        else       { if (tail) RTN_BK_LAUNCH_S(true, false, false); else RTN_BK_LAUNCH_S(false, false, false); }
    }
#undef RTN_BK_LAUNCH_S
    RTN_CHECK_LAUNCH(h, "bottleneck64_kernel");
    return RTN_OK;
}
