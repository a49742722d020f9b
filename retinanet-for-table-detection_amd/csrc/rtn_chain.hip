// rtn_chain.hip — the seam between two keras_resnet identity bottleneck blocks of the 128-channel stage as ONE kernel:
//   x_out = relu(conv1x1(h_in; w2c) + b2c + x_in)      this block's branch2c + BN + Add + ReLU   (mid -> 4 mid)
//   a_out = relu(conv1x1(x_out; w2a) + b2a)            the NEXT block's branch2a + BN + ReLU     (4 mid -> mid)
// (keras_resnet bottleneck_2d as instantiated by model/defineModel.py:376-380; the reference runs these as two Conv2D, two
// BatchNormalization, an Add and two ReLU ops.)  Unfused, x_out is written by one launch and read back by the next and both launches
// pay their own prologue, epilogue and tile quantisation; fused, a pixel's x_out leaves the registers once (it is still a tensor
// of the network: the block after next adds it) and never comes back.
//
// Same chaining as rtn_bottleneck.hip: both products are computed TRANSPOSED (weights = the MFMA's A operand, pixels = its
// columns) with the weight rows permuted so that the bf16-packed accumulators of the first product ARE the B operand of the second
// and are also what a 16-byte store wants (8 consecutive channels of a pixel).  What differs: the filters do not fit the LDS
// (2 x 128 KB; 2 x 512 KB at 256 channels), so they are STREAMED through it in chunks of 64 branch2c output channels = 64 K values of
// the next branch2a: chunk g needs rows [64 g, 64 g + 64) of w2c (all of its K) and columns [64 g, 64 g + 64) of w2a (all of its
// rows).  All waves of a workgroup walk the chunks in step (one barrier per chunk, LDS double buffer, the next chunk's filters
// loaded into registers at the start of a chunk and written to the other buffer at its end); a wave owns a strip of 16 PX pixels,
// keeps its h_in fragments and the a_out accumulators in registers for the whole pass and sees every filter chunk once.
#include "rtn_internal.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr unsigned CH_OOB = 0xFFFF0000u;       // beyond every descriptor (< 2 GiB) even with the largest uniform offset added
constexpr int IMG = 64 * 128;                  // one filter image: 64 rows x 128 B (64 K values), 16-byte slots XOR-swizzled by row & 7

struct ChParams {
    const char* hin;        // [M][mid]    bf16: this block's branch2b output (after BN + ReLU)
    const char* xin;        // [M][4 mid]  bf16: the block input (identity shortcut)
    char* xout;             // [M][4 mid]  bf16
    char* aout;             // [M][next]   bf16: the next block's branch2a output
    const char* w2c;        // [4 mid][mid] bf16, BN folded
    const char* w2a;        // [next][4 mid]
    const float* b2c;       // [4 mid], [next] f32 (folded BN shifts)
    const float* b2a;
    int M, nstrips, spread;
    int dbg;                // timing ablation only (RTN_CHAIN_DBG): 1 no shortcut loads, 2 no x_out stores, 4 no filter staging, 8 no barrier, 16 no h_in loads, 32 no a_out stores
};

__device__ __forceinline__ int perm_row(int rho) {     // MFMA row (16 f + 4 q + r) -> channel 32 (f >> 1) + 8 q + 4 (f & 1) + r
    const int f = rho >> 4, q = (rho >> 2) & 3, r = rho & 3;
    return 32 * (f >> 1) + 8 * q + 4 * (f & 1) + r;
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float relu(float v) { return v > 0.f ? v : 0.f; }
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// the store-data guard of rtn_bottleneck.hip (profiles/r3_store_hazard_isa.txt): the data registers of a 16-byte buffer store with
// an SGPR offset stay live and unmodified for four wait states whatever the compiler schedules behind it
#define CH_STORE_GUARD(V) asm volatile("s_nop 3" :: "v"(V.x), "v"(V.y), "v"(V.z), "v"(V.w));

// CM = mid / 64 (K images of branch2c), NCH = 4 mid / 64 (chunks), N3 = next / 64 (row images of the next branch2a),
// PX = 16-pixel column fragments per wave, eight waves; DEPTH = chunks the shortcut fragments are requested ahead (the kernel moves
// every tensor once, so what bounds it is bytes in flight: 8 waves x DEPTH x PX KB of shortcut per CU)
template <int CM, int NCH, int N3, int PX, int DEPTH>
__global__ __launch_bounds__(512) void chain1x1_kernel(const ChParams p) {
    constexpr int CMID = CM * 64, COUT = NCH * 64, CNEXT = N3 * 64, NI = CM + N3, BUF = NI * IMG, BIAS_OFF = 2 * BUF;
    static_assert(NCH % 2 == 0, "the chunk loop is unrolled by the two LDS buffers");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;            // column (pixel) / k quarter = row quarter of this lane
#ifdef RTN_CHAIN_ABL
    const int dbg = p.dbg;                             // timing ablations: a build with -DRTN_CHAIN_ABL reads RTN_CHAIN_DBG
#else
    constexpr int dbg = 0;
#endif

    {
        float* bl = reinterpret_cast<float*>(lds + BIAS_OFF);
        for (int i = t; i < COUT; i += 512) bl[i] = p.b2c[i];
        for (int i = t; i < CNEXT; i += 512) bl[COUT + i] = p.b2a[i];
    }
    const __amdgpu_buffer_rsrc_t wc_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2c, 0, COUT * CMID * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t wa_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2a, 0, CNEXT * COUT * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.hin, 0, p.M * CMID * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.xin, 0, p.M * COUT * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.xout, 0, p.M * COUT * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t n_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.aout, 0, p.M * CNEXT * 2, 0x00020000);

    // ---- filter staging: thread -> (image row rho, 16-byte slot); image row rho holds source row perm_row(rho), chunk s of a row
    // sits at slot s ^ (rho & 7).  One 16-byte load per image per thread.
    const int rho = (t >> 3) & 63, slot = t & 7, kchunk = slot ^ (rho & 7), src = perm_row(rho);
    const unsigned wv2 = (unsigned)((src * CMID + kchunk * 8) * 2);      // + chunk g: g * 64 rows; + K image im: im * 128 B
    const unsigned wv3 = (unsigned)((src * COUT + kchunk * 8) * 2);      // + row image j: j * 64 rows; + chunk g: g * 128 B
    const unsigned wl = (unsigned)(rho * 128 + slot * 16);
    uint4 wreg[NI];
#define CH_WLOAD(G)                                                                                                            \
    {                                                                                                                          \
        _Pragma("unroll") for (int im = 0; im < CM; ++im)                                                                      \
            wreg[im] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wc_rsrc, (int)wv2, (G) * (64 * CMID * 2) + im * 128, 0)); \
        _Pragma("unroll") for (int j = 0; j < N3; ++j)                                                                         \
            wreg[CM + j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wa_rsrc, (int)wv3, j * (64 * COUT * 2) + (G) * 128, 0)); \
    }
#define CH_WSTORE(B)                                                                                                           \
    {                                                                                                                          \
        _Pragma("unroll") for (int im = 0; im < NI; ++im)                                                                      \
            *reinterpret_cast<uint4*>(lds + (B) * BUF + im * IMG + wl) = wreg[im];                                             \
    }
    CH_WLOAD(0)
    CH_WSTORE(0)
    __syncthreads();

    // A-operand (filter) fragment of an image, row fragment f, k half ks: lane (kq = q, row c) reads row 16 f + c
    const unsigned w_lane = (unsigned)(c * 128 + ((q ^ (c & 7)) << 4));
#define CH_WFRAG(IMOFF, F, KS) (*reinterpret_cast<const uint4*>(lds + (IMOFF) + (F) * 2048 + (w_lane ^ ((KS) * 64u))))
    const float* bias_l = reinterpret_cast<const float*>(lds + BIAS_OFF);
    // bias of this lane's rows of fragment f of a 64-channel group at BASE (channels 32 (f >> 1) + 8 q + 4 (f & 1) + r: one float4)
#define CH_BIAS(BASE, F) (*reinterpret_cast<const f32x4*>(bias_l + (BASE) + 32 * ((F) >> 1) + 8 * q + 4 * ((F) & 1)))
#define CH_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)

    // How strips are dealt.  spread: wave-major inside a pass (strip = pass * 8 G + wave * G + workgroup) - the leftover strips of
    // the last pass (79 of 4,175 at stage 3 of the bench, 52 of 2,100 at stage 4) land on wave 0 of as many workgroups, which then
    // run a pass bound by the filter stream's latency: the shortest launch.  packed: strip = (pass * G + workgroup) * 8 + wave - the
    // leftover goes to all waves of a few workgroups, which run one more full pass while the other CUs are free for the launches of
    // another stream: the fewest CU-seconds.
    const int G = (int)gridDim.x;
#pragma unroll 1
    for (int pass = 0; (p.spread ? (pass * 8) * G + (int)blockIdx.x : (pass * G + (int)blockIdx.x) * 8) < p.nstrips; ++pass) {
        const int pix0 = (p.spread ? (pass * 8 + wave) * G + (int)blockIdx.x : (pass * G + (int)blockIdx.x) * 8 + wave) * (16 * PX);
        const bool active = pix0 < p.M;                         // wave-uniform: a wave without pixels still stages filters
        unsigned hoff[PX], xoff[PX], aoff[PX];
#pragma unroll
        for (int u = 0; u < PX; ++u) {
            const int pix = pix0 + u * 16 + c;
            const bool ok = pix < p.M;
            hoff[u] = ok ? (unsigned)(pix * (CMID * 2) + q * 16) : CH_OOB;
            xoff[u] = ok ? (unsigned)(pix * (COUT * 2) + q * 16) : CH_OOB;
            aoff[u] = ok ? (unsigned)(pix * (CNEXT * 2) + q * 16) : CH_OOB;
        }
        uint4 hb[2 * CM][PX];                                   // B operand of branch2c: k step ks = channels [32 ks + 8 q, + 8) of pixel c
        f32x4 acc3[4 * N3][PX];                                 // the next branch2a's accumulators
        uint4 res[DEPTH + 1][2][PX];                            // shortcut fragments of chunks g .. g + DEPTH (set = chunk % (DEPTH + 1))
        if (active) {
#pragma unroll
            for (int ks = 0; ks < 2 * CM; ++ks)
#pragma unroll
                for (int u = 0; u < PX; ++u)
                    hb[ks][u] = (dbg & 16) ? make_uint4(0x3f803f80u, 0u, 0u, 0u) : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, (int)hoff[u], ks * 64, 0));
#pragma unroll
            for (int g = 0; g < DEPTH; ++g)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int u = 0; u < PX; ++u)
                        res[g][s][u] = (dbg & 1) ? make_uint4(0u, 0u, 0u, 0u) : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)xoff[u], g * 128 + s * 64, 2));
#pragma unroll
            for (int f = 0; f < 4 * N3; ++f)
#pragma unroll
                for (int u = 0; u < PX; ++u) acc3[f][u] = CH_BIAS(COUT + 64 * (f >> 2), f & 3);
        }

        // one chunk: 64 channels of x_out, then their contribution to every a_out channel.  Fully unrolled: the register sets of
        // the shortcut and the LDS buffer are compile-time.
#pragma unroll
        for (int g = 0; g < NCH; ++g) {
            const int cur = g & 1, gn = (g + 1 == NCH) ? 0 : g + 1;
            if (!(dbg & 4)) CH_WLOAD(gn)
            if (active) {
                if (g + DEPTH < NCH && !(dbg & 1)) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int u = 0; u < PX; ++u)
                            res[(g + DEPTH) % (DEPTH + 1)][s][u] =
                                __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)xoff[u], (g + DEPTH) * 128 + s * 64, 2)     /* nt: the shortcut's last reader */);
                }
                f32x4 acc2[4][PX];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const f32x4 bv = CH_BIAS(g * 64, f);
#pragma unroll
                    for (int u = 0; u < PX; ++u) acc2[f][u] = bv;
                }
#pragma unroll
                for (int ks = 0; ks < 2 * CM; ++ks)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const uint4 a = CH_WFRAG(cur * BUF + (ks >> 1) * IMG, f, ks & 1);
#pragma unroll
                        for (int u = 0; u < PX; ++u) CH_MFMA(acc2[f][u], a, hb[ks][u]);
                    }
                uint4 xo[2][PX];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int u = 0; u < PX; ++u) {
                        const f32x4 lo = acc2[2 * s][u], hi = acc2[2 * s + 1][u];
                        const uint4 rv = res[g % (DEPTH + 1)][s][u];
                        const u32x4 ov = {pack2(relu(lo[0] + bf_lo(rv.x)), relu(lo[1] + bf_hi(rv.x))),
                                          pack2(relu(lo[2] + bf_lo(rv.y)), relu(lo[3] + bf_hi(rv.y))),
                                          pack2(relu(hi[0] + bf_lo(rv.z)), relu(hi[1] + bf_hi(rv.z))),
                                          pack2(relu(hi[2] + bf_lo(rv.w)), relu(hi[3] + bf_hi(rv.w)))};
                        if (!(dbg & 2)) __builtin_amdgcn_raw_buffer_store_b128(ov, o_rsrc, (int)xoff[u], g * 128 + s * 64, 0);
                        CH_STORE_GUARD(ov)
                        xo[s][u] = __builtin_bit_cast(uint4, ov);
                    }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int f = 0; f < 4 * N3; ++f) {
                        const uint4 a = CH_WFRAG(cur * BUF + (CM + (f >> 2)) * IMG, f & 3, s);
#pragma unroll
                        for (int u = 0; u < PX; ++u) CH_MFMA(acc3[f][u], a, xo[s][u]);
                    }
            }
            if (!(dbg & 4)) CH_WSTORE(cur ^ 1)
            if (!(dbg & 8)) __syncthreads();
        }
        if (active) {
#pragma unroll
            for (int s = 0; s < 2 * N3; ++s)
#pragma unroll
                for (int u = 0; u < PX; ++u) {
                    const f32x4 lo = acc3[2 * s][u], hi = acc3[2 * s + 1][u];
                    const u32x4 ov = {pack2(relu(lo[0]), relu(lo[1])), pack2(relu(lo[2]), relu(lo[3])),
                                      pack2(relu(hi[0]), relu(hi[1])), pack2(relu(hi[2]), relu(hi[3]))};
                    if (!(dbg & 32)) __builtin_amdgcn_raw_buffer_store_b128(ov, n_rsrc, (int)aoff[u], s * 64, 0);
                    CH_STORE_GUARD(ov)
                }
        }
    }
#undef CH_MFMA
#undef CH_BIAS
#undef CH_WFRAG
#undef CH_WSTORE
#undef CH_WLOAD
}

template <int CM, int NCH, int N3, int PX, int DEPTH>
int chain_launch(rtn_handle_t h, ChParams& p, int grid_limit) {
    constexpr int LDS = 2 * (CM + N3) * IMG + (NCH * 64 + N3 * 64) * 4;
    const long long nstrips = ((long long)p.M + 16 * PX - 1) / (16 * PX);
    p.nstrips = (int)nstrips;
    int grid = h->num_cus > 0 ? h->num_cus : 256;
    if (grid > (p.nstrips + 7) / 8) grid = (p.nstrips + 7) / 8;
    if (grid_limit > 0 && grid > grid_limit) grid = grid_limit;
    static std::atomic<unsigned long long> attr_set{0ull};      // one bit per device
    if (!((attr_set.load(std::memory_order_relaxed) >> (h->device & 63)) & 1ull)) {
        RTN_HIP(h, hipFuncSetAttribute((const void*)chain1x1_kernel<CM, NCH, N3, PX, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set.fetch_or(1ull << (h->device & 63), std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((chain1x1_kernel<CM, NCH, N3, PX, DEPTH>), dim3((unsigned)grid), dim3(512), LDS, h->stream, p);
    RTN_CHECK_LAUNCH(h, "chain1x1_kernel");
    return RTN_OK;
}

}  // namespace

extern "C" int rtn_chain1x1_supported(int mid, int out, int next) {
    return mid == 128 && out == 512 && next == 128;    // (a 256 -> 1024 -> 256 instance was built and measured no faster than its two launches: profiles/r4_seam_kernel.txt)
}

extern "C" int rtn_chain1x1_fwd(rtn_handle_t h, const rtn_chain_desc_t* d) {
    if (!h) return RTN_EINVAL;
    rtn_env_sync();
    if (!d) return rtn_fail(h, RTN_EINVAL, "chain1x1: null descriptor");
    if (d->dtype != RTN_BF16) return rtn_fail(h, RTN_EINVAL, "chain1x1: bf16 only");
    if (!rtn_chain1x1_supported(d->mid, d->out, d->next))
        return rtn_fail(h, RTN_EINVAL, "chain1x1: built for 128 -> 512 -> 128 channels, got %d -> %d -> %d", d->mid, d->out, d->next);
    const long long M = d->pixels;
    if (M < 1) return rtn_fail(h, RTN_EINVAL, "chain1x1: empty extent");
    if (M * d->out * 2 >= (1ll << 31)) return rtn_fail(h, RTN_EINVAL, "chain1x1: %lld pixels x %d channels exceed the 2 GiB tensor range", M, d->out);
    const void* need[] = {d->h_in, d->x_in, d->x_out, d->a_out, d->w2c, d->w2a, d->b2c, d->b2a};
    for (const void* q : need)
        if (!q || ((uintptr_t)q & 15)) return rtn_fail(h, RTN_EINVAL, "chain1x1: null / misaligned pointer");
    if (d->h_in_elems < M * d->mid || d->x_in_elems < M * d->out || d->x_out_elems < M * d->out || d->a_out_elems < M * d->next)
        return rtn_fail(h, RTN_EBOUNDS, "chain1x1: a tensor is smaller than pixels x channels");
    if (d->a_out == d->h_in || d->a_out == d->x_in || d->a_out == d->x_out || d->x_out == d->h_in)
        return rtn_fail(h, RTN_EINVAL, "chain1x1: outputs may not share a buffer with h_in or each other (x_out may be x_in: every pixel is read and written by one wave)");
    ChParams p;
    memset(&p, 0, sizeof(p));
    p.hin = (const char*)d->h_in; p.xin = (const char*)d->x_in; p.xout = (char*)d->x_out; p.aout = (char*)d->a_out;
    p.w2c = (const char*)d->w2c; p.w2a = (const char*)d->w2a; p.b2c = d->b2c; p.b2a = d->b2a;
    p.M = (int)M;
    p.dbg = rtn_env_int("RTN_CHAIN_DBG", 0);
    p.spread = rtn_env_int("RTN_CHAIN_SPREAD", 1) != 0;
    const int gl = rtn_env_int("RTN_CHAIN_GRID", 0);             // tests: several passes per workgroup on small inputs
    // (shortcut fragments two chunks ahead instead of one, and x_out as whole lines through an LDS transpose as in rtn_bottleneck.hip,
    // measured no faster: this kernel is not bound by bytes in flight or by the texture addresser)
    return chain_launch<2, 8, 2, 2, 1>(h, p, gl);
}
