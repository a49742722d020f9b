// rtn_internal.h — shared by the librtn.so translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "rtn.h"

struct rtn_ctx {
    int device;
    hipStream_t stream;
    void* zero_page;        // 256 B of zeros on the device: source for out-of-image taps
    int num_cus;
    int last_conv_streamk;        // rtn_debug_last_conv_streamk: workgroups of the last conv launch if it ran in stream-K form, else 0
    int last_conv_tile;           // rtn_debug_last_conv_tile: (tile rows << 16) | tile columns of the last conv launch
    int last_conv_impl;
         // kernel generation of the last conv launch on this handle (rtn_debug_last_conv_impl)
    int last_wgrad_impl;          // rtn_debug_last_wgrad_impl: 2 = 256x256 LDS-DMA, 3 = 128x128 LDS-DMA, 4 = rtn_wgrad_win.hip, 0 = register-staged
    char err[512];
};

inline int rtn_fail(rtn_ctx* h, int code, const char* fmt, ...) {
    if (h) {
        va_list ap; va_start(ap, fmt);
        vsnprintf(h->err, sizeof(h->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define RTN_HIP(h, call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return rtn_fail((h), RTN_EHIP, "%s failed: %s (%s:%d)", #call,                 \
                            hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

#define RTN_CHECK_LAUNCH(h, name)                                                          \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            return rtn_fail((h), RTN_EHIP, "launch of %s failed: %s", name,                \
                            hipGetErrorString(e_));                                        \
    } while (0)

// rtn_conv_halo8.hip: persistent 8-phase kernel for the stride-1 3x3 layers with 129..256 output channels (head towers, P3-P5,
// res4 branch2b).  RTN_OK = launched, 1 = not a layer this kernel takes, < 0 = error.
// integer environment knob (name must be a string literal): cached per thread, re-read when the environment changed, so one process
// can still A/B kernel variants (tools/ab_conv.py).  rtn_env_sync() revalidates the cache: call it at the top of an entry point.
int rtn_env_int(const char* name, int dflt);
void rtn_env_sync();
int rtn_conv_halo8_try(rtn_handle_t h, const rtn_conv_desc_t* d, int grid_limit, bool stagger, bool forced, int mi_force, float* ws,
                       long long ws_cap, size_t* query, int ksplit_force, const rtn_conv_fp8_t* q8 = nullptr);
struct rtn_wgrad_frag_t { int ncb, C, Ktot, wpt, co_tile; };   // slabs in the accumulator-fragment order of rtn_wgrad_win.hip (ncb > 0): waves per tile in the slab (8 / 4), filters per tile (128 / 64)
int rtn_wgrad_finish(rtn_handle_t h, float* dW, const float* slab, int S, long long NK, float* db, const float* bslab, int N, int db_n,
                     int bS = 0 /* parts of bslab when not S */, const rtn_wgrad_frag_t* frag = nullptr);
// rtn_wgrad_win.hip: all nine taps of a 128-filter x 64-channel block per workgroup over a sliding window of the input
size_t rtn_wgrad_win_workspace_bytes(const rtn_conv_desc_t* d);
int rtn_wgrad_win_try(rtn_handle_t h, const rtn_conv_desc_t* d, float* dW, float* db, int db_n, void* workspace, size_t workspace_bytes);
int rtn_conv_ksplit_finish(rtn_handle_t h, const float* slab, int S, long long M, int N, int ld, const float* bias, int relu, void* out,
                           int out_ld);

// rtn_conv_gemm8.hip: the same schedule as a plain GEMM for the 1x1 layers with N % 256 == 0 and a bias / ReLU epilogue, one or
// two (K-concatenated) sources, stride 1 or 2.
int rtn_conv_gemm8_try(rtn_handle_t h, const rtn_conv_desc_t* d, const rtn_conv_src2_t* s2, int grid_limit, bool stagger, bool forced,
                       int mi_force, unsigned* sync, void* ws, long long ws_cap, size_t* query, int sk_mode);
int rtn_conv_halon_try(rtn_handle_t h, const rtn_conv_desc_t* d, int grid_limit, bool forced);

static inline int rtn_dtype_size(int dt) { return dt == RTN_F32 ? 4 : (dt == RTN_FP8 ? 1 : 2); }

// bf16 helpers on raw bits (device + host)
__host__ __device__ static inline float rtn_bf16_to_f32(unsigned short b) {
    unsigned int u = ((unsigned int)b) << 16;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}
