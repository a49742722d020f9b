// rtn_internal.h — shared by the librtn.so translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "rtn.h"

struct rtn_ctx {
    int device;
    hipStream_t stream;
    void* zero_page;        // 256 B of zeros on the device: source for out-of-image taps
    // f32 slabs for split-K partial sums: library-owned scratch, 64 MiB PER STREAM that uses split-K (launches on different
    // streams may overlap in time, so each stream gets its own slabs; slot 0 is allocated at create and belongs to the first
    // stream that needs one, further slots are allocated on first use)
    static constexpr int kScratchSlots = 8;
    struct { hipStream_t stream; float* ptr; bool used; } scratch[kScratchSlots];
    size_t splitk_bytes;
    int num_cus;
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

// Scratch slabs of the handle's current stream, or nullptr (then the caller runs without split-K).
inline float* rtn_splitk_scratch(rtn_ctx* h) {
    for (int i = 0; i < rtn_ctx::kScratchSlots; ++i)
        if (h->scratch[i].used && h->scratch[i].stream == h->stream) return h->scratch[i].ptr;
    for (int i = 0; i < rtn_ctx::kScratchSlots; ++i) {
        if (h->scratch[i].used) continue;
        if (!h->scratch[i].ptr && hipMalloc((void**)&h->scratch[i].ptr, h->splitk_bytes) != hipSuccess) {
            h->scratch[i].ptr = nullptr;
            (void)hipGetLastError();
            return nullptr;
        }
        h->scratch[i].used = true;
        h->scratch[i].stream = h->stream;
        return h->scratch[i].ptr;
    }
    return nullptr;
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
