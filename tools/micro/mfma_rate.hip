// Micro-benchmark behind DESIGN.md §3.1a ("the ceiling the tower kernel is measured against"): what does a BARE bf16 MFMA loop reach on
// this part on random operands, and at which clock?  No memory traffic in the loop: every wave multiplies register-resident random
// fragments into 16 (or 32) independent 16x16 accumulators, one or two waves per SIMD, every CU busy, for several milliseconds.
// Prints TFLOP/s from the event time and the in-kernel shader clock (delta s_memtime / delta s_memrealtime x 100 MHz,
// MI355X_MICROARCH.md "DVFS give-back" item 6), for random and for all-zero operands, with and without the fragment reads of the
// tower kernel's phase (6 conflict-free ds_read_b128 per 16 MFMAs from a random LDS image).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate && tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int OUTER = 4000;           // outer iterations of 64 MFMAs per wave

// READS: 0 = operands stay in registers; 1 = every group of 16 MFMAs re-reads its 8 fragments from LDS (ds_read_b128, swizzled rows)
template <int READS>
__global__ __launch_bounds__(512, 2) void mfma_kernel(const uint4* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // a 64 KiB random LDS image (512 rows x 128 B)
    for (int i = t; i < 4096; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = src[(blockIdx.x * 64 + i) & 65535];
    __syncthreads();
    const int lrow = lane & 15, kq = lane >> 4;
    uint4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = src[(blockIdx.x * 512 + t + 1024 * i) & 65535];
        fb[i] = src[(blockIdx.x * 512 + t + 1024 * i + 4096) & 65535];
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned arow = (unsigned)(((wave & 3) * 64 + lrow) * 128 + ((kq ^ (lrow & 7)) << 4));
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if (wave == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
    }
#pragma unroll 1
    for (int it = 0; it < OUTER; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (READS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(lds + ((arow + (unsigned)(i * 2048 + g * 8192)) ^ (unsigned)((it & 1) * 64)));
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[(g + j) & 3] = *reinterpret_cast<const uint4*>(lds + (((arow & 0x1fffu) + (unsigned)(32768 + j * 2048 + g * 4096)) ^ (unsigned)((it & 1) * 64)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
        }
    }
    if (wave == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
        if (lane == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[t] = s;
}

template <int READS>
static void run(const char* label, const uint4* src, float* sink, unsigned long long* clk, int threads, int cus) {
    hipFuncSetAttribute((const void*)mfma_kernel<READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_kernel<READS>, dim3(cus), dim3(threads), 65536, 0, src, sink, clk);   // warm, and long enough for the clock to settle
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(mfma_kernel<READS>, dim3(cus), dim3(threads), 65536, 0, src, sink, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(cus * 2);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < cus; ++b) ghz.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flop = (double)cus * (threads / 64) * OUTER * 64.0 * 16 * 16 * 32 * 2 * reps;
    printf("%-52s %7.3f ms/launch  %7.1f TFLOP/s  in-kernel clock %.2f GHz (median of %d workgroups)\n", label, ms / reps, flop / (ms * 1e-3) / 1e12,
           ghz[cus / 2], cus);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<unsigned> h(65536 * 4);
    srand(7);
    for (auto& v : h) {                                      // random bf16 pairs in [-2, 2): sign, exponent 125..128, random mantissa
        unsigned w = 0;
        for (int k = 0; k < 2; ++k) w |= ((unsigned)((rand() & 1) << 15 | (125 + (rand() & 3)) << 7 | (rand() & 127))) << (16 * k);
        v = w;
    }
    uint4 *src, *zero; float* sink; unsigned long long* clk;
    hipMalloc(&src, 65536 * 16); hipMalloc(&zero, 65536 * 16); hipMalloc(&sink, 4096); hipMalloc(&clk, cus * 16);
    hipMemcpy(src, h.data(), 65536 * 16, hipMemcpyHostToDevice);
    hipMemset(zero, 0, 65536 * 16);
    printf("%d CUs; bf16 16x16x32 MFMA, 16 independent accumulators per wave, dense peak 2.5 PFLOP/s at 2.4 GHz\n", cus);
    run<0>("random operands, registers only, 2 waves/SIMD", src, sink, clk, 512, cus);
    run<0>("random operands, registers only, 1 wave/SIMD", src, sink, clk, 256, cus);
    run<0>("ZERO operands, registers only, 2 waves/SIMD", zero, sink, clk, 512, cus);
    run<1>("random operands, 6 ds_read_b128 per 16 MFMAs, 2 waves/SIMD", src, sink, clk, 512, cus);
    run<1>("random operands, 6 ds_read_b128 per 16 MFMAs, 1 wave/SIMD", src, sink, clk, 256, cus);
    run<1>("ZERO operands, 6 ds_read_b128 per 16 MFMAs, 2 waves/SIMD", zero, sink, clk, 512, cus);
    return 0;
}
