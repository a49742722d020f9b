// Micro-benchmark behind DESIGN.md §3.1a ("what bounds generation 4"): how fast can ONE CU stage L2-resident data into LDS,
//   (a) by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), and
//   (b) through registers (buffer_load_dwordx4 -> VGPR -> ds_write_b128),
// with 1, 2, 4 or 8 waves of the workgroup issuing?  One workgroup per CU, every workgroup re-reads its own 64 KiB window (L2 hits
// after the first sweep).  Prints bytes per clock and CU (s_memtime clocks of wave 0) and GB/s per CU from the event time.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ldsdma_rate.hip -o tools/micro/ldsdma_rate && tools/micro/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

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
                 : "=&s"(keep) : "v"(voff), "s"(la), "s"(srd) : "memory");
}

constexpr int WINDOW = 64 * 1024;      // LDS bytes per workgroup
constexpr int SRCW = 512 * 1024;       // source bytes a workgroup sweeps (beyond the vector L1: served by the XCD's L2)
constexpr int ITERS = 64;              // sweeps of the window

// share: 0 = every workgroup sweeps its OWN window; 1 = ALL workgroups sweep the same window in the same piece order (what the weight
// tiles of the convolution kernels do: 256 CUs request the same lines at the same time); 2 = the same window, but workgroup b starts
// its sweep at piece (b * 7) mod pieces (the simultaneous requests of different CUs go to different lines / L2 channels)
//        3 = three pieces of four from the shared window (L2 hits), one from the workgroup's own (memory side)
template <int MODE, int share>         // MODE 0 = LDS-DMA, 1 = registers
__global__ __launch_bounds__(512) void stage_kernel(const char* src, unsigned long long* clocks, int nissue, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const char* win = src + ((share && share != 3) ? 0 : (size_t)blockIdx.x * SRCW);
    const i32x4 srd_sh = make_srd(src, SRCW);           // share == 3: three pieces of four come from the window EVERY workgroup reads (L2 hits), one from its own (memory side)
    const unsigned rot = share == 2 ? (blockIdx.x * 37u) % (SRCW / 1024u) : 0u;
    const i32x4 srd = make_srd(win, SRCW);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, SRCW, 0x00020000);
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (wave == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (wave < nissue) {
        const int per = WINDOW / 1024 / nissue;          // 1-KiB pieces per issuing wave and sweep
#pragma unroll 1
        for (int it = 0; it < ITERS; ++it) {
            if (MODE == 0) {
#pragma unroll 8
                for (int k = 0; k < per; ++k) {
                    const unsigned piece = (unsigned)(wave * per + k);
                    const unsigned spiece = (piece + (unsigned)(it % (SRCW / WINDOW)) * (WINDOW / 1024u) + rot) % (SRCW / 1024u);
                    if (share == 3 && (k & 3) != 3) dma16(srd_sh, spiece * 1024u + (unsigned)lane * 16u, lds_base + piece * 1024u);
                    else dma16(srd, spiece * 1024u + (unsigned)lane * 16u, lds_base + piece * 1024u);
                }
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else {
#pragma unroll 1
                for (int k0 = 0; k0 < per; k0 += 8) {
                    u32x4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((((unsigned)(wave * per + k0 + k) + (unsigned)(it % (SRCW / WINDOW)) * (WINDOW / 1024u)) % (SRCW / 1024u)) * 1024u + (unsigned)lane * 16u), 0, 0);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        *reinterpret_cast<u32x4*>(lds + (wave * per + k0 + k) * 1024 + lane * 16) = v[k];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (wave == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
        if (lane == 0) clocks[blockIdx.x] = t1 - t0;
    }
    if (sink && t == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(lds + 128);
}

int main() {
    int dev = 0;
    hipSetDevice(dev);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    char* src;
    unsigned long long* clk;
    float* sink;
    hipMalloc(&src, (size_t)cus * SRCW);
    hipMemset(src, 1, (size_t)cus * SRCW);
    hipMalloc(&clk, cus * sizeof(unsigned long long));
    hipMalloc(&sink, cus * sizeof(float));
    hipFuncSetAttribute((const void*)stage_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WINDOW);
    hipFuncSetAttribute((const void*)stage_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WINDOW);
    hipFuncSetAttribute((const void*)stage_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, WINDOW);
    hipFuncSetAttribute((const void*)stage_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, WINDOW);
    hipFuncSetAttribute((const void*)stage_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, WINDOW);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs, %d KiB source window per workgroup staged through %d KiB of LDS, %d sweeps of 64 KiB\n", cus, SRCW / 1024, WINDOW / 1024, ITERS);
    for (int share = 0; share < 4; ++share)
    for (int mode = 0; mode < (share ? 1 : 2); ++mode)
        for (int nissue : {1, 2, 4, 8}) {
            float best_ms = 1e9f;
            std::vector<unsigned long long> h(cus);
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 1) hipLaunchKernelGGL((stage_kernel<1, 0>), dim3(cus), dim3(512), WINDOW, 0, src, clk, nissue, sink);
                else if (share == 0) hipLaunchKernelGGL((stage_kernel<0, 0>), dim3(cus), dim3(512), WINDOW, 0, src, clk, nissue, sink);
                else if (share == 1) hipLaunchKernelGGL((stage_kernel<0, 1>), dim3(cus), dim3(512), WINDOW, 0, src, clk, nissue, sink);
                else if (share == 2) hipLaunchKernelGGL((stage_kernel<0, 2>), dim3(cus), dim3(512), WINDOW, 0, src, clk, nissue, sink);
                else                 hipLaunchKernelGGL((stage_kernel<0, 3>), dim3(cus), dim3(512), WINDOW, 0, src, clk, nissue, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best_ms) { best_ms = ms; hipMemcpy(h.data(), clk, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost); }
            }
            std::sort(h.begin(), h.end());
            const double bytes = (double)WINDOW * ITERS;
            printf("%s %-10s %d issuing waves: %6.1f B/clk/CU (median CU, s_memtime), %6.1f GB/s per CU, %5.2f TB/s chip (event time %.3f ms)\n",
                   share == 0 ? "own window   " : share == 1 ? "SHARED window" : share == 2 ? "shared+rotate" : "3 shared:1 own", mode == 0 ? "LDS-DMA" : "registers", nissue, bytes / (double)h[cus / 2], bytes / (best_ms * 1e-3) / 1e9,
                   bytes * cus / (best_ms * 1e-3) / 1e12, best_ms);
        }
    return 0;
}
