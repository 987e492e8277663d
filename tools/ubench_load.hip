// Load-path microbenchmark (gfx950): how many bytes per clock per CU can a workgroup pull from an L2/MALL-resident
// tensor into LDS, by LDS-DMA vs. through registers?  Access pattern = an implicit-GEMM A tile: each wave-instruction
// fetches 16 B per lane, CPR lanes per row, rows `pitch` bytes apart.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_load.hip -o gpurun_out/ubench_load && gpurun_out/ubench_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 bload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
#else
    return f4{0, 0, 0, 0};
#endif
}

// MODE 0: LDS-DMA; 1: buffer_load -> VGPR -> ds_write_b128; 2: buffer_load -> VGPR only (consumed by a cheap xor)
// Each wave owns a window of `win` bytes inside its workgroup's region and sweeps it `iters` times, DEPTH loads in flight.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k_load(const float* x, unsigned bytes, unsigned wg_region, unsigned pitch, int cpr, int iters,
                                              float* sink) {
    __shared__ __attribute__((aligned(16))) float smem[4 * DEPTH * 256];     // 1 KiB per piece per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, bytes, 0x00020000);
    const unsigned base = pitch & 1 ? 0 : (blockIdx.x * wg_region) % (bytes - wg_region + 1);   // odd pitch flag = every WG reads the SAME region
    pitch &= ~1u;
    const int rpi = 64 / cpr;                       // rows per piece
    const unsigned lane_off = (lane / cpr) * pitch + (lane % cpr) * 16;
    const unsigned piece_stride = rpi * pitch;      // next piece = next rpi rows
    const unsigned pieces = wg_region / piece_stride / 4;   // per wave
    float* my = smem + wave * DEPTH * 256;
    f4 acc = {0, 0, 0, 0};
    unsigned p = 0;
    for (int it = 0; it < iters; ++it) {
        f4 v[DEPTH];
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) {
            const unsigned off = base + (wave * pieces + p) * piece_stride;
            if (MODE == 0) dma16(rs, my + dd * 256, lane_off + off, 0);
            else v[dd] = bload16(rs, lane_off + off, 0);
            p = p + 1 == pieces ? 0 : p + 1;
        }
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) *reinterpret_cast<f4*>(my + dd * 256 + lane * 4) = v[dd];
        } else {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) acc += v[dd];
        }
    }
    __syncthreads();
    if (MODE != 2) acc = *reinterpret_cast<f4*>(my + lane * 4);
    if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1];
}

template <int MODE, int DEPTH>
static void run(const char* name, const float* x, unsigned bytes, unsigned region, unsigned pitch, int cpr, int wgs, float* sink) {
    const int iters = 400;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_load<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, x, bytes, region, pitch, cpr, 20, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_load<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, x, bytes, region, pitch, cpr, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double tot = (double)wgs * 4 * iters * DEPTH * 1024.0;
    printf("%-28s depth %2d wgs %5d region %7u pitch %5u cpr %d: %8.1f GB/s  (%.1f B/clk/CU @2.4GHz)  %.0f ns/iter\n", name, DEPTH, wgs, region, pitch,
           cpr, tot / ms * 1e-6, tot / (ms * 1e-3) / 2.4e9 / 256, ms * 1e6 / iters);
}

int main() {
    const unsigned bytes = 1u << 30;
    float *x, *sink;
    CK(hipMalloc(&x, bytes)); CK(hipMemset(x, 0, bytes)); CK(hipMalloc(&sink, 4096));
    // region per WG: 64 KiB (L2-resident once warmed: 1024 WGs x 64 KiB = 64 MiB > 32 MiB L2, MALL-resident), and 16 KiB (L2)
    for (int wgs : {256, 512, 1024}) {
        printf("--- latency probes (depth 1-2), wgs %d: time per iteration = issue->landed latency\n", wgs);
        run<0, 1>("dma distinct 64K", x, bytes, 65536, 256, 4, wgs, sink);
        run<0, 2>("dma distinct 64K", x, bytes, 65536, 256, 4, wgs, sink);
        run<0, 1>("dma SAME 64K region", x, bytes, 65536, 257, 4, wgs, sink);
        run<0, 2>("dma SAME 64K region", x, bytes, 65536, 257, 4, wgs, sink);
        run<0, 1>("dma SAME 144K pitch2304", x, bytes, 147456, 2305, 4, wgs, sink);
        run<0, 2>("dma SAME 144K pitch2304", x, bytes, 147456, 2305, 4, wgs, sink);
        run<2, 1>("vgpr SAME 144K pitch2304", x, bytes, 147456, 2305, 4, wgs, sink);
    }
    for (unsigned region : {65536u}) {
        for (int wgs : {256 * 4}) {
            printf("--- region %u wgs %d\n", region, wgs);
            run<0, 4>("dma   rows64B", x, bytes, region, 256, 4, wgs, sink);
            run<0, 8>("dma   rows64B", x, bytes, region, 256, 4, wgs, sink);
            run<0, 8>("dma   contiguous", x, bytes, region, 64, 4, wgs, sink);
            run<0, 8>("dma   rows128B", x, bytes, region, 512, 8, wgs, sink);
            run<1, 4>("vgpr+dswrite rows64B", x, bytes, region, 256, 4, wgs, sink);
            run<1, 8>("vgpr+dswrite rows64B", x, bytes, region, 256, 4, wgs, sink);
            run<1, 8>("vgpr+dswrite contiguous", x, bytes, region, 64, 4, wgs, sink);
            run<2, 8>("vgpr only rows64B", x, bytes, region, 256, 4, wgs, sink);
            run<2, 8>("vgpr only contiguous", x, bytes, region, 64, 4, wgs, sink);
        }
    }
    return 0;
}
