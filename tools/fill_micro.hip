// fill_micro.hip -- where is the ceiling of the global -> LDS fill path?  (profiles/r02_experiments.md section 5)
// Every workgroup (256 threads, WPC per CU) streams `iters` 16-KiB tiles into LDS by LDS-DMA (16 B per lane), or into
// registers, from a region of `span` bytes per workgroup group:
//   span =   8 KiB  -> every re-read hits the CU's vector L1 (if LDS-DMA allocates there)
//   span =   2 MiB  -> per-XCD L2 hits
//   span = 512 MiB  -> HBM stream
// Reports bytes per ns and per CU.  build: hipcc --offload-arch=gfx950 -O3 tools/fill_micro.hip -o gpurun_out/fill_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int __attribute__((ext_vector_type(4))) i32x4;

__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc));
}

template <int MODE>   // 0: LDS-DMA, 1: register loads (16 B per lane), 2: LDS-DMA + an MFMA stream beside it, 3 / 4: LDS-DMA + 8 / 16
                      // ds_read_b128 per wave and tile (2 / 4 bytes of fragment reads per byte filled), 5: 3 + the MFMA stream
__global__ __launch_bounds__(256) void fill_kernel(const char* src, unsigned long long span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // each workgroup walks its own window of the region (windows of different workgroups overlap when span is small)
    const unsigned long long base = ((unsigned long long)blockIdx.x * 16384ull) % span;
    const char* p = src + base;
    i32x4 r;
    r.x = (int)(unsigned)(unsigned long long)p;
    r.y = (int)(((unsigned long long)p >> 32) & 0xffffu);
    r.z = (int)0x7fffffff;
    r.w = 0x00020000;
    typedef float __attribute__((ext_vector_type(4))) f32x4;
    typedef __bf16 __attribute__((ext_vector_type(8))) bf16x8;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    uint4 regs = make_uint4(0, 0, 0, 0);
    unsigned off = 0;
    const unsigned wrap = (unsigned)(span < (1ull << 31) ? span : (1ull << 31));
    for (int it = 0; it < iters; ++it) {
        // one 16 KiB tile: 4 waves x 4 pieces of 1 KiB
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned o = (off + (unsigned)((q * 4 + wave) * 1024 + lane * 16)) % wrap;
            if (MODE == 1) {
                const uint4 v = *reinterpret_cast<const uint4*>(p + o);
                regs.x ^= v.x; regs.y ^= v.y; regs.z ^= v.z; regs.w ^= v.w;
            } else {
                dma16(r, lds0 + (unsigned)(((it & 1) * 16 + q * 4 + wave) * 1024), o);
            }
        }
        if (MODE >= 3) {
            // fragment-read traffic from the OTHER half of the LDS (no ordering against the transfers needed for a rate test)
            const char* rd = smem + 32768 + lane * 16;
#pragma unroll
            for (int m = 0; m < (MODE == 4 ? 16 : 8); ++m) {
                unsigned __attribute__((ext_vector_type(4))) v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const char*)rd), "n"(m * 1024));
                asm volatile("" :: "v"(v));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (MODE == 2 || MODE == 5) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
        }
        if (MODE != 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        off += 16384u;
        if (off + 16384u > wrap) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = (float)(regs.x ^ regs.y ^ regs.z ^ regs.w) + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    s += (float)smem[(t * 16) & 16383];
    if (s == 123.456f) sink[0] = s;
}

// the convolution kernels' A-tile pattern: one DMA instruction = 8 rows x 128 B (lane -> row lane>>3, 16-byte chunk lane&7), the rows
// `stride` bytes apart (a pixel's channel vector), a tile = 192 consecutive rows, the K loop walks `ksteps` 128-byte columns of the
// same rows and then moves to the next 192 rows; all inside a 2 MiB window per 16 workgroups (L2 hits)
__global__ __launch_bounds__(256) void gather_kernel(const char* src, unsigned stride, int ksteps, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned rows_per_window = (2u << 20) / stride;            // rows in a 2 MiB window
    const char* p = src + (unsigned long long)(blockIdx.x & 15) * (2ull << 20);
    i32x4 r;
    r.x = (int)(unsigned)(unsigned long long)p;
    r.y = (int)(((unsigned long long)p >> 32) & 0xffffu);
    r.z = (int)0x7fffffff;
    r.w = 0x00020000;
    unsigned row0 = ((unsigned)(blockIdx.x >> 4) * 192u) % rows_per_window;
    int kt = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {                                 // 192 rows = 24 pieces of 8 rows, 6 per wave
            unsigned row = row0 + (unsigned)((q * 4 + wave) * 8 + (lane >> 3));
            if (row >= rows_per_window) row -= rows_per_window;
            const unsigned o = row * stride + (unsigned)kt * 128u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
            dma16(r, lds0 + (unsigned)(((it & 1) * 24 + q * 4 + wave) * 1024), o);
        }
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if (++kt == ksteps) {
            kt = 0;
            row0 += 192u * 97u;                                       // another tile of the window
            row0 %= rows_per_window;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((float)smem[(t * 16) & 16383] == 123.456f) sink[0] = 1.f;
}

static void run_gather(const char* d, unsigned stride, int wpc, int iters, float* sink) {
    const int grid = 256 * wpc, ksteps = (int)(stride / 128u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 49152, 0, d, stride, ksteps, iters / 8, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 49152, 0, d, stride, ksteps, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * 24576.0;
    printf("gather 8 rows x 128 B, row stride %5u B, %d WG/CU: %7.2f TB/s chip  %6.1f GB/s per CU  (%.3f ms)\n", stride, wpc,
           bytes / ms * 1e-9, bytes / ms * 1e-6 / 256.0, ms);
}

template <int MODE>
static void run(const char* name, const char* d, unsigned long long span, int wpc, int iters, float* sink) {
    const int grid = 256 * wpc;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fill_kernel<MODE>, dim3(grid), dim3(256), 49152, 0, d, span, iters / 8, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fill_kernel<MODE>, dim3(grid), dim3(256), 49152, 0, d, span, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * 16384.0;
    printf("%-28s span %10llu B  %d WG/CU: %7.2f TB/s chip  %6.1f GB/s per CU  (%.3f ms)\n", name, span, wpc, bytes / ms * 1e-9,
           bytes / ms * 1e-6 / 256.0, ms);
}

int main(int argc, char** argv) {
    const unsigned long long total = 1ull << 29;
    char* d = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&d, total + (1 << 20)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(d, 1, total + (1 << 20));
    hipDeviceSynchronize();
    const unsigned long long spans[] = {8192ull, 16384ull, 1ull << 21, 1ull << 24, total};
    const int wlo = argc > 1 ? atoi(argv[1]) : 2, whi = argc > 2 ? atoi(argv[2]) : 3;      // workgroups (of 4 waves) per CU
    for (int wpc = wlo; wpc <= whi; ++wpc)
        for (unsigned long long span : spans) {
            const int iters = span >= (1ull << 29) ? 512 : 2048;
            run<0>("LDS-DMA", d, span, wpc, iters, sink);
            run<1>("register loads", d, span, wpc, iters, sink);
            run<2>("LDS-DMA + 16 MFMA per tile", d, span, wpc, iters, sink);
            if (span == (1ull << 21)) {
                run<3>("LDS-DMA + 8 ds_read_b128", d, span, wpc, iters, sink);
                run<4>("LDS-DMA + 16 ds_read_b128", d, span, wpc, iters, sink);
                run<5>("LDS-DMA + 8 reads + 16 MFMA", d, span, wpc, iters, sink);
            }
        }
    const unsigned strides[] = {128u, 256u, 512u, 640u, 1024u, 1152u, 2048u, 2176u, 4096u, 4224u};
    for (int wpc = wlo; wpc <= whi; ++wpc)
        for (unsigned st : strides) run_gather(d, st, wpc, 2048, sink);
    return 0;
}
