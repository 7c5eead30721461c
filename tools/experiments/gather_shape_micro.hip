// gather_shape_micro.hip -- does the SHAPE of an LDS-DMA piece (64 lanes x 16 B = 1 KiB) set the rate at which a [M][ROWB] matrix streams
// from HBM into LDS?  (profiles/r05_experiments.md section 11 conjectured it: the pointwise kernels gather 8 rows x 128 B at a 2 KB row
// stride and draw 2.9 - 3.5 TB/s where linear streams draw 5+.)  Every workgroup (256 threads, 2 per CU, persistent) walks 24 KiB stages
// of the matrix through a ring of NST stages with counted vmcnt waits; nothing reads the LDS.
//   shape 0: piece = 8 rows x 128 B  (stage = 48 rows x 512 B: what conv_pwk.hip fills)
//   shape 1: piece = 2 rows x 512 B  (same 48 x 512 B stage)
//   shape 2: piece = 1 row  x 1 KiB  (stage = 24 rows x 1 KiB)
//   shape 3: piece = 1 KiB of a linear stream (stage = 24 KiB contiguous: 12 whole rows)
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/gather_shape_micro.hip -o gpurun_out/gather_shape_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int __attribute__((ext_vector_type(4))) i32x4;
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int ROWB = 2048;      // bytes per matrix row (1024 bf16 channels)
constexpr int NST = 3;          // ring stages of 24 KiB
constexpr int NP = 6;           // pieces per wave and stage (24 per stage)

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void stream_kernel(const char* src, int M, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    i32x4 r;
    r.x = (int)(unsigned)(unsigned long long)src;
    r.y = (int)(((unsigned long long)src >> 32) & 0xffffu);
    r.z = (int)((unsigned)M * (unsigned)ROWB);
    r.w = 0x00020000;
    // stages in walking order: shapes 0/1: (row tile of 48, K quarter of 512 B) quarter-major inside a row tile; 2: (24 rows, half of 1 KiB); 3: linear
    const int nstage = (int)((long long)M * ROWB / (24 * 1024));
    const int g = (int)gridDim.x, w = (int)blockIdx.x;
    const int s0 = (int)((long long)w * nstage / g), s1 = (int)((long long)(w + 1) * nstage / g);
    auto issue = [&](int s, int slot) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pi = j * 4 + wave;             // piece 0..23 of the stage
            unsigned off;
            if (SHAPE == 0) {
                const int rt = s >> 2, kq = s & 3;   // 48-row tile, 512-B quarter
                const int blk = pi / 6, rp = pi - blk * 6;      // 128-B block of the quarter, 8-row group
                off = (unsigned)(rt * 48 + rp * 8 + (lane >> 3)) * ROWB + (unsigned)(kq * 512 + blk * 128 + (lane & 7) * 16);
            } else if (SHAPE == 1) {
                const int rt = s >> 2, kq = s & 3;
                off = (unsigned)(rt * 48 + pi * 2 + (lane >> 5)) * ROWB + (unsigned)(kq * 512 + (lane & 31) * 16);
            } else if (SHAPE == 2) {
                const int rt = s >> 1, kh = s & 1;
                off = (unsigned)(rt * 24 + pi) * ROWB + (unsigned)(kh * 1024 + lane * 16);
            } else {
                off = (unsigned)s * 24576u + (unsigned)(pi * 1024 + lane * 16);
            }
            dma16(r, lds0 + (unsigned)(slot * 24576 + pi * 1024), off);
        }
    };
    int issued = s0;
    for (int k = 0; k < NST - 1 && issued < s1; ++k, ++issued) issue(issued, (issued - s0) % NST);
    for (int s = s0; s < s1; ++s) {
        if (issued < s1) { issue(issued, (issued - s0) % NST); ++issued; wait_vm<(NST - 1) * NP>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }
    if (sink && t == 0 && M < 0) sink[0] = smem[0];
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 36864;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    char* src;
    const size_t bytes = (size_t)M * ROWB;
    // several distinct matrices walked round-robin, so that a launch never finds its input in the 256 MiB Infinity Cache
    const int NBUF = 6;
    hipMalloc(&src, bytes * NBUF);
    hipMemset(src, 1, bytes * NBUF);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int shape, int grid) {
        const int lds = NST * 24576;
        auto launch = [&](int i) {
            const char* p = src + (size_t)(i % NBUF) * bytes;
            if (shape == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), lds, 0, p, M, (float*)nullptr);
            if (shape == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), lds, 0, p, M, (float*)nullptr);
            if (shape == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), lds, 0, p, M, (float*)nullptr);
            if (shape == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(256), lds, 0, p, M, (float*)nullptr);
        };
        for (int i = 0; i < 3; ++i) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch(i);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("shape %d grid %4d: %7.2f us per %.1f MB = %6.0f GB/s\n", shape, grid, us, bytes / 1e6, bytes / us / 1e3);
    };
    hipFuncSetAttribute((const void*)stream_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 24576);
    hipFuncSetAttribute((const void*)stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 24576);
    hipFuncSetAttribute((const void*)stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 24576);
    hipFuncSetAttribute((const void*)stream_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 24576);
    for (int grid : {256, 512})
        for (int shape = 0; shape < 4; ++shape) run(shape, grid);
    return 0;
}
