// Store-pattern micro-benchmark: how fast does the chip drain 16-byte-per-lane stores of a [M][N] bf16 matrix (N = 1024, 2 KB rows)
// when a wave-instruction covers   (a) 16 rows x 64 B  (the pointwise kernel's epilogue: 4 lanes per row)
//                                   (b)  4 rows x 256 B (the generic kernel's wide epilogue)
//                                   (c)  1 row  x 1 KB  (a streaming kernel)
// and the four waves of a workgroup take adjacent column groups (a: 4 x 64 B = one 256-byte run per row per workgroup).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/store_micro.hip -o /tmp/store_micro && /tmp/store_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* y, int M, int rowchunks /* 16-byte chunks per row */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 1u);
    // a workgroup owns 64 rows x 128 columns (256 B) like one pointwise tile x panel; grid = (M/64) * (N/128)
    const int panels = rowchunks / 16;
    const int tile = blockIdx.x / panels, panel = blockIdx.x % panels;
    for (int rep = 0; rep < 1; ++rep) {
        if (MODE == 0) {          // wave w: columns 32w..32w+31 (64 B = 4 chunks), 16 rows per instruction, 4 instructions
            for (int i = 0; i < 4; ++i) {
                const int row = tile * 64 + i * 16 + (lane & 15), ch = panel * 16 + wave * 4 + (lane >> 4);
                if (row < M) y[(size_t)row * rowchunks + ch] = v;
            }
        } else if (MODE == 1) {   // wave w: rows 16w..16w+15, all 16 chunks of the panel: 4 rows x 256 B per instruction
            for (int i = 0; i < 4; ++i) {
                const int row = tile * 64 + wave * 16 + i * 4 + (lane >> 4), ch = panel * 16 + (lane & 15);
                if (row < M) y[(size_t)row * rowchunks + ch] = v;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_stream(uint4* y, size_t n) {
    const uint4 v = make_uint4(threadIdx.x, 2u, blockIdx.x, 1u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = v;
}
int main() {
    const int M = 36864, N = 1024, rowchunks = N * 2 / 16;
    uint4* y;
    hipMalloc(&y, (size_t)M * rowchunks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (M / 64) * (rowchunks / 16);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 20; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, y, M, rowchunks);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, y, M, rowchunks);
            else hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, y, (size_t)M * rowchunks);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it > 2 && ms < best) best = ms;
        }
        printf("mode %d: %.1f us  %.2f TB/s  (75.5 MB written)\n", mode, best * 1e3, (double)M * N * 2 / (best * 1e-3) / 1e12);
    }
    return 0;
}
