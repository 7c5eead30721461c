// finalize_seam_micro.hip -- what does folding the BatchNorm "finalize" launch into its PRODUCER buy?  
//
// The step has 245 finalize launches of 6-8 us (profiles/r04_bench_kernel_stats.md): producer (convolution epilogue / statistics
// pass) writes per-workgroup partial rows [R][2][C], a tiny kernel reduces them to per-channel coefficients, the consumer (apply
// pass) reads the coefficients.  Round 4 priced a last-arriver reduction behind an agent-scope RELEASE fence (buffer_wbl2 writes
// back the whole L2 of the XCD -- the producer has just dirtied it with its output) and rejected it unmeasured.  MI355X_MICROARCH.md
// (cost cell splitk-seam) prices the cheaper publish form: the partial rows are stored WRITE-THROUGH (sc1), every storing wave
// drains its stores (s_waitcnt vmcnt(0)), one lane takes an arrival ticket (relaxed agent-scope atomic add), and the workgroup whose
// ticket is last reduces the rows in FIXED index order (bitwise repeatable: no float atomics) behind one agent-scope ACQUIRE
// (buffer_inv sc1: its own L1 only).  This micro measures exactly that seam, stand-alone, on the shapes of the bench step:
//
//   chain A (today):    producer -> finalize kernel -> consumer
//   chain B (variant 1): producer + last arriver per 128-column tile (one ticket level: R rows reduced by ONE workgroup) -> consumer
//   chain C (variant 2): producer + two ticket levels (groups of G row blocks -> one compact row each; the last group leader
//                        reduces the R/G compact rows) -> consumer
//
// producer = a streaming pass over x [M][C] bf16 that writes y = x and per-(row block, column tile) sums / sums of squares (the
// byte pattern of a pointwise convolution's epilogue: M = 36 864 rows, C = 256 / 1024), consumer = y -> z = y * A + S.
// Every chain is checked against chain A (coefficients bit-identical between runs; B / C equal A up to the fp64 reduction order).
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/finalize_seam_micro.hip -o gpurun_out/finalize_seam_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned short bf16raw;
__device__ __forceinline__ float bf2f(bf16raw v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16raw f2bf(float f) { unsigned u = __float_as_uint(f); return (bf16raw)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }

struct Seam {
    float* ws;            // [R][2][C] partial rows
    float* comp;          // [R/G][2][C] compact rows (variant 2)
    unsigned* tick1;      // [C/128] (variant 1) or [R/G][C/128] (variant 2, level 1)
    unsigned* tick2;      // [C/128] (variant 2, level 2)
    float* coef;          // [2][C]: A, S
    int R, G, C;
    double count;
};

// MODE 0: rows only.  1: one ticket level.  2: two levels.
template <int MODE>
__global__ __launch_bounds__(256) void producer(const bf16raw* __restrict__ x, bf16raw* __restrict__ y, int M, int rows_per_block, Seam s) {
    __shared__ float red[2][4][128];
    __shared__ unsigned last_flag;
    const int ct = blockIdx.x % (s.C / 128), rb = blockIdx.x / (s.C / 128);
    const int t = threadIdx.x, col8 = (t & 15) * 8, rsub = t >> 4;        // 16 threads x 8 columns = 128 columns, 16 rows per trip
    const int r0 = rb * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float sm[8], sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[i] = 0.f; sq[i] = 0.f; }
    for (int r = r0 + rsub; r < r1; r += 16) {
        const size_t off = (size_t)r * s.C + ct * 128 + col8;
        const uint4 v = *reinterpret_cast<const uint4*>(x + off);
        *reinterpret_cast<uint4*>(y + off) = v;
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = __uint_as_float(w[i] << 16), b = __uint_as_float(w[i] & 0xffff0000u);
            sm[2 * i] += a; sq[2 * i] += a * a; sm[2 * i + 1] += b; sq[2 * i + 1] += b * b;
        }
    }
    // fold the 16 row-threads of a column (fixed order) -> one partial row of this workgroup
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = sm[i], b = sq[i];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if ((t & 63) < 16) { red[0][t >> 6][col8 + i] = a; red[1][t >> 6][col8 + i] = b; }
    }
    __syncthreads();
    float* row = s.ws + (size_t)rb * 2 * s.C + ct * 128;
    if (t < 256) {
        const int st = t >> 7, c = t & 127;
        const float v = (red[st][0][c] + red[st][1][c]) + (red[st][2][c] + red[st][3][c]);
        if (MODE == 0) row[(size_t)st * s.C + c] = v;
        else __hip_atomic_store(row + (size_t)st * s.C + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: write-through
    }
    if (MODE == 0) return;
    // ---- publish: every storing wave drains, barrier, ONE lane takes the ticket ------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nct = s.C / 128;
    if (t == 0) {
        unsigned* tk = MODE == 1 ? s.tick1 + ct : s.tick1 + (size_t)(rb / s.G) * nct + ct;
        const unsigned members = MODE == 1 ? (unsigned)s.R : (unsigned)min(s.G, s.R - (rb / s.G) * s.G);
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (old == members - 1) ? 1u : 0u;
        if (old == members - 1) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
        if (old == members - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!last_flag) return;
    const int st = t >> 7, c = t & 127;
    if (MODE == 1) {
        // the last arriver of this column tile: R rows in index order, fp64
        double acc = 0.0;
        const float* src = s.ws + (size_t)st * s.C + ct * 128 + c;
        for (int r = 0; r < s.R; ++r) acc += (double)src[(size_t)r * 2 * s.C];
        __shared__ double tot[2][128];
        tot[st][c] = acc;
        __syncthreads();
        if (t < 128) {
            const double m = tot[0][t] / s.count;
            double var = tot[1][t] / s.count - m * m;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + 1e-5);
            s.coef[ct * 128 + t] = (float)is;
            s.coef[s.C + ct * 128 + t] = (float)(-m * is);
        }
        return;
    }
    // MODE 2, level 1: this group's rows -> one compact row (sc1), then the second ticket
    {
        const int g = rb / s.G, ra = g * s.G, rz = min(s.R, ra + s.G);
        double acc = 0.0;
        const float* src = s.ws + (size_t)st * s.C + ct * 128 + c;
        for (int r = ra; r < rz; ++r) acc += (double)src[(size_t)r * 2 * s.C];
        // (the compact row keeps the fp64 sum as two floats: hi + lo)
        const float hi = (float)acc, lo = (float)(acc - (double)hi);
        float* cr = s.comp + ((size_t)g * 2 + st) * 2 * s.C + ct * 128 + c;
        __hip_atomic_store(cr, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cr + s.C, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int ngrp = (s.R + s.G - 1) / s.G;
        if (t == 0) {
            const unsigned old = __hip_atomic_fetch_add(s.tick2 + ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = (old == (unsigned)ngrp - 1) ? 1u : 0u;
            if (old == (unsigned)ngrp - 1) __hip_atomic_store(s.tick2 + ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)ngrp - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!last_flag) return;
        double a2 = 0.0;
        for (int q = 0; q < ngrp; ++q) {
            const float* p = s.comp + ((size_t)q * 2 + st) * 2 * s.C + ct * 128 + c;
            a2 += (double)p[0] + (double)p[s.C];
        }
        __shared__ double tot2[2][128];
        tot2[st][c] = a2;
        __syncthreads();
        if (t < 128) {
            const double m = tot2[0][t] / s.count;
            double var = tot2[1][t] / s.count - m * m;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + 1e-5);
            s.coef[ct * 128 + t] = (float)is;
            s.coef[s.C + ct * 128 + t] = (float)(-m * is);
        }
    }
}

// the separate finalize launch of chain A: 8 channels x 128 partial lanes per workgroup (the shape of stats.hip's bn_finalize_kernel)
__global__ __launch_bounds__(1024) void finalize(Seam s) {
    __shared__ double sm[128][2][8];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + tx;
    double a = 0.0, b = 0.0;
    for (int r = ty; r < s.R; r += 128) {
        a += (double)s.ws[((size_t)r * 2 + 0) * s.C + c];
        b += (double)s.ws[((size_t)r * 2 + 1) * s.C + c];
    }
    sm[ty][0][tx] = a; sm[ty][1][tx] = b;
    __syncthreads();
    if (ty == 0) {
        double sa = 0.0, sb = 0.0;
        for (int k = 0; k < 128; ++k) { sa += sm[k][0][tx]; sb += sm[k][1][tx]; }
        const double m = sa / s.count;
        double var = sb / s.count - m * m;
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + 1e-5);
        s.coef[c] = (float)is;
        s.coef[s.C + c] = (float)(-m * is);
    }
}

__global__ __launch_bounds__(256) void consumer(const bf16raw* __restrict__ y, bf16raw* __restrict__ z, size_t n8, int C, const float* __restrict__ coef) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(y + i * 8);
        const int c0 = (int)((i * 8) % C);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = __uint_as_float(w[k] << 16) * coef[c0 + 2 * k] + coef[C + c0 + 2 * k];
            const float b = __uint_as_float(w[k] & 0xffff0000u) * coef[c0 + 2 * k + 1] + coef[C + c0 + 2 * k + 1];
            o[k] = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
        }
        *reinterpret_cast<uint4*>(z + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 36864, C = argc > 2 ? atoi(argv[2]) : 1024, R = argc > 3 ? atoi(argv[3]) : 64;
    const int G = argc > 4 ? atoi(argv[4]) : 16, reps = argc > 5 ? atoi(argv[5]) : 200;
    const int rpb = (M + R - 1) / R;
    const size_t n = (size_t)M * C;
    bf16raw *x, *y, *z;
    CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&z, n * 2));
    std::vector<bf16raw> hx(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) { float f = (float)(rand() % 2001 - 1000) / 500.f + (float)((i % C) % 7) * 0.1f; unsigned u; memcpy(&u, &f, 4); hx[i] = (bf16raw)(u >> 16); }
    CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice));
    Seam s;
    s.R = R; s.G = G; s.C = C; s.count = (double)M;
    const int ngrp = (R + G - 1) / G;
    CK(hipMalloc(&s.ws, (size_t)R * 2 * C * 4)); CK(hipMalloc(&s.comp, (size_t)ngrp * 4 * C * 4));
    CK(hipMalloc(&s.tick1, (size_t)(ngrp > 1 ? ngrp : 1) * (C / 128) * 4 + 64)); CK(hipMalloc(&s.tick2, (size_t)(C / 128) * 4 + 64));
    CK(hipMemset(s.tick1, 0, (size_t)(ngrp > 1 ? ngrp : 1) * (C / 128) * 4 + 64)); CK(hipMemset(s.tick2, 0, (size_t)(C / 128) * 4 + 64));
    float* coefs[3];
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&coefs[i], 2 * C * 4)); CK(hipMemset(coefs[i], 0, 2 * C * 4)); }
    const dim3 pg((unsigned)(R * (C / 128)));
    const size_t n8 = n / 8;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain = [&](int mode) {
        s.coef = coefs[mode];
        if (mode == 0) {
            hipLaunchKernelGGL(producer<0>, pg, dim3(256), 0, st, x, y, M, rpb, s);
            hipLaunchKernelGGL(finalize, dim3(C / 8), dim3(1024), 0, st, s);
        } else if (mode == 1) {
            hipLaunchKernelGGL(producer<1>, pg, dim3(256), 0, st, x, y, M, rpb, s);
        } else {
            hipLaunchKernelGGL(producer<2>, pg, dim3(256), 0, st, x, y, M, rpb, s);
        }
        hipLaunchKernelGGL(consumer, dim3(2048), dim3(256), 0, st, y, z, n8, C, (const float*)s.coef);
    };
    float ms[3] = {0, 0, 0};
    // interleaved rounds in one process (cdna_hip_programming.md rule 24)
    for (int round = 0; round < 5; ++round)
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 5; ++w) chain(mode);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) chain(mode);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (round == 0 || t / reps < ms[mode]) ms[mode] = t / reps;
        }
    // correctness: B / C against A, and repeatability
    std::vector<float> h[3];
    for (int i = 0; i < 3; ++i) { h[i].resize(2 * C); CK(hipMemcpy(h[i].data(), coefs[i], 2 * C * 4, hipMemcpyDeviceToHost)); }
    double d1 = 0, d2 = 0, ref = 0;
    for (int i = 0; i < 2 * C; ++i) { d1 = fmax(d1, fabs((double)h[1][i] - h[0][i])); d2 = fmax(d2, fabs((double)h[2][i] - h[0][i])); ref = fmax(ref, fabs((double)h[0][i])); }
    int stale = 0;
    for (int rep = 0; rep < 50; ++rep)
        for (int mode = 1; mode < 3; ++mode) {
            CK(hipMemsetAsync(coefs[mode], 0, 2 * C * 4, st));
            chain(mode);
            CK(hipStreamSynchronize(st));
            std::vector<float> g(2 * C);
            CK(hipMemcpy(g.data(), coefs[mode], 2 * C * 4, hipMemcpyDeviceToHost));
            if (memcmp(g.data(), h[mode].data(), 2 * C * 4) != 0) ++stale;
        }
    printf("M %d C %d R %d G %d (producer grid %u): chain us  A(launch) %.2f  B(one ticket) %.2f  C(two levels) %.2f  | B-A %+.2f  C-A %+.2f  | "
           "max|dcoef| B %.2e C %.2e (max |coef| %.2f)  non-repeatable runs %d / 100\n",
           M, C, R, G, pg.x, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, (ms[1] - ms[0]) * 1e3, (ms[2] - ms[0]) * 1e3, d1, d2, ref, stale);
    return 0;
}
