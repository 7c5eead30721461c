#include "conv_common.hpp"

namespace mrfp {

// ---------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 master  ->  forward pack Wf[Npad][R][S][Cpad]  (T)
//                                   ->  dgrad   pack Wd[Cin][R][S][Npad] with taps flipped (T)
// (pad channels are zero).  One thread per destination element.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int N, int C,
                                   int R, int S, int Npad, int Cpad) {
    const int64_t nf = (int64_t)Npad * R * S * Cpad, nd = (int64_t)C * R * S * Npad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nf) {
            if (!wf) continue;
            const int c = (int)(i % Cpad);
            int64_t rest = i / Cpad;
            const int s = (int)(rest % S); rest /= S;
            const int r = (int)(rest % R);
            const int n = (int)(rest / R);
            const float v = (n < N && c < C) ? w[(((int64_t)n * C + c) * R + r) * S + s] : 0.f;
            wf[i] = from_f<T>(v);
        } else {
            if (!wd) continue;
            const int64_t k = i - nf;
            const int n = (int)(k % Npad);
            int64_t rest = k / Npad;
            const int s = (int)(rest % S); rest /= S;
            const int r = (int)(rest % R);
            const int c = (int)(rest / R);
            const float v = (n < N) ? w[(((int64_t)n * C + c) * R + (R - 1 - r)) * S + (S - 1 - s)] : 0.f;
            wd[k] = from_f<T>(v);
        }
    }
}

// All weight packs of a model in ONE launch (after the optimizer step): jobs[] and the exclusive prefix of their element
// counts live in device memory; every thread finds its job by binary search and packs one element as above.
struct PackJob {
    const float* w;
    void* wf;
    void* wd;
    int N, C, R, S, Npad, Cpad;
};
// One workgroup = one (job, 64 output channels n, 8 input channels c) brick: it reads the 64 runs w[n][c0..c0+7][:][:] of
// 8*R*S contiguous floats (coalesced), keeps the brick in LDS, and writes both packs from there with the channel index
// that is contiguous in the pack as the fastest thread index: wf[n][r][s][c0..c0+7] (16-byte runs) and
// wd[c][R-1-r][S-1-s][n0..n0+63] (128-byte runs).  prefix[] counts bricks (plus the pad bricks that zero the padding).
constexpr int kBrickN = 64, kBrickC = 8, kBrickRSMax = 9;      // filters up to 3x3 (18.7 KB of LDS); larger ones pack per layer
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackJob* __restrict__ jobs,
                                                                    const int64_t* __restrict__ prefix, int njobs, int64_t total) {
    __shared__ float brick[kBrickN][kBrickC * kBrickRSMax + 1];      // (>= 64 + 1 columns: the pointwise bricks fit)
    const int64_t wg = blockIdx.x;
    int lo = 0, hi = njobs;                 // largest j with prefix[j] <= wg  (uniform: every lane does the same search)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= wg) lo = mid; else hi = mid;
    }
    const PackJob jb = jobs[lo];
    const int N = jb.N, C = jb.C, R = jb.R, S = jb.S, Npad = jb.Npad, Cpad = jb.Cpad, RS = R * S;
    // input channels per brick: 8 for 3x3 filters, 64 for pointwise ones (the same 64 x 72-float brick either way; with 8
    // channels a pointwise brick was 2 KB of work behind a 7-step binary search: 240 us for the 124 packs of ResNet-101)
    const int bc = RS == 1 ? kBrickC * 8 : kBrickC;
    const int ncb = (Cpad + bc - 1) / bc;
    const int local = (int)(wg - prefix[lo]);
    const int n0 = (local / ncb) * kBrickN, c0 = (local % ncb) * bc;
    const int run = bc * RS;                // floats per n in this brick (contiguous in w when c0 + bc <= C)
    const int t = threadIdx.x;
    for (int e = t; e < kBrickN * run; e += 256) {
        const int nn = e / run, k = e - nn * run;          // k = cc*RS + rs
        const int n = n0 + nn, c = c0 + k / RS;
        brick[nn][k] = (n < N && c < C) ? jb.w[((size_t)n * C + c0) * RS + k] : 0.f;
    }
    __syncthreads();
    T* wf = reinterpret_cast<T*>(jb.wf);
    T* wd = reinterpret_cast<T*>(jb.wd);
    if constexpr (sizeof(T) == 2) {
        // 16-bit packs with whole 16-byte chunks on both axes (every pack of the networks here): a thread converts EIGHT values and
        // stores one chunk -- the element-wise loops below issue eight 2-byte stores for it (round 6: 227 -> see profiles/r06_experiments.md
        // us per step for the 124 + 8 packs of the bench network).  Same values, same rounding: the packs are bit-identical.
        if ((Cpad & 7) == 0 && (Npad & 7) == 0) {
            const int cg = bc >> 3;                 // chunks of 8 input channels per brick: 1 (3x3) or 8 (pointwise)
            for (int e = t; e < kBrickN * RS * cg; e += 256) {          // forward pack wf[n][rs][c .. c+7]
                const int g = e % cg, rest = e / cg;
                const int rs = rest % RS, nn = rest / RS;
                const int n = n0 + nn, c = c0 + g * 8;
                if (n < Npad && c < Cpad) {
                    T v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = from_f<T>(brick[nn][(g * 8 + u) * RS + rs]);
                    uint4 q;
                    __builtin_memcpy(&q, v, 16);
                    *reinterpret_cast<uint4*>(wf + ((size_t)n * RS + rs) * Cpad + c) = q;
                }
            }
            for (int e = t; e < bc * RS * (kBrickN / 8); e += 256) {     // dgrad pack wd[c][flipped rs][n .. n+7]
                const int ng = e % (kBrickN / 8), rest = e / (kBrickN / 8);
                const int rs = rest % RS, cc = rest / RS;
                const int n = n0 + ng * 8, c = c0 + cc;
                if (n < Npad && c < C) {
                    T v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = from_f<T>(brick[ng * 8 + u][cc * RS + rs]);
                    uint4 q;
                    __builtin_memcpy(&q, v, 16);
                    *reinterpret_cast<uint4*>(wd + ((size_t)c * RS + (RS - 1 - rs)) * Npad + n) = q;
                }
            }
            return;
        }
    }
    // forward pack: wf[n][rs][c]  (c fastest over bc consecutive threads)
    for (int e = t; e < kBrickN * run; e += 256) {
        const int cc = e % bc, rest = e / bc;
        const int rs = rest % RS, nn = rest / RS;
        const int n = n0 + nn, c = c0 + cc;
        if (n < Npad && c < Cpad) wf[((size_t)n * RS + rs) * Cpad + c] = from_f<T>(brick[nn][cc * RS + rs]);
    }
    // dgrad pack: wd[c][flipped rs][n]  (n fastest over 64 consecutive threads); only real input channels have rows
    for (int e = t; e < kBrickN * run; e += 256) {
        const int nn = e % kBrickN, rest = e / kBrickN;
        const int rs = rest % RS, cc = rest / RS;
        const int n = n0 + nn, c = c0 + cc;
        if (n < Npad && c < C) wd[((size_t)c * RS + (RS - 1 - rs)) * Npad + n] = from_f<T>(brick[nn][cc * RS + rs]);
    }
}

// network input: NCHW fp32 [B,C,H,W] -> NHWC T [B,H,W,Cpad] (pad channels zero)
template <typename T>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int H, int W, int Cpad) {
    const int64_t npix = (int64_t)B * H * W;
    constexpr int EPC = 16 / (int)sizeof(T);
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = pix / ((int64_t)H * W), hw = pix % ((int64_t)H * W);
        if (Cpad == EPC && C <= EPC) {        // the network input (3 -> one 16-byte chunk per pixel): ONE store instead of Cpad
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int c = 0; c < EPC; ++c)
                if (c < C) chunk_set<T>(v, c, from_f<T>(x[(b * C + c) * (int64_t)H * W + hw]));
            *reinterpret_cast<uint4*>(y + pix * Cpad) = v;
            continue;
        }
        for (int c = 0; c < Cpad; ++c) {
            const float v = c < C ? x[(b * C + c) * (int64_t)H * W + hw] : 0.f;
            y[pix * Cpad + c] = from_f<T>(v);
        }
    }
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int mrfp_pack_weight(const float* w, void* wf, void* wd, int dtype, int64_t N, int64_t C, int64_t R, int64_t S,
                     int64_t Npad, int64_t Cpad, void* stream) {
    MRFP_CHECK(w && (wf || wd) && N > 0 && C > 0 && R > 0 && S > 0 && Npad >= N && Cpad >= C, "pack_weight: bad arguments");
    const int64_t total = Npad * R * S * Cpad + C * R * S * Npad;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((pack_weight_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (float*)wf, (float*)wd, (int)N, (int)C, (int)R, (int)S, (int)Npad, (int)Cpad);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((pack_weight_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (bf16*)wf, (bf16*)wd, (int)N, (int)C, (int)R, (int)S, (int)Npad, (int)Cpad);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((pack_weight_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (f16*)wf, (f16*)wd, (int)N, (int)C, (int)R, (int)S, (int)Npad, (int)Cpad);
    else
        MRFP_CHECK(false, "pack_weight: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_pack_weights_batched(const void* jobs, const int64_t* prefix, int64_t njobs, int64_t total, int dtype, void* stream) {
    MRFP_CHECK(jobs && prefix && njobs > 0 && total > 0, "pack_weights_batched: bad arguments");   /* jobs with R*S > 9 are the caller's error */
    const int64_t blocks = total;           // one workgroup per brick
    MRFP_CHECK(blocks < (1LL << 31), "pack_weights_batched: too many bricks");
    const mrfp::PackJob* jb = (const mrfp::PackJob*)jobs;
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((pack_weights_batched_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jb, prefix, (int)njobs, total);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((pack_weights_batched_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jb, prefix, (int)njobs, total);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((pack_weights_batched_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jb, prefix, (int)njobs, total);
    else
        MRFP_CHECK(false, "pack_weights_batched: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_nchw_to_nhwc_pad(const float* x, void* y, int dtype, int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cpad,
                          void* stream) {
    MRFP_CHECK(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_to_nhwc_pad: bad arguments");
    int64_t blocks = (B * H * W + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                           (float*)y, (int)B, (int)C, (int)H, (int)W, (int)Cpad);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                           (bf16*)y, (int)B, (int)C, (int)H, (int)W, (int)Cpad);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                           (f16*)y, (int)B, (int)C, (int)H, (int)W, (int)Cpad);
    else
        MRFP_CHECK(false, "nchw_to_nhwc_pad: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
