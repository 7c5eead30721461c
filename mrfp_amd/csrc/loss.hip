// loss.hip -- CrossEntropyLoss(ignore_index) fwd/bwd over NHWC logits, and the eval-side
// arg-max + confusion histogram.  HBM-bound: logits are read once per pass.
//
// Replaces (reference): nn.CrossEntropyLoss(ignore_index=255) (main.py:822, deepv3.py:363) and the
// host-side np.argmax + np.bincount of the eval loop (main.py:898-909, metrics.py:122-126).
#include "common.hpp"

namespace mrfp {

constexpr int kCeThreads = 256;
constexpr int kMaxClasses = 64;

template <typename T>
__global__ __launch_bounds__(kCeThreads) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                            int64_t npix, int C, int64_t ignore, float* __restrict__ ws) {
    __shared__ float sm[2][kCeThreads / 64];
    float nll = 0.f, cnt = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * kCeThreads + threadIdx.x; p < npix; p += (int64_t)gridDim.x * kCeThreads) {
        const int64_t tg = target[p];
        if (tg == ignore || tg < 0 || tg >= C) continue;
        const T* l = logits + p * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, to_f(l[c]));
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += __expf(to_f(l[c]) - m);
        nll += (m + __logf(s)) - to_f(l[tg]);
        cnt += 1.f;
    }
    nll = wave_sum(nll);
    cnt = wave_sum(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[0][w] = nll; sm[1][w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < kCeThreads / 64; ++i) { a += sm[0][i]; b += sm[1][i]; }
        ws[2 * blockIdx.x] = a;
        ws[2 * blockIdx.x + 1] = b;
    }
}

__global__ void ce_finalize_kernel(const float* ws, int nblk, float* loss) {
    __shared__ double sa[256], sb[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += ws[2 * i]; b += ws[2 * i + 1]; }
    sa[threadIdx.x] = a;
    sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss[0] = (float)(sa[0] / sb[0]);   // 0/0 -> NaN, as torch does for an all-ignored batch
        loss[1] = (float)sb[0];
    }
}

template <typename T>
__global__ __launch_bounds__(kCeThreads) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                            const float* __restrict__ loss, const float* __restrict__ gscale,
                                                            T* __restrict__ dlogits, int64_t npix, int C, int64_t ignore) {
    const float k = (gscale ? gscale[0] : 1.f) / loss[1];
    for (int64_t p = (int64_t)blockIdx.x * kCeThreads + threadIdx.x; p < npix; p += (int64_t)gridDim.x * kCeThreads) {
        const int64_t tg = target[p];
        const T* l = logits + p * C;
        T* d = dlogits + p * C;
        if (tg == ignore || tg < 0 || tg >= C) {
            for (int c = 0; c < C; ++c) d[c] = from_f<T>(0.f);
            continue;
        }
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, to_f(l[c]));
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += __expf(to_f(l[c]) - m);
        const float inv = 1.f / s;
        for (int c = 0; c < C; ++c) {
            const float pr = __expf(to_f(l[c]) - m) * inv;
            d[c] = from_f<T>((pr - (c == tg ? 1.f : 0.f)) * k);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kCeThreads) void argmax_hist_kernel(const T* __restrict__ logits,
                                                                 const int64_t* __restrict__ target, int64_t npix, int C,
                                                                 unsigned long long* __restrict__ hist,
                                                                 uint8_t* __restrict__ pred) {
    __shared__ unsigned int lh[kMaxClasses * kMaxClasses];
    for (int i = threadIdx.x; i < C * C; i += kCeThreads) lh[i] = 0;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * kCeThreads + threadIdx.x; p < npix; p += (int64_t)gridDim.x * kCeThreads) {
        const T* l = logits + p * C;
        float m = to_f(l[0]);
        int am = 0;
        for (int c = 1; c < C; ++c) {
            const float v = to_f(l[c]);
            if (v > m) { m = v; am = c; }   // first maximum, as np.argmax
        }
        if (pred) pred[p] = (uint8_t)am;
        if (target) {
            const int64_t tg = target[p];
            if (tg >= 0 && tg < C) atomicAdd(&lh[(int)tg * C + am], 1u);
        }
    }
    __syncthreads();
    if (target)
        for (int i = threadIdx.x; i < C * C; i += kCeThreads)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

static int ce_blocks(int64_t npix) {
    int64_t b = (npix + kCeThreads - 1) / kCeThreads;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int64_t mrfp_ce_nblocks(int64_t npix) { return ce_blocks(npix); }

int mrfp_ce_fwd(const void* logits, const int64_t* target, int dtype, int64_t npix, int64_t C, int64_t ignore_index,
                float* ws, float* loss, void* stream) {
    MRFP_CHECK(logits && target && ws && loss && npix > 0 && C > 0, "ce_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(npix);
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((ce_fwd_kernel<float>), dim3(nb), dim3(kCeThreads), 0, st, (const float*)logits, target, npix, (int)C, ignore_index, ws);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((ce_fwd_kernel<bf16>), dim3(nb), dim3(kCeThreads), 0, st, (const bf16*)logits, target, npix, (int)C, ignore_index, ws);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((ce_fwd_kernel<f16>), dim3(nb), dim3(kCeThreads), 0, st, (const f16*)logits, target, npix, (int)C, ignore_index, ws);
    else
        MRFP_CHECK(false, "ce_fwd: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, ws, nb, loss);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_ce_bwd(const void* logits, const int64_t* target, const float* loss, const float* gscale, void* dlogits,
                int dtype, int64_t npix, int64_t C, int64_t ignore_index, void* stream) {
    MRFP_CHECK(logits && target && loss && dlogits && npix > 0 && C > 0, "ce_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(npix);
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3(nb), dim3(kCeThreads), 0, st, (const float*)logits, target, loss, gscale, (float*)dlogits, npix, (int)C, ignore_index);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((ce_bwd_kernel<bf16>), dim3(nb), dim3(kCeThreads), 0, st, (const bf16*)logits, target, loss, gscale, (bf16*)dlogits, npix, (int)C, ignore_index);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((ce_bwd_kernel<f16>), dim3(nb), dim3(kCeThreads), 0, st, (const f16*)logits, target, loss, gscale, (f16*)dlogits, npix, (int)C, ignore_index);
    else
        MRFP_CHECK(false, "ce_bwd: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_argmax_hist(const void* logits, const int64_t* target, int dtype, int64_t npix, int64_t C, int64_t* hist,
                     uint8_t* pred, void* stream) {
    MRFP_CHECK(logits && npix > 0 && C > 0 && C <= kMaxClasses, "argmax_hist: bad arguments (C <= %d)", kMaxClasses);
    MRFP_CHECK(!target || hist, "argmax_hist: target given without hist");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_blocks(npix);
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((argmax_hist_kernel<float>), dim3(nb), dim3(kCeThreads), 0, st, (const float*)logits, target, npix, (int)C, (unsigned long long*)hist, pred);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((argmax_hist_kernel<bf16>), dim3(nb), dim3(kCeThreads), 0, st, (const bf16*)logits, target, npix, (int)C, (unsigned long long*)hist, pred);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((argmax_hist_kernel<f16>), dim3(nb), dim3(kCeThreads), 0, st, (const f16*)logits, target, npix, (int)C, (unsigned long long*)hist, pred);
    else
        MRFP_CHECK(false, "argmax_hist: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// =============================================================================================
// Fused  bilinear upsample (align_corners) of the low-resolution class scores  +  cross entropy.
// In training the reference only needs the scalar loss (deepv3.py:361-365): the full-resolution
// [B,19,H,W] logits are never written; every thread re-interpolates the 4 taps of its pixel from the
// (L2-resident) low-resolution scores.  Backward writes d(logits) once, channel-padded to a 16-byte
// multiple, for the gather-form bilinear backward.
// =============================================================================================
namespace mrfp {

__device__ __forceinline__ float up_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// CP = class count rounded up to a multiple of 8 (compile time): every loop over classes is fully unrolled with a
// `c < C` predicate, so the per-pixel class vector lives in registers (a run-time bound puts it in scratch memory:
// measured 469 / 1060 us per call at 16x768x768x19 before, see profiles/).
template <typename T, int CP>
__device__ __forceinline__ void up_logits(const T* __restrict__ P, int ld, int Hi, int Wi, int H, int W, int C, int b,
                                          int oh, int ow, float (&z)[CP]) {
    const float sh = up_scale(Hi, H), sw = up_scale(Wi, W);
    const float fh = sh * (float)oh, fw = sw * (float)ow;
    const int h0 = (int)fh, w0 = (int)fw;
    const int h1 = h0 + (h0 < Hi - 1 ? 1 : 0), w1 = w0 + (w0 < Wi - 1 ? 1 : 0);
    const float lh1 = fh - (float)h0, lh0 = 1.f - lh1, lw1 = fw - (float)w0, lw0 = 1.f - lw1;
    const T* p00 = P + (((size_t)b * Hi + h0) * Wi + w0) * ld;
    const T* p01 = P + (((size_t)b * Hi + h0) * Wi + w1) * ld;
    const T* p10 = P + (((size_t)b * Hi + h1) * Wi + w0) * ld;
    const T* p11 = P + (((size_t)b * Hi + h1) * Wi + w1) * ld;
    constexpr int EPC = 16 / (int)sizeof(T);
#pragma unroll
    for (int c0 = 0; c0 < CP; c0 += EPC) {
        float a[EPC], bb[EPC], c[EPC], d[EPC];
#pragma unroll
        for (int i = 0; i < EPC; ++i) { a[i] = 0.f; bb[i] = 0.f; c[i] = 0.f; d[i] = 0.f; }
        if (c0 < C) {        // chunks past the class count are never read (the pitch may be shorter than CP)
            load_f<T, EPC>(p00 + c0, a);
            load_f<T, EPC>(p01 + c0, bb);
            load_f<T, EPC>(p10 + c0, c);
            load_f<T, EPC>(p11 + c0, d);
        }
#pragma unroll
        for (int i = 0; i < EPC; ++i) z[c0 + i] = lh0 * (lw0 * a[i] + lw1 * bb[i]) + lh1 * (lw0 * c[i] + lw1 * d[i]);
    }
}

template <typename T, int CP>
__global__ __launch_bounds__(kCeThreads) void upsample_ce_fwd_kernel(const T* __restrict__ P, int ld, const int64_t* __restrict__ target,
                                                                     int B, int Hi, int Wi, int H, int W, int C, int64_t ignore,
                                                                     float* __restrict__ ws) {
    __shared__ float sm[2][kCeThreads / 64];
    const int64_t npix = (int64_t)B * H * W;
    float nll = 0.f, cnt = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * kCeThreads + threadIdx.x; p < npix; p += (int64_t)gridDim.x * kCeThreads) {
        const int64_t tg = target[p];
        if (tg == ignore || tg < 0 || tg >= C) continue;
        const int b = (int)(p / ((int64_t)H * W)), rem = (int)(p - (int64_t)b * H * W);
        const int oh = rem / W, ow = rem - oh * W;
        float z[CP];
        up_logits<T, CP>(P, ld, Hi, Wi, H, W, C, b, oh, ow, z);
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CP; ++c) if (c < C) m = fmaxf(m, z[c]);
        float s = 0.f, zt = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) { s += __expf(z[c] - m); zt = (c == (int)tg) ? z[c] : zt; }
        nll += (m + __logf(s)) - zt;
        cnt += 1.f;
    }
    nll = wave_sum(nll);
    cnt = wave_sum(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[0][w] = nll; sm[1][w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, bsum = 0.f;
        for (int i = 0; i < kCeThreads / 64; ++i) { a += sm[0][i]; bsum += sm[1][i]; }
        ws[2 * blockIdx.x] = a;
        ws[2 * blockIdx.x + 1] = bsum;
    }
}

template <typename T, int CP>
__global__ __launch_bounds__(kCeThreads) void upsample_ce_bwd_kernel(const T* __restrict__ P, int ld, const int64_t* __restrict__ target,
                                                                     const float* __restrict__ loss, const float* __restrict__ gscale,
                                                                     T* __restrict__ dlogits, int Cd, int B, int Hi, int Wi, int H,
                                                                     int W, int C, int64_t ignore) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int64_t npix = (int64_t)B * H * W;
    const float k = (gscale ? gscale[0] : 1.f) / loss[1];
    for (int64_t p = (int64_t)blockIdx.x * kCeThreads + threadIdx.x; p < npix; p += (int64_t)gridDim.x * kCeThreads) {
        const int64_t tg = target[p];
        T* d = dlogits + p * Cd;
        float g[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) g[c] = 0.f;
        const bool valid = !(tg == ignore || tg < 0 || tg >= C);
        if (valid) {
            const int b = (int)(p / ((int64_t)H * W)), rem = (int)(p - (int64_t)b * H * W);
            const int oh = rem / W, ow = rem - oh * W;
            up_logits<T, CP>(P, ld, Hi, Wi, H, W, C, b, oh, ow, g);
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < CP; ++c) if (c < C) m = fmaxf(m, g[c]);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c) { g[c] = c < C ? __expf(g[c] - m) : 0.f; s += g[c]; }
            const float inv = 1.f / s;
#pragma unroll
            for (int c = 0; c < CP; ++c) g[c] = c < C ? (g[c] * inv - (c == (int)tg ? 1.f : 0.f)) * k : 0.f;
        }
#pragma unroll
        for (int c0 = 0; c0 < CP; c0 += EPC) {
            if (c0 < Cd) {
                float o[EPC];
#pragma unroll
                for (int i = 0; i < EPC; ++i) o[i] = g[c0 + i];
                store_f<T, EPC>(d + c0, o);
            }
        }
    }
}

// (A backward in gather form over the LOW-resolution pixels -- the full-resolution gradient never written -- was built and
// measured in round 2: 818 us against 276 + 255 us for the two passes at 16 x 768 x 768 x 19; removed, profiles/r02_experiments.md.)

// dispatch on CP = C rounded up to 8 (8 .. kMaxClasses)
struct UpCeArgs {
    const void* P; int ld; const int64_t* target; const float* loss; const float* gscale; void* dlogits; int Cd;
    int B, Hi, Wi, H, W, C; int64_t ignore; float* ws; int nb; hipStream_t st;
};
template <typename T, int CP>
static void launch_up_ce(const UpCeArgs& a, bool bwd) {
    if (!bwd)
        hipLaunchKernelGGL((upsample_ce_fwd_kernel<T, CP>), dim3(a.nb), dim3(kCeThreads), 0, a.st, (const T*)a.P, a.ld, a.target,
                           a.B, a.Hi, a.Wi, a.H, a.W, a.C, a.ignore, a.ws);
    else
        hipLaunchKernelGGL((upsample_ce_bwd_kernel<T, CP>), dim3(a.nb), dim3(kCeThreads), 0, a.st, (const T*)a.P, a.ld, a.target,
                           a.loss, a.gscale, (T*)a.dlogits, a.Cd, a.B, a.Hi, a.Wi, a.H, a.W, a.C, a.ignore);
}
template <typename T>
static void dispatch_up_ce(const UpCeArgs& a, bool bwd) {
    switch ((a.C + 7) / 8) {
        case 1: launch_up_ce<T, 8>(a, bwd); break;
        case 2: launch_up_ce<T, 16>(a, bwd); break;
        case 3: launch_up_ce<T, 24>(a, bwd); break;
        case 4: launch_up_ce<T, 32>(a, bwd); break;
        case 5: launch_up_ce<T, 40>(a, bwd); break;
        case 6: launch_up_ce<T, 48>(a, bwd); break;
        case 7: launch_up_ce<T, 56>(a, bwd); break;
        default: launch_up_ce<T, 64>(a, bwd); break;
    }
}

}  // namespace mrfp

extern "C" {

int mrfp_upsample_ce_fwd(const void* P, int64_t ld, const int64_t* target, int dtype, int64_t B, int64_t Hi, int64_t Wi,
                         int64_t H, int64_t W, int64_t C, int64_t ignore_index, float* ws, float* loss, void* stream) {
    MRFP_CHECK(P && target && ws && loss && B > 0 && Hi > 0 && Wi > 0 && H > 0 && W > 0 && C > 0 && C <= mrfp::kMaxClasses,
               "upsample_ce_fwd: bad arguments");
    const int esz = dtype == MRFP_F32 ? 4 : 2, epc = 16 / esz;
    MRFP_CHECK(dtype == MRFP_F32 || dtype == MRFP_BF16 || dtype == MRFP_F16, "upsample_ce_fwd: unknown dtype %d", dtype);
    MRFP_CHECK(ld % epc == 0 && ld >= (C + epc - 1) / epc * epc && mrfp::aligned16(P),
               "upsample_ce_fwd: the score buffer must be channel-padded to 16-byte chunks (ld=%lld)", (long long)ld);
    hipStream_t st = (hipStream_t)stream;
    const int nb = mrfp::ce_blocks(B * H * W);
    mrfp::UpCeArgs a{P, (int)ld, target, nullptr, nullptr, nullptr, 0, (int)B, (int)Hi, (int)Wi, (int)H, (int)W, (int)C,
                     ignore_index, ws, nb, st};
    if (dtype == MRFP_F32) mrfp::dispatch_up_ce<float>(a, false);
    else if (dtype == MRFP_F16) mrfp::dispatch_up_ce<mrfp::f16>(a, false);
    else mrfp::dispatch_up_ce<mrfp::bf16>(a, false);
    MRFP_LAUNCH_CHECK();
    hipLaunchKernelGGL(mrfp::ce_finalize_kernel, dim3(1), dim3(256), 0, st, ws, nb, loss);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_upsample_ce_bwd(const void* P, int64_t ld, const int64_t* target, const float* loss, const float* gscale,
                         void* dlogits, int64_t Cd, int dtype, int64_t B, int64_t Hi, int64_t Wi, int64_t H, int64_t W,
                         int64_t C, int64_t ignore_index, void* stream) {
    MRFP_CHECK(P && target && loss && dlogits && B > 0 && Hi > 0 && Wi > 0 && H > 0 && W > 0 && C > 0 && C <= mrfp::kMaxClasses,
               "upsample_ce_bwd: bad arguments");
    const int esz = dtype == MRFP_F32 ? 4 : 2, epc = 16 / esz;
    MRFP_CHECK(dtype == MRFP_F32 || dtype == MRFP_BF16 || dtype == MRFP_F16, "upsample_ce_bwd: unknown dtype %d", dtype);
    MRFP_CHECK(ld % epc == 0 && Cd % epc == 0 && Cd >= C && ld >= Cd && mrfp::aligned16(P) && mrfp::aligned16(dlogits),
               "upsample_ce_bwd: channel pitches must be 16-byte multiples (ld=%lld Cd=%lld)", (long long)ld, (long long)Cd);
    hipStream_t st = (hipStream_t)stream;
    const int nb = mrfp::ce_blocks(B * H * W);
    mrfp::UpCeArgs a{P, (int)ld, target, loss, gscale, dlogits, (int)Cd, (int)B, (int)Hi, (int)Wi, (int)H, (int)W, (int)C,
                     ignore_index, nullptr, nb, st};
    if (dtype == MRFP_F32) mrfp::dispatch_up_ce<float>(a, true);
    else if (dtype == MRFP_F16) mrfp::dispatch_up_ce<mrfp::f16>(a, true);
    else mrfp::dispatch_up_ce<mrfp::bf16>(a, true);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
