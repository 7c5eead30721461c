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
    else
        MRFP_CHECK(false, "argmax_hist: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
