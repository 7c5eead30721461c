// fft.hip -- per-feature-map 2-D real FFT / inverse FFT with a low/high-band amplitude mix between a sample
// and a partner sample of the batch (the "Fourier amplitude perturbation" of BASELINE.json's north_star; the
// reference has no such function in its model -- nearest arithmetic: dataloaders.py:24-79 HPF/LPF/PHOT --
// so the semantics are BUILD-DEFINED, see DESIGN.md section 8 and oracle/mrfp_oracle.py::fourier_amplitude_mix).
//
//   F = rfft2(x[b,c]) ; A = |F| ; A' = |rfft2(x[perm[b],c])|
//   sel(kh,kw) = (min(kh,H-kh)^2 + kw^2 <= r^2)   (low band; complemented for the high band)
//   ratio = sel ? ((1-lam)*A + lam*A') / A : 1        (1 where A == 0)
//   y = irfft2(F * ratio)                              backward: dx = irfft2(rfft2(dy) * ratio)  (ratio detached)
//
// NHWC activations: a plane is strided by C, so the transform is done as line FFTs over tiles of
// [N points][16 channels] staged in LDS (the 16 channels of a pixel are one 32/64-byte run; consecutive lanes
// own consecutive channels, so every LDS access of the Stockham butterflies is conflict-free):
//   pass 0  rows    x (real)        -> S  (half spectrum along W)
//   pass 1  columns S               -> S  (in place, along H)
//   pass 2  columns S, S[perm], mix -> S3 (inverse along H, ratio computed or loaded, optionally stored)
//   pass 3  rows    S3              -> y  (Hermitian-extended inverse along W, real part, 1/(H*W))
// Radix-2 / radix-3 Stockham autosort stages (lengths 2^a 3^b <= 512), twiddles from a host-built table.
#include "common.hpp"

namespace mrfp {

constexpr int kCB = 16;        // channels per tile
constexpr int kMaxStages = 12;

struct FftP {
    const void* x;        // pass 0: real input [B,H,W,C] (T); pass 3: unused
    void* y;              // pass 3: real output [B,H,W,C] (T)
    float2* S;            // spectrum [B,H,Wh,C]
    float2* S3;           // mixed / inverse-column buffer [B,H,Wh,C]
    float* ratio;         // [B,H,Wh,C] or null
    const int64_t* perm;  // [B] or null
    const float2* tw;     // twiddle table of the line length: exp(-2 pi i t / N), t < N
    int B, H, W, Wh, C;
    int N;                // line length of this pass
    int nstages;
    int radix[kMaxStages];
    float radius2, lam, scale;
    int high, load_ratio;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-LDS Stockham FFT of p.N points for kCB channels; returns the buffer (0/1) holding the result
__device__ __forceinline__ int stockham(float2* buf0, float2* buf1, const FftP& p) {
    const int ch = threadIdx.x % kCB, lane = threadIdx.x / kCB, nl = kThreads / kCB;
    float2* in = buf0;
    float2* out = buf1;
    int Ns = 1, cur = 0;
    for (int s = 0; s < p.nstages; ++s) {
        const int R = p.radix[s];
        const int nb = p.N / R;
        const int tstep = p.N / (Ns * R);
        for (int j = lane; j < nb; j += nl) {
            const int k = j % Ns;
            const int j0 = (j / Ns) * Ns * R + k;
            if (R == 2) {
                const float2 a = in[j * kCB + ch];
                const float2 b = cmul(in[(j + nb) * kCB + ch], p.tw[(k * tstep) % p.N]);
                out[j0 * kCB + ch] = make_float2(a.x + b.x, a.y + b.y);
                out[(j0 + Ns) * kCB + ch] = make_float2(a.x - b.x, a.y - b.y);
            } else {   // radix 3: w3 = exp(-2 pi i / 3) = (-1/2, -sqrt(3)/2)
                const float2 a = in[j * kCB + ch];
                const float2 b = cmul(in[(j + nb) * kCB + ch], p.tw[(k * tstep) % p.N]);
                const float2 c = cmul(in[(j + 2 * nb) * kCB + ch], p.tw[(2 * k * tstep) % p.N]);
                const float2 sbc = make_float2(b.x + c.x, b.y + c.y), dbc = make_float2(b.x - c.x, b.y - c.y);
                const float h = 0.8660254037844386f;
                const float2 m = make_float2(a.x - 0.5f * sbc.x, a.y - 0.5f * sbc.y);
                out[j0 * kCB + ch] = make_float2(a.x + sbc.x, a.y + sbc.y);
                out[(j0 + Ns) * kCB + ch] = make_float2(m.x + h * dbc.y, m.y - h * dbc.x);        // m - i*h*dbc
                out[(j0 + 2 * Ns) * kCB + ch] = make_float2(m.x - h * dbc.y, m.y + h * dbc.x);    // m + i*h*dbc
            }
        }
        __syncthreads();
        float2* t = in; in = out; out = t;
        cur ^= 1;
        Ns *= R;
    }
    return cur;
}

template <typename T, int PASS>
__global__ __launch_bounds__(kThreads) void fft_pass_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf0 = reinterpret_cast<float2*>(smem);
    float2* buf1 = buf0 + (size_t)p.N * kCB;
    const int ch = threadIdx.x % kCB, lane = threadIdx.x / kCB, nl = kThreads / kCB;
    const int c0 = blockIdx.y * kCB;
    const int line = blockIdx.x;                      // pass 0/3: b*H + h ; pass 1/2: b*Wh + kw
    const size_t C = p.C;

    if (PASS == 0) {
        const T* src = reinterpret_cast<const T*>(p.x) + (size_t)line * p.W * C + c0 + ch;
        for (int n = lane; n < p.N; n += nl) buf0[n * kCB + ch] = make_float2(to_f(src[(size_t)n * C]), 0.f);
        __syncthreads();
        const int r = stockham(buf0, buf1, p);
        const float2* res = r ? buf1 : buf0;
        float2* dst = p.S + (size_t)line * p.Wh * C + c0 + ch;
        for (int n = lane; n < p.Wh; n += nl) dst[(size_t)n * C] = res[n * kCB + ch];
    } else if (PASS == 1) {
        const int b = line / p.Wh, kw = line - b * p.Wh;
        float2* col = p.S + ((size_t)b * p.H * p.Wh + kw) * C + c0 + ch;
        const size_t hs = (size_t)p.Wh * C;
        for (int n = lane; n < p.N; n += nl) buf0[n * kCB + ch] = col[(size_t)n * hs];
        __syncthreads();
        const int r = stockham(buf0, buf1, p);
        const float2* res = r ? buf1 : buf0;
        for (int n = lane; n < p.N; n += nl) col[(size_t)n * hs] = res[n * kCB + ch];
    } else if (PASS == 2) {
        const int b = line / p.Wh, kw = line - b * p.Wh;
        const size_t hs = (size_t)p.Wh * C;
        const size_t off = ((size_t)b * p.H * p.Wh + kw) * C + c0 + ch;
        const float2* own = p.S + off;
        const int64_t pb = p.perm ? p.perm[b] : b;
        const float2* par = p.S + ((size_t)pb * p.H * p.Wh + kw) * C + c0 + ch;
        float* rat = p.ratio ? p.ratio + off : nullptr;
        for (int n = lane; n < p.N; n += nl) {
            const float2 f = own[(size_t)n * hs];
            float rr;
            if (p.load_ratio) {
                rr = rat[(size_t)n * hs];
            } else {
                const int dh = n < p.H - n ? n : p.H - n;
                const bool band = (float)(dh * dh + kw * kw) <= p.radius2;
                rr = 1.f;
                if (band != (p.high != 0)) {
                    const float2 g = par[(size_t)n * hs];
                    const float a = sqrtf(f.x * f.x + f.y * f.y), a2 = sqrtf(g.x * g.x + g.y * g.y);
                    if (a > 1e-20f) rr = ((1.f - p.lam) * a + p.lam * a2) / a;
                }
                if (rat) rat[(size_t)n * hs] = rr;
            }
            buf0[n * kCB + ch] = make_float2(f.x * rr, -f.y * rr);      // conj: inverse = conj(fft(conj(.)))
        }
        __syncthreads();
        const int r = stockham(buf0, buf1, p);
        const float2* res = r ? buf1 : buf0;
        float2* dst = p.S3 + off;
        for (int n = lane; n < p.N; n += nl) {
            const float2 v = res[n * kCB + ch];
            dst[(size_t)n * hs] = make_float2(v.x, -v.y);
        }
    } else {   // PASS 3: Hermitian extension of the half spectrum, inverse along W, real part
        const float2* src = p.S3 + (size_t)line * p.Wh * C + c0 + ch;
        for (int n = lane; n < p.N; n += nl) {
            float2 v;
            if (n < p.Wh) {
                v = src[(size_t)n * C];
                v.y = -v.y;                                               // conj for the inverse
            } else {
                v = src[(size_t)(p.N - n) * C];                           // X[N-k] = conj(X[k]); conj again -> as is
            }
            buf0[n * kCB + ch] = v;
        }
        __syncthreads();
        const int r = stockham(buf0, buf1, p);
        const float2* res = r ? buf1 : buf0;
        T* dst = reinterpret_cast<T*>(p.y) + (size_t)line * p.W * C + c0 + ch;
        for (int n = lane; n < p.N; n += nl) dst[(size_t)n * C] = from_f<T>(res[n * kCB + ch].x * p.scale);
    }
}

static bool factor(int n, int* radix, int& ns) {
    ns = 0;
    while (n % 3 == 0) { radix[ns++] = 3; n /= 3; }
    while (n % 2 == 0) { radix[ns++] = 2; n /= 2; }
    return n == 1 && ns <= kMaxStages;
}

template <typename T, int PASS>
static int launch_pass(FftP p, int N, const float2* tw, int nlines, hipStream_t st) {
    p.N = N;
    p.tw = tw;
    if (!factor(N, p.radix, p.nstages)) {
        set_error("fourier_mix: line length %d is not of the form 2^a 3^b", N);
        return -1;
    }
    const int lds = 2 * N * kCB * (int)sizeof(float2);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_pass_kernel<T, PASS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * kCB * (int)sizeof(float2));
        attr_set = true;
    }
    hipLaunchKernelGGL((fft_pass_kernel<T, PASS>), dim3((unsigned)nlines, (unsigned)(p.C / kCB)), dim3(kThreads), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int run_mix(FftP p, const float2* twH, const float2* twW, hipStream_t st) {
    int rc;
    if ((rc = launch_pass<T, 0>(p, p.W, twW, p.B * p.H, st))) return rc;
    if ((rc = launch_pass<T, 1>(p, p.H, twH, p.B * p.Wh, st))) return rc;
    if ((rc = launch_pass<T, 2>(p, p.H, twH, p.B * p.Wh, st))) return rc;
    return launch_pass<T, 3>(p, p.W, twW, p.B * p.H, st);
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int64_t mrfp_fourier_spectrum_bytes(int64_t B, int64_t H, int64_t W, int64_t C) { return B * H * (W / 2 + 1) * C * 8; }

int mrfp_fourier_mix(const void* x, void* y, const int64_t* perm, void* S, void* S3, float* ratio, int load_ratio,
                     const void* twH, const void* twW, int dtype, int64_t B, int64_t H, int64_t W, int64_t C,
                     float radius, float lam, int high, void* stream) {
    MRFP_CHECK(x && y && S && S3 && twH && twW && B > 0 && H > 1 && W > 1 && C > 0, "fourier_mix: bad arguments");
    MRFP_CHECK(C % kCB == 0, "fourier_mix: C=%lld must be a multiple of %d", (long long)C, kCB);
    MRFP_CHECK(H <= 512 && W <= 512 && W % 2 == 0, "fourier_mix: plane %lldx%lld unsupported (<= 512, even W)", (long long)H, (long long)W);
    MRFP_CHECK(!load_ratio || ratio, "fourier_mix: load_ratio without a ratio buffer");
    FftP p;
    p.x = x; p.y = y; p.S = (float2*)S; p.S3 = (float2*)S3; p.ratio = ratio; p.perm = perm; p.tw = nullptr;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.Wh = (int)(W / 2 + 1); p.C = (int)C; p.N = 0; p.nstages = 0;
    p.radius2 = radius * radius; p.lam = lam; p.scale = 1.0f / (float)(H * W); p.high = high; p.load_ratio = load_ratio;
    if (dtype == MRFP_F32) return run_mix<float>(p, (const float2*)twH, (const float2*)twW, (hipStream_t)stream);
    if (dtype == MRFP_BF16) return run_mix<bf16>(p, (const float2*)twH, (const float2*)twW, (hipStream_t)stream);
    MRFP_CHECK(false, "fourier_mix: unknown dtype %d", dtype);
}

}  // extern "C"
