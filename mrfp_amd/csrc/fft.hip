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
// Low band, bf16 activations: the two ROW passes of the band-limited path are matrix products on the MFMA units
// (dft_rows_fwd_mfma_kernel / dft_rows_inv_mfma_kernel below; MRFP_FFT_MFMA=0 switches back to the FFT / direct-sum passes).
#include "common.hpp"
#include <type_traits>

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
    int Ws;               // stored bins along W of S / S3 / ratio: Wh (full spectrum) or floor(radius)+1 (band-limited path)
    int delta;            // band-limited path: S3 holds F*(ratio-1) and the inverse row pass writes x + irfft(.)
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

// =============================================================================================
// Fast path: line lengths N = N1*N2 (N2 = 16 or 32) as a TWO-STEP transform held in registers.
//   step A   thread (channel, n2) owns x[n1*N2 + n2], n1 < N1: an N1-point DFT in registers, then the twiddle W_N^(n2*k1)
//   exchange one pass through LDS (the only one per line transform), padded pitch -> conflict-free 8-byte accesses
//   step B   thread (channel, k1) owns the N2 values of row k1: an N2-point DFT in registers -> X[k1 + N1*k2]
// Three launches instead of four, and the column launch does forward transform, amplitude mix and inverse transform
// on one tile (the inverse starts from step B's register layout, so there is no exchange between them):
//   rows      x  -> S          read 1 plane, write the half spectrum
//   columns   S (+ the partner's column where the band reaches it) -> S3     (ratio written / read here)
//   rows^-1   S3 -> y
// Workgroups that share cache lines of the NHWC activation (the channel groups of one line) are numbered so that
// they run back to back on the same XCD (one L2).
// =============================================================================================
__device__ __forceinline__ float2 tw96(int j) {       // exp(-2 pi i j / 96); j is a constant after unrolling
    constexpr float kc[96] = {
            1.f, 0.997858923f, 0.991444861f, 0.98078528f, 0.965925826f, 0.946930129f, 0.923879533f, 0.896872742f,
            0.866025404f, 0.831469612f, 0.79335334f, 0.751839807f, 0.707106781f, 0.659345815f, 0.608761429f, 0.555570233f,
            0.5f, 0.44228869f, 0.382683432f, 0.321439465f, 0.258819045f, 0.195090322f, 0.130526192f, 0.0654031292f,
            0.f, -0.0654031292f, -0.130526192f, -0.195090322f, -0.258819045f, -0.321439465f, -0.382683432f, -0.44228869f,
            -0.5f, -0.555570233f, -0.608761429f, -0.659345815f, -0.707106781f, -0.751839807f, -0.79335334f, -0.831469612f,
            -0.866025404f, -0.896872742f, -0.923879533f, -0.946930129f, -0.965925826f, -0.98078528f, -0.991444861f, -0.997858923f,
            -1.f, -0.997858923f, -0.991444861f, -0.98078528f, -0.965925826f, -0.946930129f, -0.923879533f, -0.896872742f,
            -0.866025404f, -0.831469612f, -0.79335334f, -0.751839807f, -0.707106781f, -0.659345815f, -0.608761429f, -0.555570233f,
            -0.5f, -0.44228869f, -0.382683432f, -0.321439465f, -0.258819045f, -0.195090322f, -0.130526192f, -0.0654031292f,
            0.f, 0.0654031292f, 0.130526192f, 0.195090322f, 0.258819045f, 0.321439465f, 0.382683432f, 0.44228869f,
            0.5f, 0.555570233f, 0.608761429f, 0.659345815f, 0.707106781f, 0.751839807f, 0.79335334f, 0.831469612f,
            0.866025404f, 0.896872742f, 0.923879533f, 0.946930129f, 0.965925826f, 0.98078528f, 0.991444861f, 0.997858923f};
    constexpr float ks[96] = {
            0.f, -0.0654031292f, -0.130526192f, -0.195090322f, -0.258819045f, -0.321439465f, -0.382683432f, -0.44228869f,
            -0.5f, -0.555570233f, -0.608761429f, -0.659345815f, -0.707106781f, -0.751839807f, -0.79335334f, -0.831469612f,
            -0.866025404f, -0.896872742f, -0.923879533f, -0.946930129f, -0.965925826f, -0.98078528f, -0.991444861f, -0.997858923f,
            -1.f, -0.997858923f, -0.991444861f, -0.98078528f, -0.965925826f, -0.946930129f, -0.923879533f, -0.896872742f,
            -0.866025404f, -0.831469612f, -0.79335334f, -0.751839807f, -0.707106781f, -0.659345815f, -0.608761429f, -0.555570233f,
            -0.5f, -0.44228869f, -0.382683432f, -0.321439465f, -0.258819045f, -0.195090322f, -0.130526192f, -0.0654031292f,
            0.f, 0.0654031292f, 0.130526192f, 0.195090322f, 0.258819045f, 0.321439465f, 0.382683432f, 0.44228869f,
            0.5f, 0.555570233f, 0.608761429f, 0.659345815f, 0.707106781f, 0.751839807f, 0.79335334f, 0.831469612f,
            0.866025404f, 0.896872742f, 0.923879533f, 0.946930129f, 0.965925826f, 0.98078528f, 0.991444861f, 0.997858923f,
            1.f, 0.997858923f, 0.991444861f, 0.98078528f, 0.965925826f, 0.946930129f, 0.923879533f, 0.896872742f,
            0.866025404f, 0.831469612f, 0.79335334f, 0.751839807f, 0.707106781f, 0.659345815f, 0.608761429f, 0.555570233f,
            0.5f, 0.44228869f, 0.382683432f, 0.321439465f, 0.258819045f, 0.195090322f, 0.130526192f, 0.0654031292f};
    return make_float2(kc[j], ks[j]);
}

template <int N> struct Dft {      // in-place forward DFT of v[0..N), natural order in and out; N = 2^a * {1,3}
    static __device__ __forceinline__ void run(float2 (&v)[N]) {
        static_assert(N % 2 == 0 && 96 % N == 0, "radix-2 split over the 96-entry table");
        float2 e[N / 2], o[N / 2];
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
        Dft<N / 2>::run(e);
        Dft<N / 2>::run(o);
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            const float2 t = (k == 0) ? o[0] : cmul(o[k], tw96(k * (96 / N)));
            v[k] = make_float2(e[k].x + t.x, e[k].y + t.y);
            v[k + N / 2] = make_float2(e[k].x - t.x, e[k].y - t.y);
        }
    }
};
template <> struct Dft<1> { static __device__ __forceinline__ void run(float2 (&)[1]) {} };
template <> struct Dft<3> {
    static __device__ __forceinline__ void run(float2 (&v)[3]) {
        const float h = 0.8660254037844386f;
        const float2 a = v[0], s = make_float2(v[1].x + v[2].x, v[1].y + v[2].y), d = make_float2(v[1].x - v[2].x, v[1].y - v[2].y);
        const float2 m = make_float2(a.x - 0.5f * s.x, a.y - 0.5f * s.y);
        v[0] = make_float2(a.x + s.x, a.y + s.y);
        v[1] = make_float2(m.x + h * d.y, m.y - h * d.x);
        v[2] = make_float2(m.x - h * d.y, m.y + h * d.x);
    }
};

constexpr int two_nt(int n1, int n2) { return kCB * (n1 > n2 ? n1 : n2); }

template <int N1, int N2> struct TwoStep {
    static constexpr int N = N1 * N2;
    static constexpr int NMAX = N1 > N2 ? N1 : N2;
    static constexpr int NT = kCB * NMAX;                       // threads per workgroup
    static constexpr int PITCH = N2 * kCB + kCB;                // float2 per exchange row (+128 B: rows alternate bank halves)
    static constexpr int PITCH_T = N1 * kCB + kCB;              // the transposed (inverse) exchange
    static constexpr int LDS = (N1 * PITCH > N2 * PITCH_T ? N1 * PITCH : N2 * PITCH_T) * (int)sizeof(float2);
};

// a[n1] = x[n1*N2 + sub] held by thread (ch, sub < N2)  ->  b[k2] = X[sub + N1*k2] held by thread (ch, sub < N1)
template <int N1, int N2>
__device__ __forceinline__ void two_step(float2 (&a)[N1], float2 (&b)[N2], float2* lds, const float2* __restrict__ tw, int ch, int sub) {
    constexpr int PITCH = N2 * kCB + kCB;
    if (sub < N2) {
        Dft<N1>::run(a);
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) lds[k1 * PITCH + sub * kCB + ch] = k1 == 0 ? a[0] : cmul(a[k1], tw[sub * k1]);
    }
    __syncthreads();
    if (sub < N1) {
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) b[n2] = lds[sub * PITCH + n2 * kCB + ch];
        Dft<N2>::run(b);
    }
}

static int fft_nopair() {        // MRFP_FFT_NOPAIR=1: band-limited row passes one channel per transform (A/B)
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRFP_FFT_NOPAIR"); v = e ? atoi(e) : 0; }
    return v;
}

static int fft_nodirect() {      // MRFP_FFT_NODIRECT=1: band-limited inverse row pass on the register FFT (A/B)
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRFP_FFT_NODIRECT"); v = e ? atoi(e) : 0; }
    return v;
}

// workgroup id -> (line, channel group): the channel groups of one line are consecutive on one XCD
__device__ __forceinline__ bool decode_wg(int nlines, int ncg, int& line, int& cg) {
    const int wg = blockIdx.x, xcd = wg & 7, i = wg >> 3;
    cg = i % ncg;
    line = (i / ncg) * 8 + xcd;
    return line < nlines;
}

template <typename T, int N1, int N2>
__global__ __launch_bounds__(two_nt(N1, N2)) void fft_rows_fwd_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    const int ch = threadIdx.x % kCB, sub = threadIdx.x / kCB;
    int line, cg;
    if (!decode_wg(p.B * p.H, p.C / kCB, line, cg)) return;
    const size_t C = p.C;
    float2 a[N1], b[N2];
    if (sub < N2) {
        const T* src = reinterpret_cast<const T*>(p.x) + (size_t)line * p.W * C + cg * kCB + ch;
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) a[n1] = make_float2(to_f(src[(size_t)(n1 * N2 + sub) * C]), 0.f);
    }
    two_step<N1, N2>(a, b, lds, p.tw, ch, sub);
    if (sub < N1) {
        float2* dst = p.S + (size_t)line * p.Ws * C + cg * kCB + ch;
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            const int kw = sub + N1 * k2;
            if (kw < p.Ws) dst[(size_t)kw * C] = b[k2];
        }
    }
}

template <typename T, int N1, int N2>
__global__ __launch_bounds__(two_nt(N1, N2)) void fft_rows_inv_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    float2* stage = lds + TwoStep<N1, N2>::LDS / (int)sizeof(float2);     // [Wh][16 channels]: the half spectrum of this line
    const int ch = threadIdx.x % kCB, sub = threadIdx.x / kCB;
    int line, cg;
    if (!decode_wg(p.B * p.H, p.C / kCB, line, cg)) return;
    const size_t C = p.C;
    constexpr int N = N1 * N2, WH = N / 2 + 1, NSUB = two_nt(N1, N2) / kCB;
    const int Ws = p.Ws;                                                 // stored bins (<= WH); the others are zero
    // the stored bins are fetched once; the mirrored half of the Hermitian extension comes out of LDS
    {
        const float2* src = p.S3 + (size_t)line * Ws * C + cg * kCB + ch;
#pragma unroll
        for (int i = 0; i < (WH + NSUB - 1) / NSUB; ++i) {
            const int k = sub + i * NSUB;
            if (k < Ws) stage[k * kCB + ch] = src[(size_t)k * C];
        }
    }
    float xin[N2];
    if (p.delta && sub < N1) {                                           // band-limited path: y = x + correction
        const T* xs = reinterpret_cast<const T*>(p.x) + (size_t)line * p.W * C + cg * kCB + ch;
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) xin[k2] = to_f(xs[(size_t)(sub + N1 * k2) * C]);
    } else {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) xin[k2] = 0.f;
    }
    __syncthreads();
    float2 a[N1], b[N2];
    if (sub < N2) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const int k = n1 * N2 + sub;
            const int km = k < Ws ? k : N - k;                           // k >= Ws: mirrored bin (>= 1)
            float2 v = stage[(km < Ws ? km : 0) * kCB + ch];
            if (k < Ws) v.y = -v.y;                                      // conj(F[k]);  mirrored: conj(conj(F[N-k]))
            if (km >= Ws) v = make_float2(0.f, 0.f);                     // outside the stored band
            a[n1] = v;
        }
    }
    two_step<N1, N2>(a, b, lds, p.tw, ch, sub);
    if (sub < N1) {
        T* dst = reinterpret_cast<T*>(p.y) + (size_t)line * p.W * C + cg * kCB + ch;
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) dst[(size_t)(sub + N1 * k2) * C] = from_f<T>(xin[k2] + b[k2].x * p.scale);
    }
}

// ---- band-limited row passes on CHANNEL PAIRS ---------------------------------------------------
// Two real lines ride in one complex transform (z = x_even + i x_odd): half the butterflies per element and 4/8-byte
// instead of 2/4-byte activation accesses.  A thread is (pair < 16, sub); a tile is 32 channels (C % 32 == 0).
// Forward: Z = FFT(z); the stored bins kw < Ws need Z[kw] and Z[N-kw] (X_even = (Z[k] + conj Z[N-k])/2,
// X_odd = -i (Z[k] - conj Z[N-k])/2), exchanged through the (by then free) LDS buffer; S keeps its [B,H,Ws,C] layout.
template <typename T, int N1, int N2>
__global__ __launch_bounds__(two_nt(N1, N2)) void fft_rows_fwd_pair_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    const int pr = threadIdx.x % kCB, sub = threadIdx.x / kCB;
    int line, cg;
    if (!decode_wg(p.B * p.H, p.C / (2 * kCB), line, cg)) return;
    const size_t C = p.C;
    constexpr int N = N1 * N2, NT = two_nt(N1, N2);
    const int Ws = p.Ws;
    float2 a[N1], b[N2];
    if (sub < N2) {
        const T* src = reinterpret_cast<const T*>(p.x) + (size_t)line * p.W * C + cg * 2 * kCB + 2 * pr;
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            float v[2];
            load_f<T, 2>(src + (size_t)(n1 * N2 + sub) * C, v);
            a[n1] = make_float2(v[0], v[1]);
        }
    }
    two_step<N1, N2>(a, b, lds, p.tw, pr, sub);
    __syncthreads();                                       // every step-B read of the exchange buffer is done
    if (sub < N1) {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            const int k = sub + N1 * k2;                   // slot: k < Ws -> k ; k > N-Ws -> Ws + (N-k) - 1
            if (k < Ws) lds[k * kCB + pr] = b[k2];
            else if (k > N - Ws) lds[(Ws + N - k - 1) * kCB + pr] = b[k2];
        }
    }
    __syncthreads();
    float* dst = reinterpret_cast<float*>(p.S + (size_t)line * Ws * C + cg * 2 * kCB + 2 * pr);
    for (int k = sub; k < Ws; k += NT / kCB) {
        const int km = k == 0 ? 0 : N - k;
        const float2 zk = lds[k * kCB + pr];
        const float2 zm = lds[(km < Ws ? km : Ws + k - 1) * kCB + pr];
        VecT<float, 4> o;
        o.v[0] = 0.5f * (zk.x + zm.x);  o.v[1] = 0.5f * (zk.y - zm.y);      // even channel
        o.v[2] = 0.5f * (zk.y + zm.y);  o.v[3] = -0.5f * (zk.x - zm.x);     // odd channel
        *reinterpret_cast<VecT<float, 4>*>(dst + (size_t)k * C * 2) = o;
    }
}

// Inverse: Z[k] = D_even[k] + i D_odd[k] (k < Ws), Z[N-k] = conj D_even[k] + i conj D_odd[k], zero elsewhere;
// o = FFT(conj Z) -> the even line is Re o, the odd line is -Im o;  y = x + line / (H W).
template <typename T, int N1, int N2>
__global__ __launch_bounds__(two_nt(N1, N2)) void fft_rows_inv_pair_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    float4* stage = reinterpret_cast<float4*>(lds + TwoStep<N1, N2>::LDS / (int)sizeof(float2));   // [Ws][16 pairs]
    const int pr = threadIdx.x % kCB, sub = threadIdx.x / kCB;
    int line, cg;
    if (!decode_wg(p.B * p.H, p.C / (2 * kCB), line, cg)) return;
    const size_t C = p.C;
    constexpr int N = N1 * N2, NT = two_nt(N1, N2);
    const int Ws = p.Ws;
    {
        const float* src = reinterpret_cast<const float*>(p.S3 + (size_t)line * Ws * C + cg * 2 * kCB + 2 * pr);
        for (int k = sub; k < Ws; k += NT / kCB) stage[k * kCB + pr] = *reinterpret_cast<const float4*>(src + (size_t)k * C * 2);
    }
    float xin[N2][2];
    if (sub < N1) {
        const T* xs = reinterpret_cast<const T*>(p.x) + (size_t)line * p.W * C + cg * 2 * kCB + 2 * pr;
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) load_f<T, 2>(xs + (size_t)(sub + N1 * k2) * C, xin[k2]);
    }
    __syncthreads();
    float2 a[N1], b[N2];
    if (sub < N2) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const int k = n1 * N2 + sub;
            const int km = k < Ws ? k : N - k;
            const float4 d = stage[(km < Ws ? km : 0) * kCB + pr];        // (D_even, D_odd) of bin km
            float2 v = k < Ws ? make_float2(d.x - d.w, -d.y - d.z)         // conj(D_even + i D_odd)
                              : make_float2(d.x + d.w, d.y - d.z);         // conj(conj D_even + i conj D_odd)
            if (km >= Ws) v = make_float2(0.f, 0.f);
            a[n1] = v;
        }
    }
    two_step<N1, N2>(a, b, lds, p.tw, pr, sub);
    if (sub < N1) {
        T* dst = reinterpret_cast<T*>(p.y) + (size_t)line * p.W * C + cg * 2 * kCB + 2 * pr;
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            const float o[2] = {xin[k2][0] + b[k2].x * p.scale, xin[k2][1] - b[k2].y * p.scale};
            store_f<T, 2>(dst + (size_t)(sub + N1 * k2) * C, o);
        }
    }
}

// ---- band-limited inverse row pass as a DIRECT trigonometric sum ---------------------------------------------------
// With only Ws stored bins a line is  y[w] = x[w] + sum_k g_k (Re D_k cos th_kw - Im D_k sin th_kw),  th_kw = 2 pi k w / W,
// g_0 = 1, g_k = 2 (Hermitian mirror), scale folded in.  The four positions j, W/2-j, W/2+j, W-j (j <= W/4) share
// |cos| and |sin| up to the signs (-1)^k and +-1, so one pass over the Ws bins -- 2 packed FMAs per bin for a channel
// pair, into (P_even, P_odd, Q_even, Q_odd) -- yields four outputs:
//   y[j] = (Pe+Po) - (Qe+Qo)   y[W-j] = (Pe+Po) + (Qe+Qo)   y[W/2+j] = (Pe-Po) - (Qe-Qo)   y[W/2-j] = (Pe-Po) + (Qe-Qo)
// ~4x fewer instructions than the register FFT of the same line (which transforms W points of which 2 Ws - 1 are
// non-zero).  A thread is (channel pair < 32, slot < 8): 64 channels per workgroup (128-byte bf16 runs), j = slot + 8 i.
// A thread is (channel group, slot): PPT channel pairs per thread (16-byte activation accesses: 4 pairs for the 16-bit
// types, 2 for fp32 -- with one pair per thread the bf16 pass took as long as the fp32 one, i.e. it was bound by the number
// of memory instructions, not by bytes), 64 channels per workgroup, j = slot + NSLOT i.
constexpr int kDirCh = 64, kDirThreads = 256;
static int dir_lines() {         // lines per workgroup of the direct pass (the trig table is built once per workgroup)
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRFP_FFT_DIRLINES"); v = e ? atoi(e) : 1; if (v < 1) v = 1; }
    return v;
}
template <typename T, int PPT, int CH>
__global__ __launch_bounds__(kDirThreads) void dft_rows_inv_direct_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TC = CH / (2 * PPT), NSLOT = kDirThreads / TC, NP = CH / 2;
    const int Ws = p.Ws, W = p.W, NJ = W / 4 + 1;
    float4* stage = reinterpret_cast<float4*>(smem);                         // [Ws][32 pairs]: (Re, Im) of both channels
    float2* trig = reinterpret_cast<float2*>(stage + Ws * NP);               // [NJ][Ws]: g_k scale (cos, sin)(2 pi k j / W)
    const int tc = threadIdx.x % TC, slot = threadIdx.x / TC;
    int lgrp, cg;
    const int nlines = p.B * p.H;
    const int LPW = p.N;                                                     // lines per workgroup
    if (!decode_wg((nlines + LPW - 1) / LPW, p.C / CH, lgrp, cg)) return;
    const size_t C = p.C;
    for (int i = threadIdx.x; i < NJ * Ws; i += kDirThreads) {                // once per workgroup (LPW lines)
        const int j = i / Ws, k = i - j * Ws;
        const float2 t = p.tw[(k * j) % W];                                 // exp(-2 pi i k j / W) = (cos, -sin)
        const float g = (k == 0 ? 1.f : 2.f) * p.scale;
        trig[i] = make_float2(g * t.x, -g * t.y);
    }
    for (int li = 0; li < LPW; ++li) {
    const int line = lgrp * LPW + li;
    if (line >= nlines) break;                                               // workgroup-uniform
    __syncthreads();                                                         // the previous line's reads of `stage` are done
    {
        const float* src = reinterpret_cast<const float*>(p.S3 + (size_t)line * Ws * C + cg * CH);
        for (int i = threadIdx.x; i < Ws * NP; i += kDirThreads) {
            const int k = i / NP, pr = i - k * NP;
            stage[i] = *reinterpret_cast<const float4*>(src + ((size_t)k * C + 2 * pr) * 2);
        }
    }
    __syncthreads();
    const size_t cofs = (size_t)line * W * C + cg * CH + 2 * PPT * tc;
    const T* xs = reinterpret_cast<const T*>(p.x) + cofs;
    T* dst = reinterpret_cast<T*>(p.y) + cofs;
    for (int j = slot; j < NJ; j += NSLOT) {
        const int w0 = j, w1 = W / 2 - j, w2 = W / 2 + j, w3 = j == 0 ? 0 : W - j;
        float x0[2 * PPT], x1[2 * PPT], x2[2 * PPT], x3[2 * PPT];
        load_f<T, 2 * PPT>(xs + (size_t)w0 * C, x0);
        load_f<T, 2 * PPT>(xs + (size_t)w1 * C, x1);
        load_f<T, 2 * PPT>(xs + (size_t)w2 * C, x2);
        load_f<T, 2 * PPT>(xs + (size_t)w3 * C, x3);
        float Pe[2 * PPT], Po[2 * PPT], Qe[2 * PPT], Qo[2 * PPT];
#pragma unroll
        for (int u = 0; u < 2 * PPT; ++u) { Pe[u] = 0.f; Po[u] = 0.f; Qe[u] = 0.f; Qo[u] = 0.f; }
        const float2* tj = trig + j * Ws;
        const float4* sg = stage + PPT * tc;
        for (int k = 0; k < Ws; k += 2) {
            const float2 t0 = tj[k];
            const bool odd = k + 1 < Ws;
            const float2 t1 = odd ? tj[k + 1] : make_float2(0.f, 0.f);
            const int k1 = odd ? k + 1 : k;
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
                const float4 d0 = sg[k * NP + u], d1 = sg[k1 * NP + u];
                Pe[2 * u] = fmaf(d0.x, t0.x, Pe[2 * u]); Pe[2 * u + 1] = fmaf(d0.z, t0.x, Pe[2 * u + 1]);
                Qe[2 * u] = fmaf(d0.y, t0.y, Qe[2 * u]); Qe[2 * u + 1] = fmaf(d0.w, t0.y, Qe[2 * u + 1]);
                Po[2 * u] = fmaf(d1.x, t1.x, Po[2 * u]); Po[2 * u + 1] = fmaf(d1.z, t1.x, Po[2 * u + 1]);
                Qo[2 * u] = fmaf(d1.y, t1.y, Qo[2 * u]); Qo[2 * u + 1] = fmaf(d1.w, t1.y, Qo[2 * u + 1]);
            }
        }
        float o0[2 * PPT], o1[2 * PPT], o2[2 * PPT], o3[2 * PPT];
#pragma unroll
        for (int u = 0; u < 2 * PPT; ++u) {
            const float Ps = Pe[u] + Po[u], Pd = Pe[u] - Po[u], Qs = Qe[u] + Qo[u], Qd = Qe[u] - Qo[u];
            o0[u] = x0[u] + (Ps - Qs);
            o3[u] = x3[u] + (Ps + Qs);
            o2[u] = x2[u] + (Pd - Qd);
            o1[u] = x1[u] + (Pd + Qd);
        }
        const bool inner = j != 0 && 4 * j != W;          // j = 0 and j = W/4: the mirrored positions coincide
        store_f<T, 2 * PPT>(dst + (size_t)w0 * C, o0);
        if (inner || j == 0) store_f<T, 2 * PPT>(dst + (size_t)w2 * C, o2);
        if (inner || 4 * j == W) store_f<T, 2 * PPT>(dst + (size_t)w3 * C, o3);
        if (inner) store_f<T, 2 * PPT>(dst + (size_t)w1 * C, o1);
    }
    }
}

// ---- band-limited inverse row pass on the MATRIX CORES (bf16 activations) -------------------------------------------
// The direct sum above is  y[w][c] = x[w][c] + sum_k T[w][k] G[k][c]  with k = (bin, re / im) < 2 Ws: per image line a
// [W x 2 Ws] x [2 Ws x C] matrix product.  As fp32 vector arithmetic it bounds the pass (profiles/r03_fourier.md: 39 us of
// pure VALU issue at 16 x 128 x 192 x 192); v_mfma_f32_16x16x16_bf16 does it in a fraction of the memory time.  Precision:
// G (fp32, from S3) and the trigonometric matrix T are each split into bf16 hi + lo parts and three products (hi*hi, lo*hi,
// hi*lo) are accumulated in fp32 -- 16 mantissa bits per factor, the dropped lo*lo term is 2^-18 relative.
// Layout: the A operand is G with its 16 rows mapped to channels c0 + 8 (r / 4) + 4 t + r % 4 (t = 0, 1: two 16-row tiles
// = 32 channels), the B operand is T^t for 16 pixels.  The accumulator lane (g, n) then holds pixel w0 + n, channels
// c0 + 8 g .. + 7: one 16-byte access of x and of y per lane, 64-byte runs per pixel.  No LDS: G comes straight from S3
// (8-byte loads, k-contiguous), T is built once per wave from the twiddle table and stays in registers (wave q owns the
// pixel tiles q, q + 4, ...); a workgroup walks consecutive (line, 32-channel block) items.
typedef __attribute__((ext_vector_type(4))) short bfx4;
typedef __attribute__((ext_vector_type(4))) float fx4;
typedef float __attribute__((ext_vector_type(2))) fx2;
typedef __bf16 __attribute__((ext_vector_type(2))) bfx2_t;

// (a, b) -> bf16 hi parts and bf16 lo parts (a - hi(a), b - hi(b)), each pair in one dword
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    fx2 v; v.x = a; v.y = b;
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bfx2_t));
    fx2 r; r.x = a - __uint_as_float(hi << 16); r.y = b - __uint_as_float(hi & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bfx2_t));
}
__device__ __forceinline__ bfx4 as_bfx4(unsigned a, unsigned b) {
    uint2 u = make_uint2(a, b);
    return __builtin_bit_cast(bfx4, u);
}

static int fft_mfma() {          // MRFP_FFT_MFMA=0: band-limited inverse row pass as the fp32 direct sum (A/B)
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRFP_FFT_MFMA"); v = e ? atoi(e) : 1; }
    return v;
}
static int fft_mfma_wgs() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MRFP_FFT_MFMA_WGS"); v = e ? atoi(e) : 768; if (v < 8) v = 8; }
    return v;
}

// LDS image of T^t: for every (part hi / lo, k block kb, lane quarter g) an array over the pixels w of 8-byte entries
// (k = 16 kb + 4 g .. + 3), each array W * 8 + 128 bytes long: a fragment read (16 consecutive pixels x 4 quarters) covers
// 4 x 128 contiguous bytes that start 32 banks apart -- the two passes a 512-byte read needs anyway, no conflicts beyond.
__host__ __device__ inline int trig_pitch(int W) { return W * 8 + 128; }

constexpr int kInvThreads = 256;     // 4 waves share one trigonometric table (40 KB at W = 192): three workgroups per CU

// An item is (image line, block of NH x 32 channels, segment of TU pixel tiles), one wave per item at a time.  NH = 2 wherever C
// allows: the two 64-byte halves of every 128-byte line are then requested together (with 32-channel items they were touched
// an item apart, ~8 us, by which time the line had left the caches: twice the x and y traffic, 99 us instead of the 71 the
// bytes take).  ALL of an item's loads (the G values, then NH x TU 16-byte pieces of x per lane) are issued up front in
// straight-line code behind a sched_barrier, so the compiler's counted s_waitcnt vmcnt(N) lets tile u start when ITS piece has
// landed while the later pieces and the stores of the earlier tiles are still in flight.
template <int KB, int TU, int NH>      // KB: 16-deep k blocks (2 Ws <= 16 KB); TU: pixel tiles per item; NH: 32-channel halves per item
__global__ __launch_bounds__(kInvThreads) void dft_rows_inv_mfma_kernel(FftP p, int items, int ipw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int Ws = p.Ws, W = p.W, C = p.C, ncb = C / (32 * NH), nseg = (W >> 4) / TU;      // (host: TU divides the W / 16 pixel tiles)
    const int pitch = trig_pitch(W);
    // the twiddle table first (one coalesced read; the W-entry gather below would otherwise be rounds of dependent global loads)
    float2* twl = reinterpret_cast<float2*>(smem + 2 * KB * 4 * pitch);
    for (int e = threadIdx.x; e < W; e += kInvThreads) twl[e] = p.tw[e];
    __syncthreads();
    // entry (kb, q, w): k = 16 kb + 4 q + i  <->  bin 8 kb + 2 q + (i >> 1), (i & 1) ? -sin : cos, times g_k / (H W)
    for (int e = threadIdx.x; e < KB * 4 * W; e += kInvThreads) {
        const int w = e % W, kq = e / W, q = kq & 3, kb = kq >> 2;
        float v[4];
        int idx = ((8 * kb + 2 * q) * w) % W;                                // (kw w) mod W, then + w per bin
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kw = 8 * kb + 2 * q + (i >> 1);
            if (i == 2) { idx += w; if (idx >= W) idx -= W; }
            const float2 t = twl[idx];                                       // exp(-2 pi i kw w / W) = (cos, -sin)
            v[i] = kw < Ws ? (kw == 0 ? 1.f : 2.f) * p.scale * ((i & 1) ? t.y : t.x) : 0.f;
        }
        unsigned h0, l0, h1, l1;
        split2(v[0], v[1], h0, l0);
        split2(v[2], v[3], h1, l1);
        *reinterpret_cast<uint2*>(smem + kq * pitch + w * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(smem + (KB * 4 + kq) * pitch + w * 8) = make_uint2(l0, l1);
    }
    __syncthreads();
    const char* th = smem + g * pitch + r * 8;                              // + kb * 4 * pitch + wt * 128
    const char* tl = th + KB * 4 * pitch;
    const int wid = blockIdx.x * (kInvThreads / 64) + wave;
    const int it0 = wid * ipw, it1 = it0 + ipw < items ? it0 + ipw : items;
    const char* xb = reinterpret_cast<const char*>(p.x);
    char* yb = reinterpret_cast<char*>(p.y);
    const size_t tstep = (size_t)16 * C * 2;                                 // bytes from a pixel tile to the next
    constexpr int NT = 2 * NH;                                               // 16-row A tiles: channels c0 + 32 (t >> 1) + 8 (row / 4) + 4 (t & 1) + row % 4
    for (int it = it0; it < it1; ++it) {
        const int seg = it % nseg, lc = it / nseg, line = lc / ncb, c0 = (lc - line * ncb) * (32 * NH), wt0 = seg * TU;
        const float2* s3 = p.S3 + (size_t)line * Ws * C + c0 + 8 * (r >> 2) + (r & 3);
        float2 graw[NT][KB][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int k0 = 8 * kb + 2 * g;
                const int ka = k0 < Ws ? k0 : Ws - 1, kc = k0 + 1 < Ws ? k0 + 1 : Ws - 1;   // past the band: any finite value (T is 0 there)
                graw[t][kb][0] = s3[(size_t)ka * C + 32 * (t >> 1) + 4 * (t & 1)];
                graw[t][kb][1] = s3[(size_t)kc * C + 32 * (t >> 1) + 4 * (t & 1)];
            }
        const size_t pix0 = (((size_t)line * W + wt0 * 16 + r) * C + c0 + 8 * g) * 2;      // this lane's pixel of the segment's first tile
        uint4 xr[TU][NH];
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) xr[u][hh] = *reinterpret_cast<const uint4*>(xb + pix0 + (size_t)u * tstep + 64 * hh);
        __builtin_amdgcn_sched_barrier(0);        // every load of the item is in flight before anything is consumed (the scheduler sinks them otherwise)
        bfx4 ah[NT][KB], al[NT][KB];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                unsigned h0, l0, h1, l1;
                split2(graw[t][kb][0].x, graw[t][kb][0].y, h0, l0);
                split2(graw[t][kb][1].x, graw[t][kb][1].y, h1, l1);
                ah[t][kb] = as_bfx4(h0, h1);
                al[t][kb] = as_bfx4(l0, l1);
            }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int wt = wt0 + u;
            bfx4 bh[KB], bl[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                bh[kb] = *reinterpret_cast<const bfx4*>(th + kb * 4 * pitch + wt * 128);
                bl[kb] = *reinterpret_cast<const bfx4*>(tl + kb * 4 * pitch + wt * 128);
            }
            fx4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = fx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[t][kb], bh[kb], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[t][kb], bl[kb], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[t][kb], bh[kb], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                const unsigned xin[4] = {xr[u][hh].x, xr[u][hh].y, xr[u][hh].z, xr[u][hh].w};
                unsigned o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                              // channels c0 + 32 hh + 8 g + 2 q, + 1
                    const float a = __uint_as_float(xin[q] << 16) + acc[2 * hh + (q >> 1)][(2 * q) & 3];
                    const float b = __uint_as_float(xin[q] & 0xffff0000u) + acc[2 * hh + (q >> 1)][((2 * q) & 3) + 1];
                    fx2 v; v.x = a; v.y = b;
                    o[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bfx2_t));
                }
                *reinterpret_cast<uint4*>(yb + pix0 + (size_t)u * tstep + 64 * hh) = make_uint4(o[0], o[1], o[2], o[3]);
            }
            __builtin_amdgcn_sched_barrier(0);    // tile by tile: multiply, add, store (not all stores at the end)
        }
    }
}

template <int KB, int TU, int NH>
static int launch_inv_mfma_tu(const FftP& p, hipStream_t st) {
    constexpr int WPG = kInvThreads / 64;
    const int items = p.B * p.H * (p.C / (32 * NH)) * ((p.W >> 4) / TU);
    int waves = fft_mfma_wgs() * WPG;
    if (waves > items) waves = items;
    const int ipw = (items + waves - 1) / waves;
    const int grid = ((items + ipw - 1) / ipw + WPG - 1) / WPG;
    const int lds = 2 * KB * 4 * trig_pitch(p.W) + p.W * (int)sizeof(float2);
    auto kern = &dft_rows_inv_mfma_kernel<KB, TU, NH>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kInvThreads), lds, st, p, items, ipw);
    MRFP_LAUNCH_CHECK();
    return 0;
}
template <int KB>
static int launch_inv_mfma(const FftP& p, hipStream_t st) {
    const int ntile = p.W >> 4;          // pixel tiles per line; an item takes TU of them, TU | ntile (no predicated tile: every
    if (p.C % 64 == 0) {                 // load and store of an item is straight-line code)
        if (ntile % 6 == 0) return launch_inv_mfma_tu<KB, 6, 2>(p, st);
        if (ntile % 4 == 0) return launch_inv_mfma_tu<KB, 4, 2>(p, st);
        if (ntile % 3 == 0) return launch_inv_mfma_tu<KB, 3, 2>(p, st);
        if (ntile % 2 == 0) return launch_inv_mfma_tu<KB, 2, 2>(p, st);
        return launch_inv_mfma_tu<KB, 1, 2>(p, st);
    }
    if (ntile % 6 == 0) return launch_inv_mfma_tu<KB, 6, 1>(p, st);
    if (ntile % 4 == 0) return launch_inv_mfma_tu<KB, 4, 1>(p, st);
    if (ntile % 3 == 0) return launch_inv_mfma_tu<KB, 3, 1>(p, st);
    if (ntile % 2 == 0) return launch_inv_mfma_tu<KB, 2, 1>(p, st);
    return launch_inv_mfma_tu<KB, 1, 1>(p, st);
}

// ---- band-limited forward row pass on the matrix cores (bf16 activations) --------------------------------------------
// S[kw][c] = sum_w x[w][c] exp(-2 pi i kw w / W) for the Ws stored bins: a [2 Ws x W] x [W x C] product per image line, x exact in
// bf16, the trigonometric matrix split into bf16 hi + lo parts (two products).  The contraction index w is the SLOW index of x
// ([w][c], channels contiguous), so a line's 32-channel slice is staged in LDS as it lies in memory and the B fragments are formed
// with the transposing read ds_read_b64_tr_b16 (as conv_wgrad.hip does); the A fragments come from a table in LDS that is built
// once per workgroup in fragment order (lane-linear 512-byte reads).  One wave per (line, 32-channel block) at a time, the waves
// of a workgroup on neighbouring blocks of the same line; the next item's x is already on its way (in registers) while the
// current one is multiplied out of LDS.
constexpr int kFwdThreads = 512;                          // 8 waves
typedef __attribute__((address_space(3))) bfx4 lds_bfx4;

// the 16-byte pieces of one LDS segment of an item (image line, 32-channel block): lane (g, r) takes pixel r of every tile, chunk g
template <int TU>
__device__ __forceinline__ void fwd_issue(uint4 (&xr)[TU], const char* xb, int it, int seg, int ncb, int W, int C, int r, int g) {
    const int line = it / ncb, c0 = (it - line * ncb) << 5;
    const size_t pix0 = (((size_t)line * W + seg * TU * 16 + r) * C + c0 + 8 * g) * 2;
    const size_t tstep = (size_t)16 * C * 2;
#pragma unroll
    for (int u = 0; u < TU; ++u) xr[u] = *reinterpret_cast<const uint4*>(xb + pix0 + (size_t)u * tstep);
}

template <int KB, int TU>      // TU: pixel tiles (of 16) per LDS segment of a line, TU | W / 16 (no predicated piece: the register array stays in registers)
__global__ __launch_bounds__(kFwdThreads) void dft_rows_fwd_mfma_kernel(FftP p, int items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = kFwdThreads / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int Ws = p.Ws, W = p.W, C = p.C, ncb = C >> 5, ntile = W >> 4;
    // A table: [part hi / lo][kb][ks] fragments of 64 lanes x 8 bytes; lane (r, q): k row 16 kb + r = (bin 8 kb + (r >> 1), re / im),
    // pixels 16 ks + 4 q + i
    float2* twl = reinterpret_cast<float2*>(smem + 2 * KB * ntile * 512 + NW * TU * 16 * 64);
    for (int e = threadIdx.x; e < W; e += kFwdThreads) twl[e] = p.tw[e];
    __syncthreads();
    for (int e = threadIdx.x; e < KB * ntile * 64; e += kFwdThreads) {
        const int l = e & 63, f = e >> 6, ks = f % ntile, kb = f / ntile;
        const int kw = 8 * kb + ((l & 15) >> 1), ri = l & 1, w0 = 16 * ks + 4 * (l >> 4);
        float v[4];
        int idx = (kw * w0) % W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 t = twl[idx];                                       // exp(-2 pi i kw w / W) = (cos, -sin)
            v[i] = kw < Ws ? (ri ? t.y : t.x) : 0.f;
            idx += kw; if (idx >= W) idx -= W;
        }
        unsigned h0, l0, h1, l1;
        split2(v[0], v[1], h0, l0);
        split2(v[2], v[3], h1, l1);
        *reinterpret_cast<uint2*>(smem + f * 512 + l * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(smem + (KB * ntile + f) * 512 + l * 8) = make_uint2(l0, l1);
    }
    __syncthreads();
    const char* ath = smem + lane * 8;                                      // + (kb * ntile + ks) * 512
    const char* atl = ath + KB * ntile * 512;
    char* xt = smem + 2 * KB * ntile * 512 + wave * (TU * 16 * 64);           // this wave's x segment: [pixel][64 bytes], chunk-swizzled
    const int nwaves = gridDim.x * NW, wid = blockIdx.x * NW + wave;
    const char* xb = reinterpret_cast<const char*>(p.x);
    const int nseg = ntile / TU;
    // the 16-byte chunk q of pixel row w sits at chunk q ^ (2 * ((w >> 2) & 1)): rows w and w + 4 (same banks otherwise) then use
    // disjoint halves of their 64-byte row, and a transposed read (16 rows x 32 bytes) is the two passes its 512 bytes need anyway
    const int wr_off = r * 64 + ((g ^ (((r >> 2) & 1) << 1)) << 4);           // store side: lane (g, r) holds pixel r of a tile, chunk g
    // read side, B fragment of channel tile nt at k step ks: group g reads rows 16 ks + 4 g + (j >> 2), 4 channels at 4 (j & 3)
    const int rd_row = 4 * g + (r >> 2);
    const int rd_off0 = rd_row * 64 + (((0 + ((r & 3) >> 1)) ^ (((rd_row >> 2) & 1) << 1)) << 4) + 8 * (r & 1);
    const int rd_off1 = rd_row * 64 + (((2 + ((r & 3) >> 1)) ^ (((rd_row >> 2) & 1) << 1)) << 4) + 8 * (r & 1);
    int it = wid;
    if (it >= items) return;                                                // (after the last barrier)
    while (it < items) {
        fx4 acc[KB][2];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { acc[kb][0] = fx4{0.f, 0.f, 0.f, 0.f}; acc[kb][1] = fx4{0.f, 0.f, 0.f, 0.f}; }
        for (int seg = 0; seg < nseg; ++seg) {
            uint4 xr[TU];
            // (a register set carried around the loop for the next item's pieces was demoted to scratch by the compiler; the eight waves
            //  of the workgroup hide each other's load latency instead)
            fwd_issue<TU>(xr, xb, it, seg, ncb, W, C, r, g);
#pragma unroll
            for (int u = 0; u < TU; ++u) *reinterpret_cast<uint4*>(xt + u * (16 * 64) + wr_off) = xr[u];
#pragma unroll 2
            for (int u = 0; u < TU; ++u) {
                const int ks = seg * TU + u;
                const bfx4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bfx4*)(xt + u * (16 * 64) + rd_off0));
                const bfx4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bfx4*)(xt + u * (16 * 64) + rd_off1));
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const bfx4 ah = *reinterpret_cast<const bfx4*>(ath + (kb * ntile + ks) * 512);
                    const bfx4 al = *reinterpret_cast<const bfx4*>(atl + (kb * ntile + ks) * 512);
                    acc[kb][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, b0, acc[kb][0], 0, 0, 0);
                    acc[kb][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, b1, acc[kb][1], 0, 0, 0);
                    acc[kb][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, b0, acc[kb][0], 0, 0, 0);
                    acc[kb][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, b1, acc[kb][1], 0, 0, 0);
                }
            }
        }
        // accumulator lane (g, n): k rows 16 kb + 4 g + i = bins 8 kb + 2 g (+1), (re, im); channel 16 nt + n
        const int line = it / ncb, c0 = (it - line * ncb) << 5;
        float2* dst = p.S + (size_t)line * Ws * C + c0 + r;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int kw = 8 * kb + 2 * g;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (kw < Ws) dst[(size_t)kw * C + 16 * nt] = make_float2(acc[kb][nt][0], acc[kb][nt][1]);
                if (kw + 1 < Ws) dst[(size_t)(kw + 1) * C + 16 * nt] = make_float2(acc[kb][nt][2], acc[kb][nt][3]);
            }
        }
        it += nwaves;
    }
}

// LDS of the forward kernel with TU pixel tiles per segment: the A table (both parts, every k step), one x segment per wave, the twiddles
static int fwd_lds_tu(int kb, int W, int tu) { return 2 * kb * (W >> 4) * 512 + (kFwdThreads / 64) * tu * 16 * 64 + W * (int)sizeof(float2); }
// the largest segment that divides the line and fits the 160 KB of a CU next to the table (0: none)
static int fwd_tu(int kb, int W) {
    const int ntile = W >> 4;
    for (int tu : {12, 8, 6, 4, 3, 2, 1})
        if (ntile % tu == 0 && fwd_lds_tu(kb, W, tu) <= 160 * 1024) return tu;
    return 0;
}
static int fwd_lds(int kb, int W) { return fwd_lds_tu(kb, W, fwd_tu(kb, W)); }

template <int KB, int TU>
static int launch_fwd_mfma_tu(const FftP& p, hipStream_t st) {
    constexpr int NW = kFwdThreads / 64;
    const int items = p.B * p.H * (p.C >> 5);
    int grid = (items + NW - 1) / NW;
    if (grid > 256) grid = 256;                                              // one workgroup per CU (LDS), persistent over its items
    const int lds = fwd_lds(KB, p.W);
    auto kern = &dft_rows_fwd_mfma_kernel<KB, TU>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kFwdThreads), lds, st, p, items);
    MRFP_LAUNCH_CHECK();
    return 0;
}
template <int KB>
static int launch_fwd_mfma(const FftP& p, hipStream_t st) {
    switch (fwd_tu(KB, p.W)) {
        case 12: return launch_fwd_mfma_tu<KB, 12>(p, st);
        case 8: return launch_fwd_mfma_tu<KB, 8>(p, st);
        case 6: return launch_fwd_mfma_tu<KB, 6>(p, st);
        case 4: return launch_fwd_mfma_tu<KB, 4>(p, st);
        case 3: return launch_fwd_mfma_tu<KB, 3>(p, st);
        case 2: return launch_fwd_mfma_tu<KB, 2>(p, st);
    }
    return launch_fwd_mfma_tu<KB, 1>(p, st);
}

template <int N1, int N2>
__global__ __launch_bounds__(two_nt(N1, N2)) void fft_cols_mix_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* lds = reinterpret_cast<float2*>(smem);
    const int ch = threadIdx.x % kCB, sub = threadIdx.x / kCB;
    int line, cg;
    if (!decode_wg(p.B * p.Ws, p.C / kCB, line, cg)) return;
    const int b = line / p.Ws, kw = line - b * p.Ws;
    const size_t C = p.C, hs = (size_t)p.Ws * C;
    const size_t off = ((size_t)b * p.H * p.Ws + kw) * C + cg * kCB + ch;
    float2 a[N1], F[N2];
    if (sub < N2) {
        const float2* own = p.S + off;
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) a[n1] = own[(size_t)(n1 * N2 + sub) * hs];
    }
    two_step<N1, N2>(a, F, lds, p.tw, ch, sub);            // F[k2] = spectrum at kh = sub + N1*k2 (threads sub < N1)
    // does the band touch this column at all?  low band: only kw <= radius; high band: everywhere
    const bool want_partner = !p.load_ratio && (p.high != 0 || (float)(kw * kw) <= p.radius2);
    float rr[N2];
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) rr[k2] = 1.f;
    if (want_partner) {                                    // workgroup-uniform
        float2 G[N2];
        __syncthreads();                                   // step B of the own column has left the exchange buffer
        if (sub < N2) {
            const int64_t pb = p.perm ? p.perm[b] : b;
            const float2* par = p.S + ((size_t)pb * p.H * p.Ws + kw) * C + cg * kCB + ch;
#pragma unroll
            for (int n1 = 0; n1 < N1; ++n1) a[n1] = par[(size_t)(n1 * N2 + sub) * hs];
        }
        two_step<N1, N2>(a, G, lds, p.tw, ch, sub);
        if (sub < N1) {
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                const int kh = sub + N1 * k2;
                const int dh = kh < p.H - kh ? kh : p.H - kh;
                const bool band = (float)(dh * dh + kw * kw) <= p.radius2;
                if (band != (p.high != 0)) {
                    const float am = sqrtf(F[k2].x * F[k2].x + F[k2].y * F[k2].y);
                    const float a2 = sqrtf(G[k2].x * G[k2].x + G[k2].y * G[k2].y);
                    if (am > 1e-20f) rr[k2] = ((1.f - p.lam) * am + p.lam * a2) / am;
                }
            }
        }
    }
    if (sub < N1) {
        if (p.delta && p.ratio && 2 * p.Ws - 1 <= p.H) {
            // band-limited path: the ratio differs from 1 only in the 2 M + 1 rows |kh| <= M = Ws - 1, and only those are kept
            // ([B, 2 M + 1, Ws, C]: rows 0..M, then H-M..H-1) -- 1/6 of the bytes at H = 192, M = 16, in the forward and the backward call
            const int M = p.Ws - 1;
            float* rat = p.ratio + ((size_t)b * (2 * M + 1) * p.Ws + kw) * C + cg * kCB + ch;
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                const int kh = sub + N1 * k2;
                const int cr = kh <= M ? kh : kh >= p.H - M ? kh - p.H + 2 * M + 1 : -1;
                if (cr < 0) { if (p.load_ratio) rr[k2] = 1.f; continue; }
                if (p.load_ratio) rr[k2] = rat[(size_t)cr * hs];
                else rat[(size_t)cr * hs] = rr[k2];
            }
        } else {
        float* rat = p.ratio ? p.ratio + off : nullptr;
        if (p.load_ratio) {
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) rr[k2] = rat[(size_t)(sub + N1 * k2) * hs];
        } else if (rat) {
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) rat[(size_t)(sub + N1 * k2) * hs] = rr[k2];
        }
        }
        const float one = p.delta ? 1.f : 0.f;                  // band-limited path: only the change F*(ratio-1) goes back
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            const float m = rr[k2] - one;
            F[k2] = make_float2(F[k2].x * m, -F[k2].y * m);     // conj for the inverse
        }
    }
    // inverse along H: the register layout (thread k1 owns k = N1*k2 + k1) is the INPUT layout of the transposed
    // two-step transform, so it starts without an exchange
    __syncthreads();
    float2 y[N1];
    two_step<N2, N1>(F, y, lds, p.tw, ch, sub);            // y[j] = value at h = sub + N2*j (threads sub < N2)
    if (sub < N2) {
        float2* dst = p.S3 + off;
#pragma unroll
        for (int j = 0; j < N1; ++j) dst[(size_t)(sub + N2 * j) * hs] = make_float2(y[j].x, -y[j].y);
    }
}

template <typename K>
static int launch_two(K kern, const FftP& p, int nlines, int nt, int lds, hipStream_t st) {
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const unsigned grid = (unsigned)(((nlines + 7) / 8) * 8 * (p.C / kCB));
    hipLaunchKernelGGL(kern, dim3(grid), dim3((unsigned)nt), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

// pass: 0 rows forward, 1 columns + mix, 2 rows inverse
template <typename T, int N1, int N2>
static int launch_fast_pass(FftP p, int pass, const float2* tw, hipStream_t st) {
    using TS = TwoStep<N1, N2>;
    p.N = TS::N;
    p.tw = tw;
    if constexpr (std::is_same<T, bf16>::value) {
        if (p.delta && pass == 0 && p.C % 32 == 0 && p.W % 16 == 0 && 2 * p.Ws <= 48 && fft_mfma() &&
            fwd_tu((2 * p.Ws + 15) / 16, p.W) > 0) {
            const int kb = (2 * p.Ws + 15) / 16;                // band-limited forward row pass on the matrix cores
            return kb == 1 ? launch_fwd_mfma<1>(p, st) : kb == 2 ? launch_fwd_mfma<2>(p, st) : launch_fwd_mfma<3>(p, st);
        }
        if (p.delta && pass == 2 && p.C % 32 == 0 && p.W % 16 == 0 && 2 * p.Ws <= 48 && fft_mfma() && !fft_nodirect()) {
            const int kb = (2 * p.Ws + 15) / 16;                // band-limited inverse row pass on the matrix cores
            return kb == 1 ? launch_inv_mfma<1>(p, st) : kb == 2 ? launch_inv_mfma<2>(p, st) : launch_inv_mfma<3>(p, st);
        }
    }
    if (p.delta && pass == 2 && p.C % kDirCh == 0 && p.Ws <= 64 && !fft_nopair() && !fft_nodirect()) {
        // band-limited inverse row pass as a direct trigonometric sum
        constexpr int PPT = 8 / (int)sizeof(T);            // 16-byte activation accesses
        p.N = dir_lines();
        const int ngrp = (p.B * p.H + p.N - 1) / p.N;
        // (a 128-channel tile -- 256-byte bf16 runs -- measured 7 % slower; 2..8 lines per workgroup the same as 1)
        const unsigned grid = (unsigned)(((ngrp + 7) / 8) * 8 * (p.C / kDirCh));
        const int lds = p.Ws * (kDirCh / 2) * (int)sizeof(float4) + (p.W / 4 + 1) * p.Ws * (int)sizeof(float2);
        auto kern = &dft_rows_inv_direct_kernel<T, PPT, kDirCh>;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kDirThreads), lds, st, p);
        MRFP_LAUNCH_CHECK();
        return 0;
    }
    if (p.delta && p.C % (2 * kCB) == 0 && pass != 1 && !fft_nopair()) {     // band-limited row passes on channel pairs
        const FftP& q = p;
        const unsigned grid = (unsigned)(((p.B * p.H + 7) / 8) * 8 * (p.C / (2 * kCB)));
        if (pass == 0) {
            if (TS::LDS > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rows_fwd_pair_kernel<T, N1, N2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, TS::LDS);
            hipLaunchKernelGGL((fft_rows_fwd_pair_kernel<T, N1, N2>), dim3(grid), dim3((unsigned)TS::NT), TS::LDS, st, q);
        } else {
            const int lds = TS::LDS + p.Ws * kCB * (int)sizeof(float4);
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rows_inv_pair_kernel<T, N1, N2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((fft_rows_inv_pair_kernel<T, N1, N2>), dim3(grid), dim3((unsigned)TS::NT), lds, st, q);
        }
        MRFP_LAUNCH_CHECK();
        return 0;
    }
    if (pass == 0) return launch_two(fft_rows_fwd_kernel<T, N1, N2>, p, p.B * p.H, TS::NT, TS::LDS, st);
    if (pass == 1) return launch_two(fft_cols_mix_kernel<N1, N2>, p, p.B * p.Ws, TS::NT, TS::LDS, st);
    return launch_two(fft_rows_inv_kernel<T, N1, N2>, p, p.B * p.H, TS::NT, TS::LDS + (TS::N / 2 + 1) * kCB * (int)sizeof(float2), st);
}

template <typename T>
static int fast_pass(const FftP& p, int n, int pass, const float2* tw, hipStream_t st) {
    switch (n) {
        case 32: return launch_fast_pass<T, 2, 16>(p, pass, tw, st);
        case 48: return launch_fast_pass<T, 3, 16>(p, pass, tw, st);
        case 64: return launch_fast_pass<T, 4, 16>(p, pass, tw, st);
        case 96: return launch_fast_pass<T, 6, 16>(p, pass, tw, st);
        case 128: return launch_fast_pass<T, 8, 16>(p, pass, tw, st);
        case 192: return launch_fast_pass<T, 12, 16>(p, pass, tw, st);
        case 256: return launch_fast_pass<T, 16, 16>(p, pass, tw, st);
        case 384: return launch_fast_pass<T, 12, 32>(p, pass, tw, st);
        case 512: return launch_fast_pass<T, 16, 32>(p, pass, tw, st);
    }
    set_error("fourier_mix: no two-step plan for length %d", n);
    return -1;
}
static bool has_fast_plan(int n) {
    return n == 32 || n == 48 || n == 64 || n == 96 || n == 128 || n == 192 || n == 256 || n == 384 || n == 512;
}

static int fft_generic() {
    static int generic = -1;
    if (generic < 0) { const char* e = getenv("MRFP_FFT_GENERIC"); generic = e ? atoi(e) : 0; }
    return generic;
}
static int fft_full() {          // MRFP_FFT_FULL=1: full half spectrum even for a low band (A/B against the band-limited path)
    static int full = -1;
    if (full < 0) { const char* e = getenv("MRFP_FFT_FULL"); full = e ? atoi(e) : 0; }
    return full;
}
// bins along W that S / S3 / ratio hold.  Low band on the register path: the ratio differs from 1 only for
// kw <= radius, so only those columns of the spectrum are ever formed (y = x + irfft2(F*(ratio-1))).
static int stored_bins(int64_t H, int64_t W, float radius, int high) {
    const int Wh = (int)(W / 2 + 1);
    if (high || fft_generic() || fft_full() || !has_fast_plan((int)H) || !has_fast_plan((int)W) || !(radius >= 0.f)) return Wh;
    const float fl = floorf(radius);
    return fl + 1.f < (float)Wh ? (int)fl + 1 : Wh;
}

template <typename T>
static int run_mix(FftP p, const float2* twH, const float2* twW, hipStream_t st) {
    if (!fft_generic() && has_fast_plan(p.H) && has_fast_plan(p.W)) {
        int rc;
        if ((rc = fast_pass<T>(p, p.W, 0, twW, st))) return rc;
        if ((rc = fast_pass<T>(p, p.H, 1, twH, st))) return rc;
        return fast_pass<T>(p, p.W, 2, twW, st);
    }
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

int64_t mrfp_fourier_stored_bins(int64_t H, int64_t W, float radius, int high) {
    if (H < 2 || W < 2) return 0;
    return stored_bins(H, W, radius, high);
}

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
    p.Ws = stored_bins(H, W, radius, high);
    p.delta = p.Ws < p.Wh;
    if (dtype == MRFP_F32) return run_mix<float>(p, (const float2*)twH, (const float2*)twW, (hipStream_t)stream);
    if (dtype == MRFP_BF16) return run_mix<bf16>(p, (const float2*)twH, (const float2*)twW, (hipStream_t)stream);
    if (dtype == MRFP_F16) return run_mix<f16>(p, (const float2*)twH, (const float2*)twW, (hipStream_t)stream);
    MRFP_CHECK(false, "fourier_mix: unknown dtype %d", dtype);
}

}  // extern "C"
