// fft_band.hip -- the LOW-band Fourier amplitude mix as a two-sided band-limited DFT (round 3).
//
// The low band  min(kh, H-kh)^2 + kw^2 <= r^2  touches only the (2M+1) x (M+1) corner coefficients of a plane's half spectrum
// (M = floor(r); 33 x 17 at the reference's r = 16), not H x (M+1) columns of it.  So instead of three FFT passes that carry
// half spectra [B,H,Ws,C] through HBM twice and read x twice (fft.hip: 4.75 plane-equivalents for 3 algorithmic ones), the
// path is:
//   A  band_rows_fwd   x -> R[b][h][k][c] = sum_w x[h][w] e^{-2 pi i k w / W},  k <= M           (read x once; R is 0.35 planes)
//   B  band_cols_fwd   R -> F[b][m][k][c] = sum_h R[h][k] e^{-2 pi i m h / H},  |m| <= M         (a few MB)
//   C  band_mix        D = F * (ratio - 1) inside the band, ratio from |F| and the partner's |F|   (a few MB)
//   D  band_cols_inv   G[b][h][k][c] = sum_m D[m][k] e^{+2 pi i m h / H}                          (G is 0.35 planes, like R)
//   E  band_synth      y = x + the M+1-term trigonometric row sum of G_h: x read once, y written once
// = 3 planes + 1.4 of R / G instead of 4.75 -- and R and G (53 MB each at 16 x 128 x 192^2) are written and re-read within
// microseconds, i.e. through the 256 MB Infinity Cache.  (Rebuilding G_h inside the synthesis kernel from the coefficient array
// was built first: every 4-row workgroup re-reads its image's 0.6 MB coefficient slice, 455 MB of L2 traffic per call, and
// the two latency-bound phases share one workgroup: 126 us against 60 us of HBM time; profiles/r03_fourier.md.)
// All kernels are DIRECT sums (no FFT): with 17 of 97 bins wanted, a length-192 line
// costs 17 x 2 x 192 / 4 multiply-adds after folding the four positions j, W/2-j, W/2+j, W-j that share |cos| and |sin| -- 8.5
// per element -- which a streaming kernel hides under its own HBM time.  Lane = TWO adjacent channels (4-byte bf16 / 8-byte fp32
// accesses; scalar v_fma_f32 with the wave-uniform twiddle as the SGPR operand), wave =
// 128 channels of one image row (256-byte runs).
//
// Build-defined operator (no reference function: parity unpinned), oracle torch.fft
// (oracle/mrfp_oracle.py::fourier_amplitude_mix); nearest reference arithmetic dataloaders.py:24-45, 59-79.
#include "common.hpp"

namespace mrfp {

// Two channels of one lane, as two SCALAR floats on purpose: packed fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32) issues at
// roughly a quarter of the rate of the two v_fma_f32 it replaces on this part (MI355X_MICROARCH.md, cycle constants: "1
// v_pk_fma_f32 +22 cyc vs 2 v_fma_f32"; measured here: the first version of these kernels, written with 2-vectors, spent 29
// cycles per v_pk_fma_f32 and was bound by exactly that -- profiles/r03_fourier.md).  The library is built with
// -fno-slp-vectorize so that the compiler does not re-pack them.
struct bf32x2 { float x, y; };
__device__ __forceinline__ bf32x2 operator+(bf32x2 a, bf32x2 b) { return bf32x2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ bf32x2 operator-(bf32x2 a, bf32x2 b) { return bf32x2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ bf32x2 operator*(bf32x2 a, float b) { return bf32x2{a.x * b, a.y * b}; }
__device__ __forceinline__ bf32x2 fma2b(bf32x2 a, bf32x2 b, bf32x2 c) {      // two v_fma_f32 (the library is built with -ffp-contract=off)
    return bf32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
}
__device__ __forceinline__ bf32x2 splat(float v) { return bf32x2{v, v}; }
typedef float __attribute__((ext_vector_type(2))) cvt_f32x2;      // (only for the two-element conversion instructions)

// two adjacent channels of one pixel as fp32
template <typename T> struct Pair;
template <> struct Pair<float> {
    static __device__ __forceinline__ bf32x2 load(const float* p) { const float2 v = *reinterpret_cast<const float2*>(p); return bf32x2{v.x, v.y}; }
    static __device__ __forceinline__ void store(float* p, bf32x2 v) { *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y); }
};
template <> struct Pair<bf16> {
    static __device__ __forceinline__ bf32x2 load(const bf16* p) {
        const unsigned w = *reinterpret_cast<const unsigned*>(p);
        return bf32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    }
    static __device__ __forceinline__ void store(bf16* p, bf32x2 v) {
        typedef __bf16 __attribute__((ext_vector_type(2))) bf16x2v;
        const cvt_f32x2 f = {v.x, v.y};
        *reinterpret_cast<unsigned*>(p) = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2v));
    }
};
template <> struct Pair<f16> {
    typedef _Float16 __attribute__((ext_vector_type(2))) f16x2v;
    static __device__ __forceinline__ bf32x2 load(const f16* p) {
        const f16x2v h = *reinterpret_cast<const f16x2v*>(p);
        const cvt_f32x2 f = __builtin_convertvector(h, cvt_f32x2);
        return bf32x2{f.x, f.y};
    }
    static __device__ __forceinline__ void store(f16* p, bf32x2 v) {
        const cvt_f32x2 f = {v.x, v.y};
        *reinterpret_cast<f16x2v*>(p) = __builtin_convertvector(f, f16x2v);
    }
};

#ifndef MRFP_BAND_JB
#define MRFP_BAND_JB 4
#endif
// Four channels of one lane for the two STREAMING kernels (rows, synthesis): 8-byte (16-bit types) / 16-byte (fp32) accesses.
// With two channels per lane (4-byte accesses) both kernels were bound by the ISSUE of vector-memory instructions -- one
// 256-byte wave-instruction per ~17 cycles per CU, SQ_WAIT_INST_ANY 64 % of the wave cycles, 87 us for a pass with 60 us of HBM
// time (profiles/r03_fourier.md; MI355X_MICROARCH.md: narrow stores are issue-bound, not bandwidth-bound).
constexpr int kCPL = 4;
struct fv4 { float v[kCPL]; };
__device__ __forceinline__ fv4 operator+(const fv4& a, const fv4& b) { fv4 r; for (int i = 0; i < kCPL; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
__device__ __forceinline__ fv4 operator-(const fv4& a, const fv4& b) { fv4 r; for (int i = 0; i < kCPL; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
__device__ __forceinline__ fv4 fmas(float s, const fv4& x, const fv4& acc) {      // acc + s * x, four v_fma_f32 with the scalar in an SGPR
    fv4 r;
#pragma unroll
    for (int i = 0; i < kCPL; ++i) r.v[i] = __builtin_fmaf(s, x.v[i], acc.v[i]);
    return r;
}
__device__ __forceinline__ fv4 zero4() { fv4 r; for (int i = 0; i < kCPL; ++i) r.v[i] = 0.f; return r; }
template <typename T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ fv4 load(const float* p) { const float4 v = *reinterpret_cast<const float4*>(p); return fv4{{v.x, v.y, v.z, v.w}}; }
    static __device__ __forceinline__ void store(float* p, const fv4& v) { *reinterpret_cast<float4*>(p) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]); }
};
template <> struct Quad<bf16> {
    static __device__ __forceinline__ fv4 load(const bf16* p) {
        const uint2 w = *reinterpret_cast<const uint2*>(p);
        return fv4{{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)}};
    }
    static __device__ __forceinline__ void store(bf16* p, const fv4& v) {
        typedef __bf16 __attribute__((ext_vector_type(2))) bf16x2v;
        const cvt_f32x2 a = {v.v[0], v.v[1]}, b = {v.v[2], v.v[3]};
        *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2v)),
                                                  __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2v)));
    }
};
template <> struct Quad<f16> {
    typedef _Float16 __attribute__((ext_vector_type(2))) f16x2v;
    static __device__ __forceinline__ fv4 load(const f16* p) {
        const uint2 w = *reinterpret_cast<const uint2*>(p);
        const cvt_f32x2 a = __builtin_convertvector(__builtin_bit_cast(f16x2v, w.x), cvt_f32x2), b = __builtin_convertvector(__builtin_bit_cast(f16x2v, w.y), cvt_f32x2);
        return fv4{{a.x, a.y, b.x, b.y}};
    }
    static __device__ __forceinline__ void store(f16* p, const fv4& v) {
        const cvt_f32x2 a = {v.v[0], v.v[1]}, b = {v.v[2], v.v[3]};
        *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, f16x2v)),
                                                  __builtin_bit_cast(unsigned, __builtin_convertvector(b, f16x2v)));
    }
};
// Lane geometry of the streaming kernels: lpr = C / 4 lanes per image row.  lpr >= 64: a wave is one 256-channel chunk of one
// row; lpr < 64: a wave holds 64 / lpr consecutive rows (C = 128: two) -- the twiddles depend on the position in the row only,
// so they stay wave-uniform either way.
struct StreamGeom { int bh; int c0; bool on; };
__device__ __forceinline__ StreamGeom stream_geom(const int unit, const int lane, const int C, const int rows) {
    const int lpr = C / kCPL;
    StreamGeom g;
    if (lpr >= 64) {
        const int nch = (lpr + 63) / 64;
        g.bh = unit / nch;
        g.c0 = ((unit % nch) * 64 + lane) * kCPL;
        g.on = g.c0 < C && g.bh < rows;
    } else {
        const int rpw = 64 / lpr, r = lane / lpr;
        g.bh = unit * rpw + r;
        g.c0 = (lane - r * lpr) * kCPL;
        g.on = r < rpw && g.bh < rows;
    }
    return g;
}
__host__ __device__ inline int stream_units(int rows, int C) {
    const int lpr = C / kCPL;
    return lpr >= 64 ? rows * ((lpr + 63) / 64) : (rows + 64 / lpr - 1) / (64 / lpr);
}

constexpr int kTabStride = 17;      // bins per table row (M <= 16)

// One wave-uniform table row (NB twiddles = 2 NB dwords of scalar loads).  Every kernel below fetches the row of step i + 1
// BEFORE it multiplies with the row of step i (two sets of scalar registers): a scalar load that the multiplies of its own
// step wait for costs 0.3-1 us per step (tabH, 26 KB at H = 192, does not even fit the 16 KB scalar cache) -- a column kernel
// with 13 steps per wave took 28 us for 4 us of arithmetic.
template <int NB>
__device__ __forceinline__ void load_tw(const float2* __restrict__ tab, int row, float2 (&t)[NB]) {
#pragma unroll
    for (int k = 0; k < NB; ++k) t[k] = tab[row * kTabStride + k];
}

struct BandP {
    const void* x;
    void* y;
    float2* R;          // [B][NBu][H][C]   rows transformed (scratch S)
    float2* F;          // [B][2M+1][NBu][C] corner coefficients (scratch S3); m >= 0 at index m, m < 0 at index M - m
    float4* D;          // [B][M+1][NBu][C]  (P.re, P.im, Q.re, Q.im): P = D[+m] + D[-m], Q = D[+m] - D[-m] (m = 0: P = D[0], Q = 0); in S3 behind F
    float2* G;          // [B][NBu][H][C]   columns transformed back (scratch S: R is dead by then)
    float* rm1;         // [B][2M+1][NBu][C] ratio - 1 (0 outside the band): written by the forward call, read by the backward call
    const int64_t* perm;
    const float2* tabH; // [H][17]      (cos, sin)(2 pi m h / H)          host-built (mrfp_fourier_band_tables)
    const float2* tabW; // [W/4 + 1][17] (cos, sin)(2 pi k j / W): a wave-uniform row of twiddles per trip = a few scalar loads
    int B, H, W, C, M, NBu, nchunk;
    float radius2, lam, scale;
    int load_ratio;
};

// ---- A: rows.  One wave = 256 channels of one image row (or 64 / (C/4) whole rows); lane = 4 channels; NB bins in registers.
template <typename T, int NB>
__global__ __launch_bounds__(256) void band_rows_fwd_kernel(BandP p) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave;
    const int rows = p.B * p.H;
    if (unit >= stream_units(rows, p.C)) return;
    const StreamGeom g = stream_geom(unit, lane, p.C, rows);
    const int W = p.W, C = p.C, Q = W >> 2, Wh = W >> 1;
    const T* xr = reinterpret_cast<const T*>(p.x) + (g.on ? (size_t)g.bh * W * C + g.c0 : 0);
    fv4 re[NB], im[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) { re[k] = zero4(); im[k] = zero4(); }
    constexpr int JB = 2;                                       // quads per trip: 8 independent 8/16-byte loads in flight per lane
    float2 tn[NB];
    load_tw<NB>(p.tabW, 0, tn);
    for (int j0 = 0; j0 <= Q; j0 += JB) {
        fv4 a[JB], b[JB], c[JB], d[JB];
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            const int j = min(j0 + u, Q);                       // (tail trips re-load quad Q; masked below)
            a[u] = Quad<T>::load(xr + (size_t)j * C);
            b[u] = Quad<T>::load(xr + (size_t)(Wh - j) * C);
            c[u] = Quad<T>::load(xr + (size_t)(Wh + j) * C);
            d[u] = Quad<T>::load(xr + (size_t)(j == 0 ? 0 : W - j) * C);
        }
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            const int j = j0 + u;
            if (j > Q) break;                                   // (uniform)
            float2 t[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) t[k] = tn[k];
            load_tw<NB>(p.tabW, min(j + 1, Q), tn);             // next quad's twiddles (cos, sin)(2 pi k j / W)
            // j = 0: positions 0 and W/2 only;  j = W/4: positions W/4 and 3W/4 only
            const fv4 z = zero4();
            const fv4 bb = (j == 0 || j == Q) ? z : b[u];
            const fv4 cc = (j == Q) ? z : c[u];
            const fv4 dd = (j == 0) ? z : d[u];
            const fv4 s1 = a[u] + dd, s2 = bb + cc, d1 = a[u] - dd, d2 = cc - bb;
            const fv4 pe = s1 + s2, po = s1 - s2, qe = d1 + d2, qo = d1 - d2;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                re[k] = fmas(t[k].x, (k & 1) ? po : pe, re[k]);
                im[k] = fmas(-t[k].y, (k & 1) ? qo : qe, im[k]);
            }
        }
    }
    if (!g.on) return;
    // R[b][k][h][c]: the column kernel walks h for a fixed (b, k) -- contiguous rows
    const int bb_ = g.bh / p.H, hh_ = g.bh - bb_ * p.H;
    float2* out = p.R + (((size_t)bb_ * p.NBu) * p.H + hh_) * C + g.c0;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (k < p.NBu) {
            float4* o = reinterpret_cast<float4*>(out + (size_t)k * p.H * C);
            o[0] = make_float4(re[k].v[0], im[k].v[0], re[k].v[1], im[k].v[1]);
            o[1] = make_float4(re[k].v[2], im[k].v[2], re[k].v[3], im[k].v[3]);
        }
}

// ---- B: columns.  One workgroup = (b, bin k, chunk); its 8 waves split the row pairs (h, H - h); lane = 2 channels. ----------
//   F[+-m] = R_0 + (-1)^m R_{H/2} + A_m -+ i B_m,   A_m = sum_h cos(phi) (R_h + R_{H-h}),  B_m = sum_h sin(phi) (R_h - R_{H-h})
// A wave loads ALL the rows of a trip (up to kColsPairs pairs = 2 kColsPairs 16-byte loads per lane) before it multiplies: with
// one (b, k, chunk) column per workgroup there are only ~300 workgroups, so the latency must be paid once, not per row pair
// (4 waves x 24 serial pairs: 23.7 us; this form: see profiles/r03_fourier.md).
constexpr int kColsWaves = 8, kColsPairs = 12;
template <int NB>
__global__ __launch_bounds__(64 * kColsWaves) __attribute__((amdgpu_waves_per_eu(1, 2))) void band_cols_fwd_kernel(BandP p) {
    __shared__ float4 red[kColsWaves - 1][64][2];               // partial (A.re, A.im, B.re, B.im) x 2 channels of waves 1.., per m
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int wg = blockIdx.x;
    const int chunk = wg % p.nchunk, k = (wg / p.nchunk) % p.NBu, b = wg / (p.nchunk * p.NBu);
    const int c0 = chunk * 128 + lane * 2;
    const bool on = c0 < p.C;
    const int H = p.H, C = p.C, Hh = H >> 1;
    const size_t hstride = (size_t)C;                           // float2 elements between rows of R[b][k][h][c]
    const float2* Rb = p.R + (((size_t)b * p.NBu + k) * H) * C + (on ? c0 : 0);
    bf32x2 Ar[NB], Ai[NB], Br[NB], Bi[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) { Ar[m] = splat(0.f); Ai[m] = splat(0.f); Br[m] = splat(0.f); Bi[m] = splat(0.f); }
    float2 tn[NB];
    load_tw<NB>(p.tabH, min(1 + wave, Hh - 1), tn);
    for (int hb = 1 + wave; hb < Hh; hb += kColsWaves * kColsPairs) {          // trip: pairs hb, hb + 8, hb + 16, ...
        float4 r1[kColsPairs], r2[kColsPairs];
#pragma unroll
        for (int u = 0; u < kColsPairs; ++u) {
            const int h = min(hb + u * kColsWaves, Hh - 1);      // (clamped: the tail pairs are masked below)
            r1[u] = *reinterpret_cast<const float4*>(Rb + (size_t)h * hstride);
            r2[u] = *reinterpret_cast<const float4*>(Rb + (size_t)(H - h) * hstride);
        }
#pragma unroll
        for (int u = 0; u < kColsPairs; ++u) {
            const int h = hb + u * kColsWaves;
            if (h >= Hh) break;                                 // (uniform)
            const bf32x2 ur = bf32x2{r1[u].x + r2[u].x, r1[u].z + r2[u].z}, ui = bf32x2{r1[u].y + r2[u].y, r1[u].w + r2[u].w};
            const bf32x2 vr = bf32x2{r1[u].x - r2[u].x, r1[u].z - r2[u].z}, vi = bf32x2{r1[u].y - r2[u].y, r1[u].w - r2[u].w};
            float2 t[NB];
#pragma unroll
            for (int m = 0; m < NB; ++m) t[m] = tn[m];
            load_tw<NB>(p.tabH, min(h + kColsWaves, Hh - 1), tn);           // the next pair's (cos, sin)(2 pi m h / H)
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                Ar[m] = fma2b(splat(t[m].x), ur, Ar[m]);
                Ai[m] = fma2b(splat(t[m].x), ui, Ai[m]);
                Br[m] = fma2b(splat(t[m].y), vr, Br[m]);
                Bi[m] = fma2b(splat(t[m].y), vi, Bi[m]);
            }
        }
    }
    // combine the waves in a fixed order (wave 0 + wave 1 + ... + wave 7), one m at a time through LDS
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), rh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave == 0) {
        r0 = *reinterpret_cast<const float4*>(Rb);
        rh = *reinterpret_cast<const float4*>(Rb + (size_t)Hh * hstride);
    }
    float2* Fb = p.F + ((size_t)b * (2 * p.M + 1) * p.NBu + k) * C + c0;
    const size_t mstride = (size_t)p.NBu * C;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        __syncthreads();
        if (wave > 0) {
            red[wave - 1][lane][0] = make_float4(Ar[m].x, Ai[m].x, Br[m].x, Bi[m].x);
            red[wave - 1][lane][1] = make_float4(Ar[m].y, Ai[m].y, Br[m].y, Bi[m].y);
        }
        __syncthreads();
        if (wave == 0 && on && m <= p.M) {
            float4 s0 = make_float4(Ar[m].x, Ai[m].x, Br[m].x, Bi[m].x), s1 = make_float4(Ar[m].y, Ai[m].y, Br[m].y, Bi[m].y);
#pragma unroll
            for (int w = 0; w < kColsWaves - 1; ++w) {
                const float4 q0 = red[w][lane][0], q1 = red[w][lane][1];
                s0.x += q0.x; s0.y += q0.y; s0.z += q0.z; s0.w += q0.w;
                s1.x += q1.x; s1.y += q1.y; s1.z += q1.z; s1.w += q1.w;
            }
            const float sg = (m & 1) ? -1.f : 1.f;
            // channel 0: (r0.x, r0.y), channel 1: (r0.z, r0.w)
            const float e0r = r0.x + sg * rh.x + s0.x, e0i = r0.y + sg * rh.y + s0.y;
            const float e1r = r0.z + sg * rh.z + s1.x, e1i = r0.w + sg * rh.w + s1.y;
            // -i B = (B.im, -B.re)
            *reinterpret_cast<float4*>(Fb + (size_t)m * mstride) = make_float4(e0r + s0.w, e0i - s0.z, e1r + s1.w, e1i - s1.z);
            if (m > 0)
                *reinterpret_cast<float4*>(Fb + (size_t)(p.M + m) * mstride) = make_float4(e0r - s0.w, e0i + s0.z, e1r - s1.w, e1i + s1.z);
        }
    }
}

// ---- C: amplitude mix on the corner coefficients.  One thread = (b, m >= 0, k, channel): both signs of m. ----------------------
__global__ __launch_bounds__(256) void band_mix_kernel(BandP p) {
    const int64_t n = (int64_t)p.B * (p.M + 1) * p.NBu * p.C;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % p.C);
    int64_t rest = i / p.C;
    const int k = (int)(rest % p.NBu); rest /= p.NBu;
    const int m = (int)(rest % (p.M + 1));
    const int b = (int)(rest / (p.M + 1));
    const int NK = 2 * p.M + 1;
    const bool inband = (float)(m * m + k * k) <= p.radius2;
    const int pb = (!p.load_ratio && p.perm) ? (int)p.perm[b] : b;
    float2 dl[2];
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
        const int mi = sgn == 0 ? m : p.M + m;
        if (sgn == 1 && m == 0) { dl[1] = make_float2(0.f, 0.f); break; }
        const size_t o = (((size_t)b * NK + mi) * p.NBu + k) * p.C + c;
        const float2 f = p.F[o];
        float r;
        if (p.load_ratio) {
            r = p.rm1[o];
        } else {
            r = 0.f;
            if (inband) {
                const float2 g = p.F[(((size_t)pb * NK + mi) * p.NBu + k) * p.C + c];
                const float A = sqrtf(f.x * f.x + f.y * f.y), Ap = sqrtf(g.x * g.x + g.y * g.y);
                if (A > 1e-20f) r = ((1.f - p.lam) * A + p.lam * Ap) / fmaxf(A, 1e-30f) - 1.f;
            }
            if (p.rm1) p.rm1[o] = r;
        }
        dl[sgn] = make_float2(f.x * r, f.y * r);
    }
    // irfft2 scaling 1/(H W), and the factor 2 of the bins 0 < k < W/2 (k = 0 counts once), folded in here
    const float sc = p.scale * (k == 0 ? 1.f : 2.f);
    float4 out;
    if (m == 0) out = make_float4(dl[0].x * sc, dl[0].y * sc, 0.f, 0.f);
    else out = make_float4((dl[0].x + dl[1].x) * sc, (dl[0].y + dl[1].y) * sc, (dl[0].x - dl[1].x) * sc, (dl[0].y - dl[1].y) * sc);
    p.D[(((size_t)b * (p.M + 1) + m) * p.NBu + k) * p.C + c] = out;
}

// ---- D: columns back.  One single-wave workgroup = (b, bin k, chunk, 1/8 of the row pairs); the coefficients of all M+1
//   steps sit in registers (loaded once): G_h = D0 + sum_m (cos P_m + i sin Q_m), G_{H-h} = D0 + sum_m (cos P_m - i sin Q_m).
//   (Single waves: with one 4-wave workgroup per column the grid was 272 workgroups on 256 CUs -- the 16 CUs that got two of
//   them set the kernel time, 23.6 us for 53 MB.)
constexpr int kInvSlices = 8;
template <int NB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void band_cols_inv_kernel(BandP p) {
    const int wave = blockIdx.x % kInvSlices, lane = threadIdx.x & 63;
    const int wg = blockIdx.x / kInvSlices;
    const int chunk = wg % p.nchunk, k = (wg / p.nchunk) % p.NBu, b = wg / (p.nchunk * p.NBu);
    const int c0 = chunk * 128 + lane * 2;
    if (c0 >= p.C) return;
    const int H = p.H, C = p.C, Hh = H >> 1;
    const float4* Db = p.D + ((size_t)b * (p.M + 1) * p.NBu + k) * C + c0;
    const size_t mstride = (size_t)p.NBu * C;
    bf32x2 Pr[NB], Pi[NB], Qr[NB], Qi[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const int mc = min(m, p.M);
        const float4 d0 = Db[(size_t)mc * mstride], d1 = Db[(size_t)mc * mstride + 1];
        const float use = m <= p.M ? 1.f : 0.f;
        Pr[m] = bf32x2{d0.x, d1.x} * use; Pi[m] = bf32x2{d0.y, d1.y} * use; Qr[m] = bf32x2{d0.z, d1.z} * use; Qi[m] = bf32x2{d0.w, d1.w} * use;
    }
    float2* Gb = p.G + (((size_t)b * p.NBu + k) * H) * C + c0;      // G[b][k][h][c]
    const size_t hstride = (size_t)C;
    float2 tn[NB];
    load_tw<NB>(p.tabH, min(wave, Hh), tn);
    for (int h = wave; h <= Hh; h += kInvSlices) {
        bf32x2 ur = splat(0.f), ui = splat(0.f), vr = splat(0.f), vi = splat(0.f);
        float2 t[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) t[m] = tn[m];
        load_tw<NB>(p.tabH, min(h + kInvSlices, Hh), tn);                   // the next row's (cos, sin)(phi)
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            ur = fma2b(splat(t[m].x), Pr[m], ur);
            ui = fma2b(splat(t[m].x), Pi[m], ui);
            vr = fma2b(splat(-t[m].y), Qi[m], vr);                          // i sin (Qr + i Qi) = -sin Qi + i sin Qr
            vi = fma2b(splat(t[m].y), Qr[m], vi);
        }
        const bf32x2 gr = ur + vr, gi = ui + vi;
        *reinterpret_cast<float4*>(Gb + (size_t)h * hstride) = make_float4(gr.x, gi.x, gr.y, gi.y);
        if (h != 0 && h != Hh) {
            const bf32x2 hr = ur - vr, hi = ui - vi;
            *reinterpret_cast<float4*>(Gb + (size_t)(H - h) * hstride) = make_float4(hr.x, hi.x, hr.y, hi.y);
        }
    }
}

// ---- E: synthesis.  Lane geometry of the rows kernel: y[w] = x[w] + sum_k (G.re cos(theta) - G.im sin(theta)); the four
//   positions j, W/2 - j, W/2 + j, W - j of a trip share the products.  Pure streaming: no LDS.
template <typename T, int NB>
__global__ __launch_bounds__(256) void band_synth_kernel(BandP p) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave;
    const int rows = p.B * p.H;
    if (unit >= stream_units(rows, p.C)) return;
    const StreamGeom g = stream_geom(unit, lane, p.C, rows);
    const int W = p.W, C = p.C;
    fv4 gr[NB], gi[NB];
    {
        const int bh = g.on ? g.bh : 0;
        const int bb_ = bh / p.H, hh_ = bh - bb_ * p.H;
        const float2* Gr = p.G + (((size_t)bb_ * p.NBu) * p.H + hh_) * C + (g.on ? g.c0 : 0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const float4* src = reinterpret_cast<const float4*>(Gr + (size_t)min(k, p.NBu - 1) * p.H * C);
            const float4 v0 = src[0], v1 = src[1];
            const float use = k < p.NBu ? 1.f : 0.f;
            gr[k] = fv4{{v0.x * use, v0.z * use, v1.x * use, v1.z * use}};
            gi[k] = fv4{{v0.y * use, v0.w * use, v1.y * use, v1.w * use}};
        }
    }
    const size_t rowoff = g.on ? ((size_t)g.bh * W) * C + g.c0 : 0;
    const T* xr = reinterpret_cast<const T*>(p.x) + rowoff;
    T* yr = reinterpret_cast<T*>(p.y) + rowoff;
    const int Q = W >> 2, Wh = W >> 1;
    constexpr int JB = 2;
    float2 tn[NB];
    load_tw<NB>(p.tabW, 0, tn);
    for (int j0 = 0; j0 <= Q; j0 += JB) {
        fv4 a[JB], bq[JB], c[JB], d[JB];
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            const int j = min(j0 + u, Q);
            a[u] = Quad<T>::load(xr + (size_t)j * C);
            bq[u] = Quad<T>::load(xr + (size_t)(Wh - j) * C);
            c[u] = Quad<T>::load(xr + (size_t)(Wh + j) * C);
            d[u] = Quad<T>::load(xr + (size_t)(j == 0 ? 0 : W - j) * C);
        }
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            const int j = j0 + u;
            if (j > Q) break;
            fv4 Ee = zero4(), Eo = zero4(), Se = zero4(), So = zero4();
            float2 t[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) t[k] = tn[k];
            load_tw<NB>(p.tabW, min(j + 1, Q), tn);                          // next quad's (cos, sin)(theta)
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (k & 1) { Eo = fmas(t[k].x, gr[k], Eo); So = fmas(t[k].y, gi[k], So); }
                else { Ee = fmas(t[k].x, gr[k], Ee); Se = fmas(t[k].y, gi[k], Se); }
            }
            const fv4 E1 = Ee + Eo, E2 = Ee - Eo, S1 = Se + So, S2 = Se - So;
            if (!g.on) continue;
            // y_j = E1 - S1;  y_{W-j} = E1 + S1;  y_{W/2-j} = E2 + S2;  y_{W/2+j} = E2 - S2
            Quad<T>::store(yr + (size_t)j * C, a[u] + (E1 - S1));
            if (j != 0 && j != Q) Quad<T>::store(yr + (size_t)(Wh - j) * C, bq[u] + (E2 + S2));
            if (j != Q) Quad<T>::store(yr + (size_t)(Wh + j) * C, c[u] + (E2 - S2));
            if (j != 0) Quad<T>::store(yr + (size_t)(W - j) * C, d[u] + (E1 + S1));
        }
    }
}

template <typename T, int NB>
static int band_run_nb(const BandP& p, hipStream_t st) {
    const int units = stream_units(p.B * p.H, p.C);
    hipLaunchKernelGGL((band_rows_fwd_kernel<T, NB>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, p);
    MRFP_LAUNCH_CHECK();
    hipLaunchKernelGGL((band_cols_fwd_kernel<NB>), dim3((unsigned)(p.B * p.NBu * p.nchunk)), dim3(64 * kColsWaves), 0, st, p);
    MRFP_LAUNCH_CHECK();
    const int64_t n = (int64_t)p.B * (p.M + 1) * p.NBu * p.C;
    hipLaunchKernelGGL(band_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    MRFP_LAUNCH_CHECK();
    hipLaunchKernelGGL((band_cols_inv_kernel<NB>), dim3((unsigned)(p.B * p.NBu * p.nchunk * kInvSlices)), dim3(64), 0, st, p);
    MRFP_LAUNCH_CHECK();
    hipLaunchKernelGGL((band_synth_kernel<T, NB>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

// MRFP_FFT_BAND=0: keep the three-pass FFT path of fft.hip for the low band (A/B measurements)
static int band_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MRFP_FFT_BAND"); on = e ? atoi(e) : 1; }
    return on;
}

bool band_applicable(int64_t H, int64_t W, int64_t C, float radius, int high) {
    if (high || !band_enabled() || !(radius >= 0.f)) return false;
    const int M = (int)floorf(radius);
    return M <= 16 && (W % 4) == 0 && (H % 2) == 0 && 4 * M + 3 <= H && M + 1 < W / 2 && (C % 4) == 0 &&
           (C / 4 >= 64 || 64 % (C / 4) == 0 || C / 4 < 64);
}

int band_mix_run(const void* x, void* y, const int64_t* perm, void* S, void* S3, float* ratio, int load_ratio, const void* tabs,
                 int dtype, int64_t B, int64_t H, int64_t W, int64_t C, float radius, float lam, hipStream_t st) {
    BandP p;
    p.x = x; p.y = y; p.R = (float2*)S; p.G = (float2*)S; p.F = (float2*)S3; p.rm1 = ratio; p.perm = perm;
    // S3 holds [B][H][M+1][C] float2: F takes [B][2M+1][M+1][C] float2, D [B][M+1][M+1][C] float4 = 2(M+1) more rows of the
    // same size -- band_applicable() checked 2M+1 + 2(M+1) <= H
    p.D = (float4*)((float2*)S3 + (size_t)B * (2 * (int)floorf(radius) + 1) * ((int)floorf(radius) + 1) * C);
    p.tabH = (const float2*)tabs; p.tabW = p.tabH + (size_t)H * kTabStride;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.M = (int)floorf(radius); p.NBu = p.M + 1;
    p.nchunk = (int)((C + 127) / 128);
    p.radius2 = radius * radius; p.lam = lam; p.scale = 1.0f / (float)(H * W); p.load_ratio = load_ratio;
#define MRFP_BAND_GO(T)                                                                   \
    return p.NBu <= 5 ? band_run_nb<T, 5>(p, st) : p.NBu <= 9 ? band_run_nb<T, 9>(p, st) : band_run_nb<T, 17>(p, st)
    if (dtype == MRFP_F32) { MRFP_BAND_GO(float); }
    if (dtype == MRFP_BF16) { MRFP_BAND_GO(bf16); }
    if (dtype == MRFP_F16) { MRFP_BAND_GO(f16); }
#undef MRFP_BAND_GO
    set_error("fourier_mix: unknown dtype %d", dtype);
    return -1;
}

}  // namespace mrfp
