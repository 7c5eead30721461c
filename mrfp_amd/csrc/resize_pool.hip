// resize_pool.hip -- bilinear (align_corners=True) resize fwd/bwd and MaxPool2d(3,2,1) fwd/bwd, NHWC.
// Both backward kernels are written in GATHER form (walk the gradient of the input, sum the
// contributions of the few outputs that touch it): no atomics, bitwise reproducible.
//
// Replaces (reference): Upsample() = F.interpolate(mode='bilinear', align_corners=True)
// (mynn.py:114-119; call sites deepv3.py:121, 351, 356, 361) and nn.MaxPool2d(3, 2, 1)
// (Resnet.py:551, deepv3.py:315).
#include "common.hpp"

namespace mrfp {

// ATen: scale = (in-1)/(out-1) in float (0 when out == 1); src = scale*dst; i0 = (int)src;
// lambda1 = src - i0; i1 = i0 + (i0 < in-1)
__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap ac_tap(float scale, int dst, int in) {
    const float src = scale * (float)dst;
    Tap t;
    t.i0 = (int)src;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bilinear_fwd_kernel(const T* __restrict__ x, const T* __restrict__ addend,
                                                                T* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo,
                                                                int C, int ldi, int ly, int ldo) {
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    if (trow >= L.rowthreads) return;
    const float sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
    for (int cv = tcol; cv < L.lpr; cv += kThreads) {
        for (int oh = j; oh < Ho; oh += ly) {
            const Tap th = ac_tap(sh, oh, Hi);
            const T* r0 = x + ((size_t)b * Hi + th.i0) * Wi * ldi + (size_t)cv * VEC;
            const T* r1 = x + ((size_t)b * Hi + th.i1) * Wi * ldi + (size_t)cv * VEC;
            const size_t dl = ((size_t)b * Ho + oh) * Wo * ldo + (size_t)cv * VEC;     // (ldo = C unless y is a channel block of a wider tensor)
            // 4 output pixels per trip, all 16-20 loads issued before the first use
            for (int ow0 = trow; ow0 < Wo; ow0 += 4 * L.rowthreads) {
                Tap tw[4];
                VecT<T, VEC> ra[4], rb[4], rc[4], rd[4], re[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {          // unconditional loads, clamped column (tail masked at the store)
                    const int ow = min(ow0 + u * L.rowthreads, Wo - 1);
                    tw[u] = ac_tap(sw, ow, Wi);
                    ra[u] = load_raw<T, VEC>(r0 + (size_t)tw[u].i0 * ldi);
                    rb[u] = load_raw<T, VEC>(r0 + (size_t)tw[u].i1 * ldi);
                    rc[u] = load_raw<T, VEC>(r1 + (size_t)tw[u].i0 * ldi);
                    rd[u] = load_raw<T, VEC>(r1 + (size_t)tw[u].i1 * ldi);
                    if (addend) re[u] = load_raw<T, VEC>(addend + dl + (size_t)ow * ldo);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ow = ow0 + u * L.rowthreads;
                    if (ow >= Wo) continue;
                    float a[VEC], bb[VEC], c[VEC], d[VEC], o[VEC];
                    cvt_f<T, VEC>(ra[u], a);
                    cvt_f<T, VEC>(rb[u], bb);
                    cvt_f<T, VEC>(rc[u], c);
                    cvt_f<T, VEC>(rd[u], d);
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        o[i] = th.l0 * (tw[u].l0 * a[i] + tw[u].l1 * bb[i]) + th.l1 * (tw[u].l0 * c[i] + tw[u].l1 * d[i]);
                    if (addend) {
                        float e[VEC];
                        cvt_f<T, VEC>(re[u], e);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) o[i] += e[i];
                    }
                    store_f<T, VEC>(y + dl + (size_t)ow * ldo, o);
                }
            }
        }
    }
}

// weight with which destination index `dst` reads source index `src`
__device__ __forceinline__ float ac_weight(float scale, int dst, int in, int src) {
    const Tap t = ac_tap(scale, dst, in);
    return (t.i0 == src ? t.l0 : 0.f) + (t.i1 == src ? t.l1 : 0.f);
}
// destination range that can touch source index `src`: src-1 < dst*(in-1)/(out-1) < src+1, in exact integer
// arithmetic (a destination whose float position rounds onto the open boundary carries a weight of ~1 ulp; every
// candidate's weight is re-evaluated with the forward's float expression, so the range only bounds the work)
__device__ __forceinline__ void ac_range(int in, int out, int src, int& lo, int& hi) {
    if (in <= 1 || out <= 1) { lo = 0; hi = out - 1; return; }
    const long long a = in - 1, b = out - 1;
    lo = src >= 1 ? (int)(((long long)(src - 1) * b) / a) : 0;            // floor((src-1)*b/a): weight may be 0, cheap
    hi = (int)(((long long)(src + 1) * b + a - 1) / a);                     // ceil((src+1)*b/a)
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

// KW = compile-time bound on the number of destination columns that can touch one source column (0: unbounded,
// plain nested loops).  With a bound the column weights are computed once per source pixel and the KW loads of a
// destination row are issued together; the accumulation order (rows outer, columns inner) is the same in both forms.
// KN (round 5; 0: off) = a tighter compile-time bound on the destination columns that carry a NON-ZERO weight for one source column: the open
// interval ((iw-1)/sw, (iw+1)/sw) holds at most kw - 2 of the kw candidates ac_range() brackets it with -- 5 of 8 slots at a 2x upsample, 9 of 12
// at 4x -- and they are consecutive, starting at candidate 0 or 1.  The kernel is bound by its instruction stream (a slot costs a 16-byte load,
// VEC conversions and 2 VEC arithmetic instructions whether its weight is zero or not: ~1 000 vector instructions per 16 output bytes at 2x), so
// only the window is evaluated, its last slot under a branch that is rarely taken (a fifth non-zero column occurs for ~0.5 % of the source
// columns at 2.0026x).  The same non-zero products in the same order: bit-identical output.  A float position that rounds onto the interval's
// boundary can put a ~1 ulp weight outside the window: the candidates behind it are checked, and a wave that sees one takes the full loop.
template <typename T, int VEC, int KW, int KN = 0>
__global__ __launch_bounds__(kThreads) void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B,
                                                                int Hi, int Wi, int Ho, int Wo, int C, int ldi, int ly, int ldd) {
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    if (trow >= L.rowthreads) return;
    const float sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
    for (int cv = tcol; cv < L.lpr; cv += kThreads) {
        for (int ih = j; ih < Hi; ih += ly) {
            int oh0, oh1;
            ac_range(Hi, Ho, ih, oh0, oh1);
            for (int iw = trow; iw < Wi; iw += L.rowthreads) {
                int ow0, ow1;
                ac_range(Wi, Wo, iw, ow0, ow1);
                float acc[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
                bool windowed = false;
                if constexpr (KW > 0 && KN > 0) {
                    const int kf = ac_weight(sw, ow0, Wi, iw) != 0.f ? 0 : 1;       // first candidate with a weight
                    float wn[KN > 0 ? KN : 1];
#pragma unroll
                    for (int k = 0; k < KN; ++k) wn[k] = (ow0 + kf + k <= ow1) ? ac_weight(sw, ow0 + kf + k, Wi, iw) : 0.f;
                    bool more = false;                                              // a weight behind the window (boundary rounding)?
#pragma unroll
                    for (int k = KN; k < KW; ++k) more = more || (ow0 + kf + k <= ow1 && ac_weight(sw, ow0 + kf + k, Wi, iw) != 0.f);
                    if (__builtin_amdgcn_ballot_w64(more) == 0) {
                        windowed = true;
                        for (int oh = oh0; oh <= oh1; ++oh) {
                            const float wh = ac_weight(sh, oh, Hi, ih);
                            if (wh == 0.f) continue;
                            const T* dl = dy + ((size_t)b * Ho + oh) * Wo * ldd + (size_t)cv * VEC;
                            VecT<T, VEC> r[KN > 0 ? KN : 1];
#pragma unroll
                            for (int k = 0; k < KN; ++k) r[k] = load_raw<T, VEC>(dl + (size_t)min(ow0 + kf + k, ow1) * ldd);
#pragma unroll
                            for (int k = 0; k < KN - 1; ++k) {
                                const float w = wh * wn[k];
                                float dv[VEC];
                                cvt_f<T, VEC>(r[k], dv);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) acc[i] += (wn[k] != 0.f) ? w * dv[i] : 0.f;
                            }
                            if (wn[KN - 1] != 0.f) {
                                const float w = wh * wn[KN - 1];
                                float dv[VEC];
                                cvt_f<T, VEC>(r[KN - 1], dv);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) acc[i] += w * dv[i];
                            }
                        }
                    }
                }
                if (windowed) {
                } else if (KW > 0) {
                    float ww[KW > 0 ? KW : 1];
#pragma unroll
                    for (int k = 0; k < KW; ++k) ww[k] = (ow0 + k <= ow1) ? ac_weight(sw, ow0 + k, Wi, iw) : 0.f;
                    for (int oh = oh0; oh <= oh1; ++oh) {
                        const float wh = ac_weight(sh, oh, Hi, ih);
                        if (wh == 0.f) continue;
                        const T* dl = dy + ((size_t)b * Ho + oh) * Wo * ldd + (size_t)cv * VEC;
                        // unconditional loads from clamped (always valid) columns: no branch sits between them, so
                        // all KW are in flight together; slots past the range carry weight 0 and are not added
                        VecT<T, VEC> r[KW > 0 ? KW : 1];
#pragma unroll
                        for (int k = 0; k < KW; ++k) r[k] = load_raw<T, VEC>(dl + (size_t)min(ow0 + k, ow1) * ldd);
#pragma unroll
                        for (int k = 0; k < KW; ++k) {
                            const float w = wh * ww[k];
                            float dv[VEC];
                            cvt_f<T, VEC>(r[k], dv);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) acc[i] += (ww[k] != 0.f) ? w * dv[i] : 0.f;
                        }
                    }
                } else {
                    for (int oh = oh0; oh <= oh1; ++oh) {
                        const float wh = ac_weight(sh, oh, Hi, ih);
                        if (wh == 0.f) continue;
                        const T* dl = dy + ((size_t)b * Ho + oh) * Wo * ldd + (size_t)cv * VEC;
                        for (int ow = ow0; ow <= ow1; ++ow) {
                            const float w = wh * ac_weight(sw, ow, Wi, iw);
                            if (w == 0.f) continue;
                            float dv[VEC];
                            load_f<T, VEC>(dl + (size_t)ow * ldd, dv);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) acc[i] += w * dv[i];
                        }
                    }
                }
                store_f<T, VEC>(dx + (((size_t)b * Hi + ih) * Wi + iw) * ldi + (size_t)cv * VEC, acc);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                               uint8_t* __restrict__ idx, int B, int H, int W, int Ho,
                                                               int Wo, int C, int ly, const float* __restrict__ A,
                                                               const float* __restrict__ S, int coef_per_image, int relu) {
    // A != NULL: the pooled tensor is relu(x*A + S) (the apply pass of the normalisation in front of the pool, never stored):
    // each window value goes through the arithmetic of affine_fwd_kernel and the rounding of its store before the comparison,
    // so maxima AND arg-max positions are those of the two-kernel sequence
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    if (trow >= L.rowthreads) return;
    const bool aff = A != nullptr;
    const size_t cbase = (size_t)(coef_per_image ? b : 0) * C;
    for (int cv = tcol; cv < L.lpr; cv += kThreads) {
        float ca[VEC], cs[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { ca[i] = 1.f; cs[i] = 0.f; }
        if (aff) {
            load_coef<VEC>(A + cbase + (size_t)cv * VEC, ca);
            load_coef<VEC>(S + cbase + (size_t)cv * VEC, cs);
        }
        for (int oh = j; oh < Ho; oh += ly) {
            // TWO adjacent output columns per trip: their windows share the middle column pair, so 3 x 5 loads serve both
            // (3 x 3 each before: 17 % fewer vector loads, and two stores per trip)
            for (int op = trow; 2 * op < Wo; op += L.rowthreads) {
                const int ow0 = 2 * op;
                float best[2][VEC];
                uint8_t bi[2][VEC];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { best[q][i] = -INFINITY; bi[q][i] = 0; }
                // the 15 window loads are unconditional (clamped coordinates, all in flight together); positions outside
                // the image are skipped in the comparison, so the first-maximum rule of ATen is unchanged
                VecT<T, VEC> win[15];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int ih = min(max(2 * oh - 1 + r, 0), H - 1);
#pragma unroll
                    for (int s = 0; s < 5; ++s) {
                        const int iw = min(max(2 * ow0 - 1 + s, 0), W - 1);
                        win[r * 5 + s] = load_raw<T, VEC>(x + (((size_t)b * H + ih) * W + iw) * C + (size_t)cv * VEC);
                    }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int ih = 2 * oh - 1 + r;
#pragma unroll
                    for (int s = 0; s < 5; ++s) {
                        const int iw = 2 * ow0 - 1 + s;
                        const bool inside = ih >= 0 && ih < H && iw >= 0 && iw < W;
                        float v[VEC];
                        cvt_f<T, VEC>(win[r * 5 + s], v);
                        if (aff) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                float a = v[i] * ca[i] + cs[i];
                                if (relu) a = a > 0.f ? a : 0.f;
                                v[i] = to_f(from_f<T>(a));
                            }
                        }
                        if (s < 3) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i)
                                if (inside && (v[i] > best[0][i] || v[i] != v[i])) { best[0][i] = v[i]; bi[0][i] = (uint8_t)(r * 3 + s); }
                        }
                        if (s >= 2) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i)
                                if (inside && (v[i] > best[1][i] || v[i] != v[i])) { best[1][i] = v[i]; bi[1][i] = (uint8_t)(r * 3 + s - 2); }
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (ow0 + q >= Wo) continue;
                    const size_t o = (((size_t)b * Ho + oh) * Wo + ow0 + q) * C + (size_t)cv * VEC;
                    store_f<T, VEC>(y + o, best[q]);
                    VecT<uint8_t, VEC> pk;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) pk.v[i] = bi[q][i];
                    *reinterpret_cast<VecT<uint8_t, VEC>*>(idx + o) = pk;
                }
            }
        }
    }
}

// (gather over the 1-2 x 1-2 windows that contain the pixel; a variant with four unconditional candidate loads was
//  measured slower: 403 vs 284 us at 16x384x384x128 -- it reads 1.8x the gradients)
// gradient of the pool's INPUT at the two adjacent columns 2*ip, 2*ip + 1 of input row ih (image b, channel vector cv):
// column 2k lies in output window k only, 2k+1 in k and k+1 -- the gradients and indices of outputs k, k+1 (nrow rows) are
// loaded once for both, all in flight together
template <typename T, int VEC>
__device__ __forceinline__ void pool_grad_pair(const T* __restrict__ dy, const uint8_t* __restrict__ idx, int b, int ih, int ip,
                                               int W, int Ho, int Wo, int C, int cv, float (&acc)[2][VEC]) {
    const int oh0 = ih / 2, oh1 = (ih + 1) / 2;   // windows 2*oh-1 .. 2*oh+1 containing ih
    const int nrow = (oh1 != oh0 && oh1 < Ho) ? 2 : 1;          // (uniform over the workgroup: ih is)
    const int iw0 = 2 * ip;
    const int owa = min(ip, Wo - 1), owb = min(ip + 1, Wo - 1);
    VecT<T, VEC> dr[2][2];
    VecT<uint8_t, VEC> pr[2][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        if (rr < nrow) {
            const size_t ol = ((size_t)b * Ho + (oh0 + rr)) * Wo * C + (size_t)cv * VEC;
            dr[rr][0] = load_raw<T, VEC>(dy + ol + (size_t)owa * C);
            dr[rr][1] = load_raw<T, VEC>(dy + ol + (size_t)owb * C);
            pr[rr][0] = *reinterpret_cast<const VecT<uint8_t, VEC>*>(idx + ol + (size_t)owa * C);
            pr[rr][1] = *reinterpret_cast<const VecT<uint8_t, VEC>*>(idx + ol + (size_t)owb * C);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[q][i] = 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        if (rr >= nrow) continue;
        const int oh = oh0 + rr;
        const int r = ih - (2 * oh - 1);                       // row of ih inside window oh
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {                       // output column ip + cc
            const int ow = ip + cc;
            float dv[VEC];
            cvt_f<T, VEC>(dr[rr][cc], dv);
            // input column iw0 + q sits at position s = iw0 + q - (2*ow - 1) of window ow (0 <= s <= 2 to belong)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sft = iw0 + q - (2 * ow - 1);
                const bool in_win = ow < Wo && sft >= 0 && sft <= 2 && iw0 + q < W;
                const uint8_t want = (uint8_t)(r * 3 + sft);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[q][i] += (in_win && pr[rr][cc].v[i] == want) ? dv[i] : 0.f;
            }
        }
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                               T* __restrict__ dx, int B, int H, int W, int Ho, int Wo,
                                                               int C, int ly) {
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    if (trow >= L.rowthreads) return;
    for (int cv = tcol; cv < L.lpr; cv += kThreads) {
        for (int ih = j; ih < H; ih += ly) {
            for (int ip = trow; 2 * ip < W; ip += L.rowthreads) {
                float acc[2][VEC];
                pool_grad_pair<T, VEC>(dy, idx, b, ih, ip, W, Ho, Wo, C, cv, acc);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (2 * ip + q < W) store_f<T, VEC>(dx + (((size_t)b * H + ih) * W + 2 * ip + q) * C + (size_t)cv * VEC, acc[q]);
            }
        }
    }
}

// The backward passes of a normalisation (+ReLU) whose output only the pool reads (the stem: norm -> ReLU -> maxpool, reference
// Resnet.py:549-551 / deepv3.py:309-315), taking the POOLED gradient: the gradient of the pool's input is rebuilt per pixel from the
// (at most four) windows that contain it -- rounded to T, as maxpool_bwd_kernel stores it -- instead of being written by one kernel
// and read back by these two.  x is the normalisation's input, the ReLU gate is (x*fA + fS) > 0 as in stats.hip / affine.hip.
//   PASS 0: partial sums  s += d, q += d*(x - mean)  -> ws[blockIdx.x][2][C]   (rows as stats_kernel MODE 1 writes them)
//   PASS 1: dx = P*d + (Q*x + R)
template <typename T, int VEC, int PASS>
__global__ __launch_bounds__(kThreads) void pool_norm_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                                 const T* __restrict__ x, T* __restrict__ dx,
                                                                 float* __restrict__ ws, int B, int H, int W, int Ho, int Wo,
                                                                 int C, int ly, const float* __restrict__ mean,
                                                                 const float* __restrict__ fA, const float* __restrict__ fS,
                                                                 const float* __restrict__ P, const float* __restrict__ Q,
                                                                 const float* __restrict__ R, int coef_per_image, int relu) {
    __shared__ float sm[PASS == 0 ? kThreads * 2 * VEC : 1];
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    const bool active = trow < L.rowthreads;
    const size_t cbase = (size_t)(coef_per_image ? b : 0) * C;
    for (int cv0 = 0; cv0 < L.lpr; cv0 += kThreads) {
        const int cv = cv0 + tcol;
        const bool on = active && cv < L.lpr;
        float s[VEC], q2[VEC], mu[VEC], fa[VEC], fs[VEC], p[VEC], qq[VEC], r[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s[i] = 0.f; q2[i] = 0.f; mu[i] = 0.f; fa[i] = 0.f; fs[i] = 1.f; p[i] = 1.f; qq[i] = 0.f; r[i] = 0.f; }
        if (on) {
            const size_t co = cbase + (size_t)cv * VEC;
            if (relu) { load_coef<VEC>(fA + co, fa); load_coef<VEC>(fS + co, fs); }
            if (PASS == 0) {
                load_coef<VEC>(mean + co, mu);
            } else {
                load_coef<VEC>(P + co, p);
                load_coef<VEC>(Q + co, qq);
                load_coef<VEC>(R + co, r);
            }
            for (int ih = j; ih < H; ih += ly) {
                const size_t sl = ((size_t)b * H + ih) * W * C + (size_t)cv * VEC;
                for (int ip = trow; 2 * ip < W; ip += L.rowthreads) {
                    const int iw1 = min(2 * ip + 1, W - 1);
                    VecT<T, VEC> xr[2];
                    if (PASS == 0) {        // (read again by PASS 1)
                        xr[0] = load_raw<T, VEC>(x + sl + (size_t)(2 * ip) * C);
                        xr[1] = load_raw<T, VEC>(x + sl + (size_t)iw1 * C);
                    } else {
                        xr[0] = load_raw_nt<T, VEC>(x + sl + (size_t)(2 * ip) * C);
                        xr[1] = load_raw_nt<T, VEC>(x + sl + (size_t)iw1 * C);
                    }
                    float acc[2][VEC];
                    pool_grad_pair<T, VEC>(dy, idx, b, ih, ip, W, Ho, Wo, C, cv, acc);
#pragma unroll
                    for (int qd = 0; qd < 2; ++qd) {
                        const bool inside = 2 * ip + qd < W;
                        float xv[VEC], o[VEC];
                        cvt_f<T, VEC>(xr[qd], xv);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const float gate = xv[i] * fa[i] + fs[i];
                            const float d = (inside && gate > 0.f) ? to_f(from_f<T>(acc[qd][i])) : 0.f;
                            if (PASS == 0) {
                                s[i] += d;
                                q2[i] += d * (xv[i] - mu[i]);
                            } else {
                                o[i] = p[i] * d + (qq[i] * xv[i] + r[i]);
                            }
                        }
                        if (PASS == 1 && inside) store_f<T, VEC>(dx + sl + (size_t)(2 * ip + qd) * C, o);
                    }
                }
            }
        }
        if constexpr (PASS == 0) {
            // combine the row-threads of each channel vector through LDS (as stats_kernel)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sm[(t * 2 + 0) * VEC + i] = s[i];
                sm[(t * 2 + 1) * VEC + i] = q2[i];
            }
            __syncthreads();
            float* out = ws + (size_t)blockIdx.x * 2 * C;
            const int nout = L.colthreads * 2 * VEC;
            for (int o = t; o < nout; o += kThreads) {
                const int oc = o / (2 * VEC), rest = o % (2 * VEC);
                if (cv0 + oc < L.lpr) {
                    float a = 0.f;
                    for (int rr = 0; rr < L.rowthreads; ++rr) a += sm[((rr * L.colthreads + oc) * 2) * VEC + rest];
                    const int stat = rest / VEC, i = rest % VEC;
                    out[(size_t)stat * C + (size_t)(cv0 + oc) * VEC + i] = a;
                }
            }
        }
    }
}

#define DISPATCH_VEC(T, C, ok, KERNEL, grid, st, ...)                                                      \
    do {                                                                                                   \
        if (pick_vec<T>(C) > 1 && (ok))                                                                    \
            hipLaunchKernelGGL((KERNEL<T, FullVec<T>::value>), grid, dim3(kThreads), 0, st, __VA_ARGS__);  \
        else                                                                                               \
            hipLaunchKernelGGL((KERNEL<T, 1>), grid, dim3(kThreads), 0, st, __VA_ARGS__);                  \
        MRFP_LAUNCH_CHECK();                                                                               \
    } while (0)

template <typename T>
static int do_bilinear_fwd(const void* x, const void* addend, void* y, int64_t B, int64_t Hi, int64_t Wi, int64_t Ho,
                           int64_t Wo, int64_t C, int64_t ldi, hipStream_t st, int64_t ldo = 0) {
    const int ly = lines_per_image(B, Ho);
    if (ldo <= 0) ldo = C;
    const bool ok = aligned16(x) && aligned16(y) && (!addend || aligned16(addend)) && ldi % FullVec<T>::value == 0 &&
                    ldo % FullVec<T>::value == 0;
    DISPATCH_VEC(T, C, ok, bilinear_fwd_kernel, dim3((unsigned)(B * ly)), st, (const T*)x, (const T*)addend, (T*)y,
                 (int)B, (int)Hi, (int)Wi, (int)Ho, (int)Wo, (int)C, (int)ldi, ly, (int)ldo);
    return 0;
}
template <typename T>
static int do_bilinear_bwd(const void* dy, void* dx, int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo,
                           int64_t C, int64_t ldi, hipStream_t st, int64_t ldd = 0) {
    const int ly = lines_per_image(B, Hi);
    if (ldd <= 0) ldd = C;
    const bool ok = aligned16(dy) && aligned16(dx) && ldi % FullVec<T>::value == 0 && ldd % FullVec<T>::value == 0;
    // destination columns per source column: ceil(2*(Wo-1)/(Wi-1)) + 2 (see ac_range)
    const int64_t kw = (Wi > 1 && Wo > 1) ? (2 * (Wo - 1) + (Wi - 1) - 1) / (Wi - 1) + 2 : (int64_t)1 << 30;
    const dim3 grid((unsigned)(B * ly));
    const int full = FullVec<T>::value;
    const bool vec = ok && pick_vec<T>(C) > 1;
#define MRFP_BWD_LAUNCH(VECV, KWV, ...)                                                                                     \
    hipLaunchKernelGGL((bilinear_bwd_kernel<T, VECV, KWV, ##__VA_ARGS__>), grid, dim3(kThreads), 0, st, (const T*)dy, (T*)dx, (int)B, \
                       (int)Hi, (int)Wi, (int)Ho, (int)Wo, (int)C, (int)ldi, ly, (int)ldd)
    static int win = -1;          // MRFP_BILINEAR_WINDOW=0: every candidate column evaluated (A/B runs, the bit-identity test)
    if (win < 0) { const char* e = getenv("MRFP_BILINEAR_WINDOW"); win = e ? atoi(e) : 1; }
    if (vec) {
        if (kw <= 4) MRFP_BWD_LAUNCH(full, 4);
        else if (kw == 7 && win) MRFP_BWD_LAUNCH(full, 8, 5);          // 2x (the loss head, the class-score upsample): 5 of 8 slots
        else if (kw <= 8) MRFP_BWD_LAUNCH(full, 8);
        else if (kw == 11 && win) MRFP_BWD_LAUNCH(full, 12, 9);        // 4x (the decoder): 9 of 12
        else if (kw <= 12) MRFP_BWD_LAUNCH(full, 12);
        else MRFP_BWD_LAUNCH(full, 0);
    } else {
        if (kw <= 4) MRFP_BWD_LAUNCH(1, 4);
        else if (kw <= 8) MRFP_BWD_LAUNCH(1, 8);
        else if (kw <= 12) MRFP_BWD_LAUNCH(1, 12);
        else MRFP_BWD_LAUNCH(1, 0);
    }
#undef MRFP_BWD_LAUNCH
    MRFP_LAUNCH_CHECK();
    return 0;
}
template <typename T>
static int do_maxpool_fwd(const void* x, void* y, uint8_t* idx, int64_t B, int64_t H, int64_t W, int64_t C,
                          hipStream_t st, const float* A = nullptr, const float* S = nullptr, int per_image = 0,
                          int relu = 0) {
    const int Ho = (int)((H + 2 - 3) / 2 + 1), Wo = (int)((W + 2 - 3) / 2 + 1);
    const int ly = lines_per_image(B, Ho);
    const bool ok = aligned16(x) && aligned16(y) && ((uintptr_t)idx % FullVec<T>::value) == 0 &&
                    (!A || (aligned16(A) && aligned16(S) && C % 4 == 0));
    DISPATCH_VEC(T, C, ok, maxpool_fwd_kernel, dim3((unsigned)(B * ly)), st, (const T*)x, (T*)y, idx, (int)B, (int)H,
                 (int)W, Ho, Wo, (int)C, ly, A, S, per_image, relu);
    return 0;
}
template <typename T>
static int do_pool_norm_bwd(int pass, const void* dy, const uint8_t* idx, const void* x, void* dx, float* ws, int64_t B,
                            int64_t H, int64_t W, int64_t C, const float* mean, const float* fA, const float* fS,
                            const float* P, const float* Q, const float* R, int per_image, int relu, hipStream_t st) {
    const int Ho = (int)((H + 2 - 3) / 2 + 1), Wo = (int)((W + 2 - 3) / 2 + 1);
    const int ly = lines_per_image(B, H);
    bool ok = aligned16(dy) && aligned16(x) && (!dx || aligned16(dx)) && ((uintptr_t)idx % FullVec<T>::value) == 0 && C % 4 == 0;
    for (const float* c : {mean, fA, fS, P, Q, R}) ok = ok && (!c || aligned16(c));
    const dim3 grid((unsigned)(B * ly));
#define MRFP_PNB(VECV, PASSV)                                                                                                  \
    hipLaunchKernelGGL((pool_norm_bwd_kernel<T, VECV, PASSV>), grid, dim3(kThreads), 0, st, (const T*)dy, idx, (const T*)x,   \
                       (T*)dx, ws, (int)B, (int)H, (int)W, Ho, Wo, (int)C, ly, mean, fA, fS, P, Q, R, per_image, relu)
    if (pick_vec<T>(C) > 1 && ok) {
        if (pass == 0) MRFP_PNB(FullVec<T>::value, 0); else MRFP_PNB(FullVec<T>::value, 1);
    } else {
        if (pass == 0) MRFP_PNB(1, 0); else MRFP_PNB(1, 1);
    }
#undef MRFP_PNB
    MRFP_LAUNCH_CHECK();
    return 0;
}
template <typename T>
static int do_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int64_t B, int64_t H, int64_t W, int64_t C,
                          hipStream_t st) {
    const int Ho = (int)((H + 2 - 3) / 2 + 1), Wo = (int)((W + 2 - 3) / 2 + 1);
    const int ly = lines_per_image(B, H);
    const bool ok = aligned16(dy) && aligned16(dx) && ((uintptr_t)idx % FullVec<T>::value) == 0;
    DISPATCH_VEC(T, C, ok, maxpool_bwd_kernel, dim3((unsigned)(B * ly)), st, (const T*)dy, idx, (T*)dx, (int)B, (int)H,
                 (int)W, Ho, Wo, (int)C, ly);
    return 0;
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int mrfp_bilinear_fwd(const void* x, const void* addend, void* y, int dtype, int64_t B, int64_t Hi, int64_t Wi,
                      int64_t Ho, int64_t Wo, int64_t C, int64_t ld_in, void* stream) {
    MRFP_CHECK(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && ld_in >= C, "bilinear_fwd: bad arguments");
    if (dtype == MRFP_F32) return do_bilinear_fwd<float>(x, addend, y, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream);
    if (dtype == MRFP_BF16) return do_bilinear_fwd<bf16>(x, addend, y, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream);
    if (dtype == MRFP_F16) return do_bilinear_fwd<f16>(x, addend, y, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream);
    MRFP_CHECK(false, "bilinear_fwd: unknown dtype %d", dtype);
}
int mrfp_bilinear_bwd(const void* dy, void* dx, int dtype, int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo,
                      int64_t C, int64_t ld_in, void* stream) {
    MRFP_CHECK(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && ld_in >= C, "bilinear_bwd: bad arguments");
    if (dtype == MRFP_F32) return do_bilinear_bwd<float>(dy, dx, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream);
    if (dtype == MRFP_BF16) return do_bilinear_bwd<bf16>(dy, dx, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream);
    if (dtype == MRFP_F16) return do_bilinear_bwd<f16>(dy, dx, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream);
    MRFP_CHECK(false, "bilinear_bwd: unknown dtype %d", dtype);
}
/* The same with the OUTPUT (forward) / the incoming gradient (backward) being a block of C channels inside a wider NHWC tensor
 * of ld_out channels per pixel (y / dy point at the block's first channel): Upsample() writing straight into its slot of a
 * torch.cat(dim=1) buffer and reading its slice of that buffer's gradient (reference deepv3.py:349-353), no copy either way. */
int mrfp_bilinear_fwd_into(const void* x, void* y, int dtype, int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo,
                           int64_t C, int64_t ld_in, int64_t ld_out, void* stream) {
    MRFP_CHECK(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && ld_in >= C && ld_out >= C, "bilinear_fwd_into: bad arguments");
    if (dtype == MRFP_F32) return do_bilinear_fwd<float>(x, nullptr, y, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream, ld_out);
    if (dtype == MRFP_BF16) return do_bilinear_fwd<bf16>(x, nullptr, y, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream, ld_out);
    if (dtype == MRFP_F16) return do_bilinear_fwd<f16>(x, nullptr, y, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream, ld_out);
    MRFP_CHECK(false, "bilinear_fwd_into: unknown dtype %d", dtype);
}
int mrfp_bilinear_bwd_from(const void* dy, void* dx, int dtype, int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo,
                           int64_t C, int64_t ld_in, int64_t ld_dy, void* stream) {
    MRFP_CHECK(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && ld_in >= C && ld_dy >= C, "bilinear_bwd_from: bad arguments");
    if (dtype == MRFP_F32) return do_bilinear_bwd<float>(dy, dx, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream, ld_dy);
    if (dtype == MRFP_BF16) return do_bilinear_bwd<bf16>(dy, dx, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream, ld_dy);
    if (dtype == MRFP_F16) return do_bilinear_bwd<f16>(dy, dx, B, Hi, Wi, Ho, Wo, C, ld_in, (hipStream_t)stream, ld_dy);
    MRFP_CHECK(false, "bilinear_bwd_from: unknown dtype %d", dtype);
}
int mrfp_maxpool_fwd(const void* x, void* y, uint8_t* idx, int dtype, int64_t B, int64_t H, int64_t W, int64_t C,
                     void* stream) {
    MRFP_CHECK(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0, "maxpool_fwd: bad arguments");
    if (dtype == MRFP_F32) return do_maxpool_fwd<float>(x, y, idx, B, H, W, C, (hipStream_t)stream);
    if (dtype == MRFP_BF16) return do_maxpool_fwd<bf16>(x, y, idx, B, H, W, C, (hipStream_t)stream);
    if (dtype == MRFP_F16) return do_maxpool_fwd<f16>(x, y, idx, B, H, W, C, (hipStream_t)stream);
    MRFP_CHECK(false, "maxpool_fwd: unknown dtype %d", dtype);
}
int mrfp_maxpool_affine_fwd(const void* x, const float* A, const float* S, int coef_per_image, int relu, void* y, uint8_t* idx,
                            int dtype, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    MRFP_CHECK(x && A && S && y && idx && B > 0 && H > 0 && W > 0 && C > 0, "maxpool_affine_fwd: bad arguments");
    if (dtype == MRFP_F32) return do_maxpool_fwd<float>(x, y, idx, B, H, W, C, (hipStream_t)stream, A, S, coef_per_image, relu);
    if (dtype == MRFP_BF16) return do_maxpool_fwd<bf16>(x, y, idx, B, H, W, C, (hipStream_t)stream, A, S, coef_per_image, relu);
    if (dtype == MRFP_F16) return do_maxpool_fwd<f16>(x, y, idx, B, H, W, C, (hipStream_t)stream, A, S, coef_per_image, relu);
    MRFP_CHECK(false, "maxpool_affine_fwd: unknown dtype %d", dtype);
}
int mrfp_pool_norm_bwd_stats(const void* dy, const uint8_t* idx, const void* x, const float* mean, const float* fA,
                             const float* fS, int coef_per_image, int relu, float* ws, int dtype, int64_t B, int64_t H,
                             int64_t W, int64_t C, void* stream) {
    MRFP_CHECK(dy && idx && x && mean && ws && (!relu || (fA && fS)) && B > 0 && H > 0 && W > 0 && C > 0,
               "pool_norm_bwd_stats: bad arguments");
#define MRFP_PNS(TT) return do_pool_norm_bwd<TT>(0, dy, idx, x, nullptr, ws, B, H, W, C, mean, fA, fS, nullptr, nullptr, nullptr, \
                                                 coef_per_image, relu, (hipStream_t)stream)
    if (dtype == MRFP_F32) MRFP_PNS(float);
    if (dtype == MRFP_BF16) MRFP_PNS(bf16);
    if (dtype == MRFP_F16) MRFP_PNS(f16);
#undef MRFP_PNS
    MRFP_CHECK(false, "pool_norm_bwd_stats: unknown dtype %d", dtype);
}
int mrfp_pool_norm_bwd_apply(const void* dy, const uint8_t* idx, const void* x, const float* P, const float* Q, const float* R,
                             const float* fA, const float* fS, int coef_per_image, int relu, void* dx, int dtype, int64_t B,
                             int64_t H, int64_t W, int64_t C, void* stream) {
    MRFP_CHECK(dy && idx && x && P && Q && R && dx && (!relu || (fA && fS)) && B > 0 && H > 0 && W > 0 && C > 0,
               "pool_norm_bwd_apply: bad arguments");
#define MRFP_PNA(TT) return do_pool_norm_bwd<TT>(1, dy, idx, x, dx, nullptr, B, H, W, C, nullptr, fA, fS, P, Q, R, coef_per_image, \
                                                 relu, (hipStream_t)stream)
    if (dtype == MRFP_F32) MRFP_PNA(float);
    if (dtype == MRFP_BF16) MRFP_PNA(bf16);
    if (dtype == MRFP_F16) MRFP_PNA(f16);
#undef MRFP_PNA
    MRFP_CHECK(false, "pool_norm_bwd_apply: unknown dtype %d", dtype);
}
int mrfp_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype, int64_t B, int64_t H, int64_t W,
                     int64_t C, void* stream) {
    MRFP_CHECK(dy && dx && idx && B > 0 && H > 0 && W > 0 && C > 0, "maxpool_bwd: bad arguments");
    if (dtype == MRFP_F32) return do_maxpool_bwd<float>(dy, idx, dx, B, H, W, C, (hipStream_t)stream);
    if (dtype == MRFP_BF16) return do_maxpool_bwd<bf16>(dy, idx, dx, B, H, W, C, (hipStream_t)stream);
    if (dtype == MRFP_F16) return do_maxpool_bwd<f16>(dy, idx, dx, B, H, W, C, (hipStream_t)stream);
    MRFP_CHECK(false, "maxpool_bwd: unknown dtype %d", dtype);
}

}  // extern "C"
