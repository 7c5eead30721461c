// affine.hip -- the "apply" half of every normalisation / perturbation on the hot path:
//   y = act(x[src]*A[g,c] + S[g,c] + res)           (BN / IN / NP+ apply, nearest-resize gather)
//   dx = sum_{dst->src} P*dy' + n*(Q*x + R)          (their backward, gather form, no atomics)
// One read of each input, one write of each output, 16 bytes per lane per instruction.
//
// Replaces (reference): the normalise+affine(+ReLU)(+residual add) tails of F.batch_norm /
// InstanceNorm2d / NP+ (Resnet.py:202-225, deepv3.py:276) and F.interpolate(mode='nearest')
// feeding BatchNorm+ReLU in the HRFP branch (deepv3.py:320-327).
#include "common.hpp"

namespace mrfp {

template <typename T, int VEC, bool RESIZE>
__global__ __launch_bounds__(kThreads) void affine_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                              T* __restrict__ y, RowGeom g, int ly,
                                                              const float* __restrict__ A, const float* __restrict__ S,
                                                              int coef_per_image, int relu, uint8_t* __restrict__ mask_out) {
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(g.C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    if (trow >= L.rowthreads) return;
    const size_t cbase = (size_t)(coef_per_image ? b : 0) * g.C;
    for (int cv = tcol; cv < L.lpr; cv += kThreads) {
        float a[VEC], s[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { a[i] = 1.f; s[i] = 0.f; }
        if (A) load_coef<VEC>(A + cbase + (size_t)cv * VEC, a);
        if (S) load_coef<VEC>(S + cbase + (size_t)cv * VEC, s);
        for (int oh = j; oh < g.Ho; oh += ly) {
            const int ih = RESIZE ? g.tabH[oh] : oh;
            const T* xl = x ? x + ((size_t)b * g.Hs + ih) * g.Ws * g.C + (size_t)cv * VEC : nullptr;
            const size_t dl = ((size_t)b * g.Ho + oh) * g.Wo * g.C + (size_t)cv * VEC;
            // 4 independent pixels per trip (up to 8 16-byte loads in flight per lane)
            for (int ow0 = trow; ow0 < g.Wo; ow0 += 4 * L.rowthreads) {
                VecT<T, VEC> xr[4], rr[4];
                // unconditional loads from clamped pixel indices, RESIZE as a template parameter: see stats.hip
                int owc[4], iwv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    owc[u] = min(ow0 + u * L.rowthreads, g.Wo - 1);
                    iwv[u] = RESIZE ? g.tabW[owc[u]] : owc[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (xl) xr[u] = RESIZE ? load_raw<T, VEC>(xl + (size_t)iwv[u] * g.C)       // gathered: pixels repeat
                                           : load_raw_nt<T, VEC>(xl + (size_t)iwv[u] * g.C);
                    if (res) rr[u] = load_raw_nt<T, VEC>(res + dl + (size_t)owc[u] * g.C);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ow = ow0 + u * L.rowthreads;
                    if (ow >= g.Wo) continue;
                    float v[VEC];
                    if (xl) {
                        cvt_f<T, VEC>(xr[u], v);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[i] = v[i] * a[i] + s[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[i] = s[i];
                    }
                    if (res) {
                        float r[VEC];
                        cvt_f<T, VEC>(rr[u], r);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[i] += r[i];
                    }
                    if (relu) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                    }
                    store_f<T, VEC>(y + dl + (size_t)ow * g.C, v);
                    if constexpr (VEC == 8) {
                        // sign mask of the STORED (rounded) output, one bit per element: what the backward passes of a residual
                        // block need of y (the ReLU gate) at 1/16 of its bytes
                        if (mask_out) {
                            unsigned m = 0;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) m |= (to_f(from_f<T>(v[i])) > 0.f ? 1u : 0u) << i;
                            mask_out[(dl + (size_t)ow * g.C) >> 3] = (uint8_t)m;
                        }
                    }
                }
            }
        }
    }
}

// The same apply pass (identity geometry) that ALSO emits the per-workgroup partial sums of its STORED output -- exactly the rows
// `stats_kernel<T, VEC, 0, false>` would write for y (same grid, same lines per workgroup, same per-thread accumulation order, same
// LDS combination: bit-identical partials) -- for a consumer that normalises y per image next: the InstanceNorm behind a residual
// tail (reference Resnet.py:218-225, the `iw` taps), NP+ behind an InstanceNorm (deepv3.py:333-335).  One pass over y disappears.
template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void affine_fwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                                    T* __restrict__ y, RowGeom g, int ly,
                                                                    const float* __restrict__ A, const float* __restrict__ S,
                                                                    int coef_per_image, int relu, float* __restrict__ ws) {
    __shared__ float sm[kThreads * 2 * VEC];
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(g.C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    const bool active = trow < L.rowthreads;
    const size_t cbase = (size_t)(coef_per_image ? b : 0) * g.C;
    float* out = ws + (size_t)blockIdx.x * 2 * g.C;
    for (int cv0 = 0; cv0 < L.lpr; cv0 += kThreads) {
        const int cv = cv0 + tcol;
        const bool on = active && cv < L.lpr;
        float a[VEC], s[VEC], su[VEC], sq[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { a[i] = 1.f; s[i] = 0.f; su[i] = 0.f; sq[i] = 0.f; }
        if (on) {
            if (A) load_coef<VEC>(A + cbase + (size_t)cv * VEC, a);
            if (S) load_coef<VEC>(S + cbase + (size_t)cv * VEC, s);
            for (int oh = j; oh < g.Ho; oh += ly) {
                const size_t dl = ((size_t)b * g.Ho + oh) * g.Wo * g.C + (size_t)cv * VEC;
                for (int ow0 = trow; ow0 < g.Wo; ow0 += 4 * L.rowthreads) {
                    VecT<T, VEC> xr[4], rr[4];
                    int owc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) owc[u] = min(ow0 + u * L.rowthreads, g.Wo - 1);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xr[u] = load_raw_nt<T, VEC>(x + dl + (size_t)owc[u] * g.C);
                        if (res) rr[u] = load_raw_nt<T, VEC>(res + dl + (size_t)owc[u] * g.C);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ow = ow0 + u * L.rowthreads;
                        const bool inside = ow < g.Wo;
                        float v[VEC];
                        cvt_f<T, VEC>(xr[u], v);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[i] = v[i] * a[i] + s[i];
                        if (res) {
                            float r[VEC];
                            cvt_f<T, VEC>(rr[u], r);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[i] += r[i];
                        }
                        if (relu) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                        }
                        if (inside) store_f<T, VEC>(y + dl + (size_t)ow * g.C, v);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {        // what the statistics pass would read back: the rounded stored value
                            const float vr = inside ? to_f(from_f<T>(v[i])) : 0.f;
                            su[i] += vr;
                            sq[i] += vr * vr;
                        }
                    }
                }
            }
        }
        // combine the row-threads of each channel vector through LDS (as stats_kernel does)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sm[(t * 2 + 0) * VEC + i] = su[i];
            sm[(t * 2 + 1) * VEC + i] = sq[i];
        }
        __syncthreads();
        const int nout = L.colthreads * 2 * VEC;
        for (int o = t; o < nout; o += kThreads) {
            const int oc = o / (2 * VEC), rest = o % (2 * VEC);
            if (cv0 + oc < L.lpr) {
                float acc = 0.f;
                for (int r = 0; r < L.rowthreads; ++r) acc += sm[((r * L.colthreads + oc) * 2) * VEC + rest];
                const int stat = rest / VEC, i = rest % VEC;
                out[(size_t)stat * g.C + (size_t)(cv0 + oc) * VEC + i] = acc;
            }
        }
    }
}

// Backward, walking SOURCE lines.  invH[2*ih], invH[2*ih+1] = half-open range of destination rows that
// read source row ih (NULL = identity).  dres (destination geometry) is only supported with
// identity maps (the residual branches of the network never sit behind a resize).
template <typename T, int VEC, int KR, bool YM = false>      // YM: `y` is the 1-bit-per-element sign mask written by affine_fwd_kernel
__global__ __launch_bounds__(kThreads) void affine_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const T* __restrict__ y, T* __restrict__ dx,
                                                              T* __restrict__ dres, RowGeom g, int ly,
                                                              const int32_t* __restrict__ invH,
                                                              const int32_t* __restrict__ invW,
                                                              const float* __restrict__ P, const float* __restrict__ Q,
                                                              const float* __restrict__ R, const float* __restrict__ fA,
                                                              const float* __restrict__ fS, int coef_per_image) {
    // here g.Hs/g.Ws is the geometry of dx (the walked tensor), g.Ho/g.Wo that of dy
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(g.C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    if (trow >= L.rowthreads) return;
    const size_t cbase = (size_t)(coef_per_image ? b : 0) * g.C;
    for (int cv = tcol; cv < L.lpr; cv += kThreads) {
        float p[VEC], q[VEC], r[VEC], fa[VEC], fs[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { p[i] = 1.f; q[i] = 0.f; r[i] = 0.f; fa[i] = 0.f; fs[i] = 1.f; }
        if (P) load_coef<VEC>(P + cbase + (size_t)cv * VEC, p);
        if (Q) load_coef<VEC>(Q + cbase + (size_t)cv * VEC, q);
        if (R) load_coef<VEC>(R + cbase + (size_t)cv * VEC, r);
        const bool remask = !y && fA && x;      // ReLU mask recomputed as (x*A + S) > 0 -- one tensor less to read
        if (remask) {
            load_coef<VEC>(fA + cbase + (size_t)cv * VEC, fa);
            load_coef<VEC>(fS + cbase + (size_t)cv * VEC, fs);
        }
        if (KR < 0) {
            // identity geometry (own instantiation, KR = -1: keeps its register count independent of the resize paths) (every BN / IN / NP+ outside the HRFP branch): 4 pixels per trip, loads first
            for (int ih = j; ih < g.Hs; ih += ly) {
                const size_t sl = ((size_t)b * g.Hs + ih) * g.Ws * g.C + (size_t)cv * VEC;
                for (int iw0 = trow; iw0 < g.Ws; iw0 += 4 * L.rowthreads) {
                    VecT<T, VEC> dr[4], xr[4], yr[4];
                    unsigned mb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {      // unconditional, clamped (see stats.hip)
                        const int iw = min(iw0 + u * L.rowthreads, g.Ws - 1);
                        dr[u] = load_raw_nt<T, VEC>(dy + sl + (size_t)iw * g.C);
                        if (x && (Q || remask)) xr[u] = load_raw_nt<T, VEC>(x + sl + (size_t)iw * g.C);
                        if constexpr (YM) mb[u] = reinterpret_cast<const uint8_t*>(y)[(sl + (size_t)iw * g.C) >> 3];
                        else if (y) yr[u] = load_raw_nt<T, VEC>(y + sl + (size_t)iw * g.C);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int iw = iw0 + u * L.rowthreads;
                        if (iw >= g.Ws) continue;
                        float dv[VEC], xv[VEC], o[VEC];
                        cvt_f<T, VEC>(dr[u], dv);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) xv[i] = 0.f;
                        if (x && (Q || remask)) cvt_f<T, VEC>(xr[u], xv);
                        if constexpr (YM) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) dv[i] = ((mb[u] >> i) & 1u) ? dv[i] : 0.f;
                        } else if (y) {
                            float yv[VEC];
                            cvt_f<T, VEC>(yr[u], yv);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) dv[i] = yv[i] > 0.f ? dv[i] : 0.f;
                        } else if (remask) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) dv[i] = (xv[i] * fa[i] + fs[i] > 0.f) ? dv[i] : 0.f;
                        }
                        if (dres) store_f<T, VEC>(dres + sl + (size_t)iw * g.C, dv);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) o[i] = p[i] * dv[i] + (q[i] * xv[i] + r[i]);
                        store_f<T, VEC>(dx + sl + (size_t)iw * g.C, o);
                    }
                }
            }
            continue;
        }
        if (KR >= 0)
        for (int ih = j; ih < g.Hs; ih += ly) {
            const int oh0 = invH ? invH[2 * ih] : ih, oh1 = invH ? invH[2 * ih + 1] : ih + 1;
            const size_t sl = ((size_t)b * g.Hs + ih) * g.Ws * g.C + (size_t)cv * VEC;
            const bool need_x = x && (Q || remask);
            // one source pixel by the plain loops (any fan-in; y / dres supported)
            auto slow_pixel = [&](int iw, int ow0, int ow1) {
                float acc[VEC], xv[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) { acc[i] = 0.f; xv[i] = 0.f; }
                const float n = (float)((oh1 - oh0) * (ow1 - ow0));
                if (need_x) load_f<T, VEC>(x + sl + (size_t)iw * g.C, xv);
                for (int oh = oh0; oh < oh1; ++oh) {
                    const size_t dl = ((size_t)b * g.Ho + oh) * g.Wo * g.C + (size_t)cv * VEC;
                    for (int ow = ow0; ow < ow1; ++ow) {
                        float dv[VEC];
                        load_f<T, VEC>(dy + dl + (size_t)ow * g.C, dv);
                        if (y) {
                            float yv[VEC];
                            load_f<T, VEC>(y + dl + (size_t)ow * g.C, yv);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) dv[i] = yv[i] > 0.f ? dv[i] : 0.f;
                        } else if (remask) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) dv[i] = (xv[i] * fa[i] + fs[i] > 0.f) ? dv[i] : 0.f;
                        }
                        if (dres) store_f<T, VEC>(dres + dl + (size_t)ow * g.C, dv);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc[i] += dv[i];
                    }
                }
                float o[VEC];
                if (x && Q) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o[i] = p[i] * acc[i] + n * (q[i] * xv[i] + r[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o[i] = p[i] * acc[i] + n * r[i];
                }
                store_f<T, VEC>(dx + sl + (size_t)iw * g.C, o);
            };
            if constexpr (KR > 0) {
                // bounded fan-in (nearest resize by a small factor; launched only without y / dres): U source pixels per trip,
                // their index-table entries first, then every candidate gradient (KR*KR per pixel, loaded unconditionally from
                // clamped positions; absent ones are masked out) and x -- all in flight together, converted on use.  (One
                // pixel per trip left a dependent table -> gradient load chain exposed on every pixel: 3.2-3.7 TB/s.)
                constexpr int U = KR == 1 ? 4 : 1;       // (KR = 2 with two pixels per trip: 314 vs 300 us)
                for (int iw0 = trow; iw0 < g.Ws; iw0 += U * L.rowthreads) {
                    int iwu[U], ow0[U], ow1[U];
                    bool fast = oh1 - oh0 <= KR;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        iwu[u] = min(iw0 + u * L.rowthreads, g.Ws - 1);
                        ow0[u] = invW[2 * iwu[u]];
                        ow1[u] = invW[2 * iwu[u] + 1];
                        fast = fast && ow1[u] - ow0[u] <= KR;
                    }
                    if (!fast) {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            if (iw0 + u * L.rowthreads < g.Ws) slow_pixel(iwu[u], ow0[u], ow1[u]);
                        continue;
                    }
                    VecT<T, VEC> xr[U], dr[U][KR * KR];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (need_x) xr[u] = KR == 1 ? load_raw_nt<T, VEC>(x + sl + (size_t)iwu[u] * g.C) : load_raw<T, VEC>(x + sl + (size_t)iwu[u] * g.C);
#pragma unroll
                        for (int kh = 0; kh < KR; ++kh) {
                            const int oh = min(oh0 + kh, g.Ho - 1);
                            const size_t dl = ((size_t)b * g.Ho + oh) * g.Wo * g.C + (size_t)cv * VEC;
#pragma unroll
                            for (int kw = 0; kw < KR; ++kw)
                                dr[u][kh * KR + kw] = KR == 1 ? load_raw_nt<T, VEC>(dy + dl + (size_t)min(ow0[u] + kw, g.Wo - 1) * g.C)
                                                              : load_raw<T, VEC>(dy + dl + (size_t)min(ow0[u] + kw, g.Wo - 1) * g.C);
                                // (KR > 1: the clamped candidates of neighbouring source pixels overlap -- the non-temporal policy
                                //  evicted lines the next pixel re-reads: 300 -> 361 us)
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (iw0 + u * L.rowthreads >= g.Ws) continue;
                        float acc[VEC], xv[VEC], keep[VEC];      // the ReLU mask depends on the source pixel only
#pragma unroll
                        for (int i = 0; i < VEC; ++i) { acc[i] = 0.f; xv[i] = 0.f; }
                        if (need_x) cvt_f<T, VEC>(xr[u], xv);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) keep[i] = (!remask || xv[i] * fa[i] + fs[i] > 0.f) ? 1.f : 0.f;
#pragma unroll
                        for (int kh = 0; kh < KR; ++kh)
#pragma unroll
                            for (int kw = 0; kw < KR; ++kw) {
                                const bool have = oh0 + kh < oh1 && ow0[u] + kw < ow1[u];
                                float dv[VEC];
                                cvt_f<T, VEC>(dr[u][kh * KR + kw], dv);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) acc[i] += (have && keep[i] != 0.f) ? dv[i] : 0.f;
                            }
                        const float n = (float)((oh1 - oh0) * (ow1[u] - ow0[u]));
                        float o[VEC];
                        if (x && Q) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) o[i] = p[i] * acc[i] + n * (q[i] * xv[i] + r[i]);
                        } else {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) o[i] = p[i] * acc[i] + n * r[i];
                        }
                        store_f<T, VEC>(dx + sl + (size_t)iwu[u] * g.C, o);
                    }
                }
            } else {
                for (int iw = trow; iw < g.Ws; iw += L.rowthreads)
                    slow_pixel(iw, invW ? invW[2 * iw] : iw, invW ? invW[2 * iw + 1] : iw + 1);
            }
        }
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void add_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                       T* __restrict__ y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kThreads) {
        float u[VEC], v[VEC];
        load_f<T, VEC>(a + i * VEC, u);
        load_f<T, VEC>(b + i * VEC, v);
#pragma unroll
        for (int k = 0; k < VEC; ++k) u[k] += v[k];
        store_f<T, VEC>(y + i * VEC, u);
    }
}

// dst[p][c0d + c] = src[p][c0s + c] for c < C over npix pixels (row pitches lds / ldd elements): one block of channels
// of an NHWC tensor copied into / out of a wider one -- torch.cat(dim=1) and its backward without the detour through a
// dense temporary.  16-byte chunks; 4 pixels in flight per thread.
template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void copy_channels_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t npix,
                                                                  int C, int lds, int c0s, int ldd, int c0d) {
    const int cpr = C / VEC;                                   // chunks per pixel
    const int64_t total = npix * cpr;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x; i0 < total; i0 += 4 * stride) {
        VecT<T, VEC> r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = min(i0 + u * stride, total - 1);
            const int64_t pix = i / cpr;
            const int ch = (int)(i - pix * cpr);
            r[u] = load_raw<T, VEC>(src + pix * lds + c0s + ch * VEC);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < total) {
                const int64_t pix = i / cpr;
                const int ch = (int)(i - pix * cpr);
                *reinterpret_cast<VecT<T, VEC>*>(dst + pix * ldd + c0d + ch * VEC) = r[u];
            }
        }
    }
}

template <typename T>
static int launch_copy_channels(const void* src, void* dst, int64_t npix, int64_t C, int64_t lds, int64_t c0s, int64_t ldd,
                                int64_t c0d, hipStream_t st) {
    const int full = FullVec<T>::value;
    const bool vec_ok = C % full == 0 && lds % full == 0 && ldd % full == 0 && c0s % full == 0 && c0d % full == 0 &&
                        aligned16(src) && aligned16(dst);
    const int64_t total = npix * (vec_ok ? C / full : C);
    int64_t blocks = (total + 4 * kThreads - 1) / (4 * kThreads);
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    if (vec_ok)
        hipLaunchKernelGGL((copy_channels_kernel<T, FullVec<T>::value>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T*)src, (T*)dst, npix, (int)C, (int)lds, (int)c0s, (int)ldd, (int)c0d);
    else
        hipLaunchKernelGGL((copy_channels_kernel<T, 1>), dim3((unsigned)blocks), dim3(kThreads), 0, st, (const T*)src, (T*)dst,
                           npix, (int)C, (int)lds, (int)c0s, (int)ldd, (int)c0d);
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_affine_fwd(const void* x, const void* res, void* y, int64_t B, int64_t Ho, int64_t Wo, int64_t C,
                             int64_t Hs, int64_t Ws, const int32_t* tabH, const int32_t* tabW, const float* A,
                             const float* S, int cpi, int relu, hipStream_t st, uint8_t* mask_out = nullptr) {
    RowGeom g{(int)B, (int)Ho, (int)Wo, (int)C, (int)Hs, (int)Ws, tabH, tabW};
    const int ly = lines_per_image(B, Ho);
    dim3 grid((unsigned)(B * ly));
    const bool vec_ok = pick_vec<T>(C) > 1 && (!x || aligned16(x)) && aligned16(y) && (!res || aligned16(res)) &&
                        (!A || aligned16(A)) && (!S || aligned16(S));
    if (mask_out && !(vec_ok && FullVec<T>::value == 8)) { set_error("affine_fwd_mask: 16-bit activations with C %% 8 == 0, 16-byte aligned"); return -1; }
    const bool resize = tabH != nullptr || tabW != nullptr;
    if (resize && !(tabH && tabW)) { set_error("affine_fwd: both index tables or none"); return -1; }
#define MRFP_AFF_LAUNCH(VECV, RS)                                                                                      \
    hipLaunchKernelGGL((affine_fwd_kernel<T, VECV, RS>), grid, dim3(kThreads), 0, st, (const T*)x, (const T*)res,      \
                       (T*)y, g, ly, A, S, cpi, relu, mask_out)
    if (vec_ok) { if (resize) MRFP_AFF_LAUNCH(FullVec<T>::value, true); else MRFP_AFF_LAUNCH(FullVec<T>::value, false); }
    else { if (resize) MRFP_AFF_LAUNCH(1, true); else MRFP_AFF_LAUNCH(1, false); }
#undef MRFP_AFF_LAUNCH
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_affine_bwd(const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t B, int64_t Ho,
                             int64_t Wo, int64_t C, int64_t Hs, int64_t Ws, const int32_t* invH, const int32_t* invW,
                             const float* P, const float* Q, const float* R, const float* fA, const float* fS, int cpi,
                             hipStream_t st, bool ymask = false) {
    RowGeom g{(int)B, (int)Ho, (int)Wo, (int)C, (int)Hs, (int)Ws, nullptr, nullptr};
    const int ly = lines_per_image(B, Hs);
    dim3 grid((unsigned)(B * ly));
    const bool vec_ok = pick_vec<T>(C) > 1 && aligned16(dy) && aligned16(dx) && (!x || aligned16(x)) &&
                        (!y || aligned16(y)) && (!dres || aligned16(dres)) && (!P || aligned16(P)) &&
                        (!Q || aligned16(Q)) && (!R || aligned16(R)) && (!fA || (aligned16(fA) && aligned16(fS)));
    // fan-in of a nearest resize: ceil(out/in) destinations per source and axis (one more can appear through the float
    // rounding of ATen's index rule: the kernel checks the exact count per pixel and takes the plain loops for those)
    int kr = 0;
    if (invH && invW && !y && !dres) {
        const int64_t kh = (Ho + Hs - 1) / Hs, kw = (Wo + Ws - 1) / Ws;
        const int64_t k = kh > kw ? kh : kw;
        kr = k <= 3 ? (int)k : 0;
    }
#define MRFP_AFFB_LAUNCH(VECV, KRV)                                                                                    \
    hipLaunchKernelGGL((affine_bwd_kernel<T, VECV, KRV>), grid, dim3(kThreads), 0, st, (const T*)dy, (const T*)x,      \
                       (const T*)y, (T*)dx, (T*)dres, g, ly, invH, invW, P, Q, R, fA, fS, cpi)
    const bool identity = !invH && !invW;
    if (ymask) {
        if constexpr (FullVec<T>::value == 8) {
            if (!(vec_ok && identity && y)) { set_error("affine_bwd_mask: 16-bit activations, C %% 8 == 0, identity geometry"); return -1; }
            hipLaunchKernelGGL((affine_bwd_kernel<T, 8, -1, true>), grid, dim3(kThreads), 0, st, (const T*)dy, (const T*)x, (const T*)y,
                               (T*)dx, (T*)dres, g, ly, invH, invW, P, Q, R, fA, fS, cpi);
            MRFP_LAUNCH_CHECK();
            return 0;
        } else {
            set_error("affine_bwd_mask: 16-bit activations only");
            return -1;
        }
    }
    if (vec_ok) {
        if (identity) MRFP_AFFB_LAUNCH(FullVec<T>::value, -1);
        else if (kr == 1) MRFP_AFFB_LAUNCH(FullVec<T>::value, 1);
        else if (kr == 2) MRFP_AFFB_LAUNCH(FullVec<T>::value, 2);
        else if (kr == 3) MRFP_AFFB_LAUNCH(FullVec<T>::value, 3);
        else MRFP_AFFB_LAUNCH(FullVec<T>::value, 0);
    } else {
        if (identity) MRFP_AFFB_LAUNCH(1, -1);
        else if (kr == 1) MRFP_AFFB_LAUNCH(1, 1);
        else if (kr == 2) MRFP_AFFB_LAUNCH(1, 2);
        else if (kr == 3) MRFP_AFFB_LAUNCH(1, 3);
        else MRFP_AFFB_LAUNCH(1, 0);
    }
#undef MRFP_AFFB_LAUNCH
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_add(const void* a, const void* b, void* y, int64_t n, hipStream_t st) {
    const int full = FullVec<T>::value;
    const bool vec_ok = n % full == 0 && aligned16(a) && aligned16(b) && aligned16(y);
    const int64_t nvec = vec_ok ? n / full : n;
    int64_t blocks = (nvec + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (vec_ok)
        hipLaunchKernelGGL((add_kernel<T, FullVec<T>::value>), dim3((unsigned)blocks), dim3(kThreads), 0, st,
                           (const T*)a, (const T*)b, (T*)y, nvec);
    else
        hipLaunchKernelGGL((add_kernel<T, 1>), dim3((unsigned)blocks), dim3(kThreads), 0, st, (const T*)a, (const T*)b,
                           (T*)y, nvec);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int mrfp_affine_fwd(const void* x, const void* res, void* y, int dtype, int64_t B, int64_t Ho, int64_t Wo, int64_t C,
                    int64_t Hs, int64_t Ws, const int32_t* tabH, const int32_t* tabW, const float* A, const float* S,
                    int coef_per_image, int relu, void* stream) {
    MRFP_CHECK(y && B > 0 && Ho > 0 && Wo > 0 && C > 0 && Hs > 0 && Ws > 0, "affine_fwd: bad arguments");
    MRFP_CHECK(x || S, "affine_fwd: neither x nor S given");
    MRFP_CHECK((tabH && tabW) || (Hs == Ho && Ws == Wo), "affine_fwd: resize geometry without index tables");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return launch_affine_fwd<float>(x, res, y, B, Ho, Wo, C, Hs, Ws, tabH, tabW, A, S, coef_per_image, relu, st);
    if (dtype == MRFP_BF16) return launch_affine_fwd<bf16>(x, res, y, B, Ho, Wo, C, Hs, Ws, tabH, tabW, A, S, coef_per_image, relu, st);
    if (dtype == MRFP_F16) return launch_affine_fwd<f16>(x, res, y, B, Ho, Wo, C, Hs, Ws, tabH, tabW, A, S, coef_per_image, relu, st);
    MRFP_CHECK(false, "affine_fwd: unknown dtype %d", dtype);
}

int mrfp_affine_fwd_stats(const void* x, const void* res, void* y, int dtype, int64_t B, int64_t H, int64_t W, int64_t C,
                          const float* A, const float* S, int coef_per_image, int relu, float* ws, void* stream) {
    MRFP_CHECK(x && y && ws && B > 0 && H > 0 && W > 0 && C > 0, "affine_fwd_stats: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    RowGeom g{(int)B, (int)H, (int)W, (int)C, (int)H, (int)W, nullptr, nullptr};
    const int ly = lines_per_image(B, H);
    const bool al = aligned16(x) && aligned16(y) && (!res || aligned16(res)) && (!A || aligned16(A)) && (!S || aligned16(S));
#define MRFP_AFS(TT)                                                                                                           \
    do {                                                                                                                       \
        if (pick_vec<TT>(C) > 1 && al)                                                                                         \
            hipLaunchKernelGGL((affine_fwd_stats_kernel<TT, FullVec<TT>::value>), dim3((unsigned)(B * ly)), dim3(kThreads), 0, st, (const TT*)x,  \
                               (const TT*)res, (TT*)y, g, ly, A, S, coef_per_image, relu, ws);                                   \
        else                                                                                                                   \
            hipLaunchKernelGGL((affine_fwd_stats_kernel<TT, 1>), dim3((unsigned)(B * ly)), dim3(kThreads), 0, st, (const TT*)x,  \
                               (const TT*)res, (TT*)y, g, ly, A, S, coef_per_image, relu, ws);                                   \
    } while (0)
    if (dtype == MRFP_F32) MRFP_AFS(float);
    else if (dtype == MRFP_BF16) MRFP_AFS(bf16);
    else if (dtype == MRFP_F16) MRFP_AFS(f16);
    else MRFP_CHECK(false, "affine_fwd_stats: unknown dtype %d", dtype);
#undef MRFP_AFS
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_affine_bwd(const void* dy, const void* x, const void* y, void* dx, void* dres, int dtype, int64_t B,
                    int64_t Ho, int64_t Wo, int64_t C, int64_t Hs, int64_t Ws, const int32_t* invH, const int32_t* invW,
                    const float* P, const float* Q, const float* R, const float* fA, const float* fS, int coef_per_image,
                    void* stream) {
    MRFP_CHECK(!fA == !fS, "affine_bwd: fA and fS go together");
    MRFP_CHECK(dy && dx && B > 0 && Ho > 0 && Wo > 0 && C > 0 && Hs > 0 && Ws > 0, "affine_bwd: bad arguments");
    MRFP_CHECK((invH && invW) || (Hs == Ho && Ws == Wo), "affine_bwd: resize geometry without inverse tables");
    MRFP_CHECK(!dres || (!invH && !invW), "affine_bwd: dres is not supported behind a resize");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return launch_affine_bwd<float>(dy, x, y, dx, dres, B, Ho, Wo, C, Hs, Ws, invH, invW, P, Q, R, fA, fS, coef_per_image, st);
    if (dtype == MRFP_BF16) return launch_affine_bwd<bf16>(dy, x, y, dx, dres, B, Ho, Wo, C, Hs, Ws, invH, invW, P, Q, R, fA, fS, coef_per_image, st);
    if (dtype == MRFP_F16) return launch_affine_bwd<f16>(dy, x, y, dx, dres, B, Ho, Wo, C, Hs, Ws, invH, invW, P, Q, R, fA, fS, coef_per_image, st);
    MRFP_CHECK(false, "affine_bwd: unknown dtype %d", dtype);
}

int mrfp_affine_fwd_relu_mask(const void* x, const void* res, void* y, void* mask, int dtype, int64_t B, int64_t H, int64_t W,
                              int64_t C, const float* A, const float* S, int coef_per_image, void* stream) {
    MRFP_CHECK(x && y && mask && A && S && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "affine_fwd_relu_mask: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_BF16) return launch_affine_fwd<bf16>(x, res, y, B, H, W, C, H, W, nullptr, nullptr, A, S, coef_per_image, 1, st, (uint8_t*)mask);
    if (dtype == MRFP_F16) return launch_affine_fwd<f16>(x, res, y, B, H, W, C, H, W, nullptr, nullptr, A, S, coef_per_image, 1, st, (uint8_t*)mask);
    MRFP_CHECK(false, "affine_fwd_relu_mask: 16-bit activations only (dtype %d)", dtype);
}

int mrfp_affine_bwd_mask(const void* dy, const void* x, const void* mask, void* dx, void* dres, int dtype, int64_t B, int64_t H,
                         int64_t W, int64_t C, const float* P, const float* Q, const float* R, int coef_per_image, void* stream) {
    MRFP_CHECK(dy && dx && mask && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "affine_bwd_mask: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_BF16) return launch_affine_bwd<bf16>(dy, x, mask, dx, dres, B, H, W, C, H, W, nullptr, nullptr, P, Q, R, nullptr, nullptr, coef_per_image, st, true);
    if (dtype == MRFP_F16) return launch_affine_bwd<f16>(dy, x, mask, dx, dres, B, H, W, C, H, W, nullptr, nullptr, P, Q, R, nullptr, nullptr, coef_per_image, st, true);
    MRFP_CHECK(false, "affine_bwd_mask: 16-bit activations only (dtype %d)", dtype);
}

int mrfp_copy_channels(const void* src, void* dst, int dtype, int64_t npix, int64_t C, int64_t ld_src, int64_t c0_src,
                       int64_t ld_dst, int64_t c0_dst, void* stream) {
    MRFP_CHECK(src && dst && npix > 0 && C > 0 && c0_src >= 0 && c0_dst >= 0 && c0_src + C <= ld_src && c0_dst + C <= ld_dst,
               "copy_channels: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return launch_copy_channels<float>(src, dst, npix, C, ld_src, c0_src, ld_dst, c0_dst, st);
    if (dtype == MRFP_BF16) return launch_copy_channels<bf16>(src, dst, npix, C, ld_src, c0_src, ld_dst, c0_dst, st);
    if (dtype == MRFP_F16) return launch_copy_channels<f16>(src, dst, npix, C, ld_src, c0_src, ld_dst, c0_dst, st);
    MRFP_CHECK(false, "copy_channels: unknown dtype %d", dtype);
}

int mrfp_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream) {
    MRFP_CHECK(a && b && y && n > 0, "add: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return launch_add<float>(a, b, y, n, st);
    if (dtype == MRFP_BF16) return launch_add<bf16>(a, b, y, n, st);
    if (dtype == MRFP_F16) return launch_add<f16>(a, b, y, n, st);
    MRFP_CHECK(false, "add: unknown dtype %d", dtype);
}

}  // extern "C"
