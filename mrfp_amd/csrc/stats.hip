// stats.hip -- per-channel statistics over NHWC rows and the per-op "finalize" kernels that
// turn them into apply coefficients.  HBM-bound: every activation element is read exactly once
// with 16-byte loads; partial sums go to a small fp32 workspace, are combined in fp64.
//
// Replaces (reference): F.batch_norm statistics behind Norm2d (mynn.py:19-25), InstanceNorm2d
// (Resnet.py:176-178, 534-536), feat.mean((2,3)) / torch.std of NP+ (deepv3.py:268-277),
// AdaptiveAvgPool2d(1) (deepv3.py:109).
#include "common.hpp"

namespace mrfp {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------
// stage 1: partial sums.  grid.x = B * ly; block (b, j) walks lines oh = j, j+ly, ... of image b.
// MODE 0: s += x, q += x*x                         (forward statistics)
// MODE 1: s += dy', q += dy'*(x - mean[g,c])       (backward statistics; dy' masked by y > 0)
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, int MODE, bool RESIZE, bool YM = false>     // YM: `y` is the 1-bit-per-element sign mask (affine.hip)
__global__ __launch_bounds__(kThreads) void stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ fA, const float* __restrict__ fS,
                                                         int per_image, RowGeom g, int ly, float* __restrict__ ws) {
    __shared__ float sm[kThreads * 2 * VEC];
    const int b = blockIdx.x / ly, j = blockIdx.x % ly;
    const Lanes L = make_lanes(g.C, VEC);
    const int t = threadIdx.x;
    const int tcol = t % L.colthreads, trow = t / L.colthreads;
    const bool active = trow < L.rowthreads;
    float* out = ws + (size_t)blockIdx.x * 2 * g.C;

    for (int cv0 = 0; cv0 < L.lpr; cv0 += kThreads) {
        const int cv = cv0 + tcol;
        const bool on = active && cv < L.lpr;
        float s[VEC], q[VEC], mu[VEC], fa[VEC], fs[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s[i] = 0.f; q[i] = 0.f; mu[i] = 0.f; fa[i] = 0.f; fs[i] = 1.f; }
        if (MODE == 1 && on) {
            const size_t co = (size_t)(per_image ? b : 0) * g.C + (size_t)cv * VEC;
            if (mean != nullptr) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) mu[i] = mean[co + i];
            }
            if (fA != nullptr) {      // ReLU mask recomputed from x with the forward coefficients: (x*A + S) > 0
#pragma unroll
                for (int i = 0; i < VEC; ++i) { fa[i] = fA[co + i]; fs[i] = fS[co + i]; }
            }
        }
        const bool remask = MODE == 1 && y == nullptr && fA != nullptr;
        if (on) {
            for (int oh = j; oh < g.Ho; oh += ly) {
                const int ih = RESIZE ? g.tabH[oh] : oh;
                const T* xl = x + ((size_t)b * g.Hs + ih) * g.Ws * g.C + (size_t)cv * VEC;
                const size_t dl = ((size_t)b * g.Ho + oh) * g.Wo * g.C + (size_t)cv * VEC;
                // 4 independent pixels per trip: 4 (MODE 0) or up to 12 (MODE 1) 16-byte loads in flight per lane.
                // RESIZE is a template parameter: with a run-time `tabW ? tabW[ow] : ow` the compiler parks an
                // s_waitcnt vmcnt(0) for the table value in front of every pixel's loads and serialises them
                // (measured: 3.5 TB/s instead of 5 TB/s on the identity geometry).
                for (int ow0 = trow; ow0 < g.Wo; ow0 += 4 * L.rowthreads) {
                    // loads are UNCONDITIONAL (pixel index clamped into the line; the tail is masked in the arithmetic
                    // below): a load inside a divergent `if (ow < Wo)` region is waited for at the region's end
                    // (s_waitcnt vmcnt(0) per pixel), which leaves 2-3 loads in flight instead of 8-12
                    VecT<T, VEC> xr[4], dr[4], yr[4];
                    unsigned mb[4] = {0u, 0u, 0u, 0u};
                    int owc[4], iwv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        owc[u] = min(ow0 + u * L.rowthreads, g.Wo - 1);
                        iwv[u] = RESIZE ? g.tabW[owc[u]] : owc[u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xr[u] = load_raw<T, VEC>(xl + (size_t)iwv[u] * g.C);
                        if (MODE == 1) {
                            dr[u] = load_raw<T, VEC>(dy + dl + (size_t)owc[u] * g.C);
                            if constexpr (YM) mb[u] = reinterpret_cast<const uint8_t*>(y)[(dl + (size_t)owc[u] * g.C) >> 3];
                            else if (y != nullptr) yr[u] = load_raw<T, VEC>(y + dl + (size_t)owc[u] * g.C);
                        }
                    }
                    // branch-free arithmetic: the line tail and the ReLU mask are selects, not divergent regions
                    const bool use_y = MODE == 1 && y != nullptr && !YM;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool inside = ow0 + u * L.rowthreads < g.Wo;
                        float xv[VEC];
                        cvt_f<T, VEC>(xr[u], xv);
                        if (MODE == 0) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                const float v = inside ? xv[i] : 0.f;
                                s[i] += v;
                                q[i] += v * v;
                            }
                        } else {
                            float dv[VEC], yv[VEC];
                            cvt_f<T, VEC>(dr[u], dv);
                            if (use_y) {
                                cvt_f<T, VEC>(yr[u], yv);
                            } else {
#pragma unroll
                                for (int i = 0; i < VEC; ++i) yv[i] = 1.f;
                            }
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                const float gate = YM ? (((mb[u] >> i) & 1u) ? 1.f : 0.f)
                                                      : use_y ? yv[i] : (remask ? xv[i] * fa[i] + fs[i] : 1.f);
                                const float d = (inside && gate > 0.f) ? dv[i] : 0.f;
                                s[i] += d;
                                q[i] += d * (xv[i] - mu[i]);
                            }
                        }
                    }
                }
            }
        }
        // combine the row-threads of each channel vector through LDS
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sm[(t * 2 + 0) * VEC + i] = s[i];
            sm[(t * 2 + 1) * VEC + i] = q[i];
        }
        __syncthreads();
        const int nout = L.colthreads * 2 * VEC;   // (tcol, stat, i)
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

template <typename T, int MODE>
static int launch_stats(const void* x, const void* dy, const void* y, const float* mean, const float* fA,
                        const float* fS, int per_image,
                        int64_t B, int64_t Ho, int64_t Wo, int64_t C, int64_t Hs, int64_t Ws,
                        const int32_t* tabH, const int32_t* tabW, float* ws, hipStream_t st, bool ymask = false) {
    RowGeom g{(int)B, (int)Ho, (int)Wo, (int)C, (int)Hs, (int)Ws, tabH, tabW};
    const int ly = lines_per_image(B, Ho);
    dim3 grid((unsigned)(B * ly));
    const bool vec_ok = pick_vec<T>(C) > 1 && aligned16(x) && (MODE == 0 || (aligned16(dy) && (y == nullptr || ymask || aligned16(y))));
    const bool resize = tabH != nullptr || tabW != nullptr;
    if (resize && !(tabH && tabW)) { set_error("stats: both index tables or none"); return -1; }
    if (ymask) {
        if constexpr (MODE == 1 && FullVec<T>::value == 8) {
            if (!(vec_ok && !resize && y)) { set_error("stats_bwd_mask: 16-bit activations, C %% 8 == 0, identity geometry"); return -1; }
            hipLaunchKernelGGL((stats_kernel<T, 8, 1, false, true>), grid, dim3(kThreads), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean,
                               fA, fS, per_image, g, ly, ws);
            MRFP_LAUNCH_CHECK();
            return 0;
        } else {
            set_error("stats_bwd_mask: 16-bit activations only");
            return -1;
        }
    }
#define MRFP_STATS_LAUNCH(VECV, RS)                                                                                    \
    hipLaunchKernelGGL((stats_kernel<T, VECV, MODE, RS>), grid, dim3(kThreads), 0, st, (const T*)x, (const T*)dy,      \
                       (const T*)y, mean, fA, fS, per_image, g, ly, ws)
    if (vec_ok) { if (resize) MRFP_STATS_LAUNCH(FullVec<T>::value, true); else MRFP_STATS_LAUNCH(FullVec<T>::value, false); }
    else { if (resize) MRFP_STATS_LAUNCH(1, true); else MRFP_STATS_LAUNCH(1, false); }
#undef MRFP_STATS_LAUNCH
    MRFP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// stage 2: combine partials.  block = 32 channels x 8 partial lanes; every thread sums a strided
// subset of the partials of its channel in fp64; lanes are combined through LDS.
// ------------------------------------------------------------------------------------------
struct Red { double s, q; };

// block = (kFC channels) x (kFL partial lanes): few channels per block so that even a 64-channel layer
// spreads over 8 workgroups, many lanes so that every thread only walks n/kFL partials (all independent loads).
#ifndef MRFP_FC
#define MRFP_FC 8
#define MRFP_FL 128
#endif
constexpr int kFC = MRFP_FC, kFL = MRFP_FL;      // 1024 threads: 2048 partial rows = 16 per thread = 4 batches of independent loads
constexpr int kFold = 8;                // lanes folded per thread in the first level of the final sum
__device__ __forceinline__ Red reduce_partials(const float* ws, int64_t first, int64_t n, int64_t C, int c, bool valid,
                                               double (*sm)[2][kFC]) {
    const int ty = threadIdx.y, tx = threadIdx.x;
    double s = 0.0, q = 0.0;
    if (valid) {
        float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t p = first + ty;
        for (; p + 3 * kFL < first + n; p += 4 * kFL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                fs[u] = ws[((p + u * kFL) * 2 + 0) * C + c];
                fq[u] = ws[((p + u * kFL) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += (double)fs[u]; q += (double)fq[u]; }
        }
        for (; p < first + n; p += kFL) {
            s += (double)ws[(p * 2 + 0) * C + c];
            q += (double)ws[(p * 2 + 1) * C + c];
        }
    }
    __syncthreads();
    sm[ty][0][tx] = s;
    sm[ty][1][tx] = q;
    __syncthreads();
    // two-level sum in a fixed order: kFL/kFold lanes fold kFold entries each, lane 0 folds those
    double s1 = 0.0, q1 = 0.0;
    if (ty < kFL / kFold) {
#pragma unroll
        for (int k = 0; k < kFold; ++k) { s1 += sm[ty * kFold + k][0][tx]; q1 += sm[ty * kFold + k][1][tx]; }
    }
    __syncthreads();
    if (ty < kFL / kFold) { sm[ty][0][tx] = s1; sm[ty][1][tx] = q1; }
    __syncthreads();
    Red r{0.0, 0.0};
    if (ty == 0) {
#pragma unroll
        for (int k = 0; k < kFL / kFold; ++k) { r.s += sm[k][0][tx]; r.q += sm[k][1][tx]; }
    }
    return r;
}

__global__ void bn_finalize_kernel(const float* ws, int64_t nparts, double count, int C, const float* weight,
                                   const float* bias, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean, float* invstd, float* A, float* S) {
    __shared__ double sm[kFL][2][kFC];
    const int c = blockIdx.x * kFC + threadIdx.x;
    const bool valid = c < C;
    Red r = reduce_partials(ws, 0, nparts, C, c, valid, sm);
    if (threadIdx.y == 0 && valid) {
        const double m = r.s / count;
        double var = r.q / count - m * m;
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const float w = weight ? weight[c] : 1.f, bb = bias ? bias[c] : 0.f;
        mean[c] = (float)m;
        invstd[c] = (float)is;
        const float a = (float)(w * is);
        A[c] = a;
        S[c] = (float)(bb - m * (double)a);
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
}

__global__ void bn_eval_coef_kernel(int C, const float* weight, const float* bias, const float* rm, const float* rv,
                                    float eps, float* A, float* S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(rv[c] + eps);
    const float a = (weight ? weight[c] : 1.f) * is;
    A[c] = a;
    S[c] = (bias ? bias[c] : 0.f) - rm[c] * a;
}

// dx = w*invstd*(dy' - m1 - xhat*m2): P = w*invstd, Q = -w*invstd^3*Sq/N, R = -P*Ss/N - Q*mean
__device__ __forceinline__ void norm_bwd_coef(double Ss, double Sq, double count, float w, float mean, float invstd,
                                              float& P, float& Q, float& R) {
    const double is = invstd;
    const double p = (double)w * is;
    const double qq = -(double)w * is * is * is * Sq / count;
    P = (float)p;
    Q = (float)qq;
    R = (float)(-p * Ss / count - qq * (double)mean);
}

__global__ void bn_bwd_finalize_kernel(const float* ws, int64_t nparts, double count, int C, const float* weight,
                                       const float* mean, const float* invstd, float* dweight, float* dbias,
                                       float* P, float* Q, float* R) {
    __shared__ double sm[kFL][2][kFC];
    const int c = blockIdx.x * kFC + threadIdx.x;
    const bool valid = c < C;
    Red r = reduce_partials(ws, 0, nparts, C, c, valid, sm);
    if (threadIdx.y == 0 && valid) {
        if (dbias) dbias[c] = (float)r.s;
        if (dweight) dweight[c] = (float)(r.q * (double)invstd[c]);
        norm_bwd_coef(r.s, r.q, count, weight ? weight[c] : 1.f, mean[c], invstd[c], P[c], Q[c], R[c]);
    }
}

// InstanceNorm: grid (C/32, B)
__global__ void in_finalize_kernel(const float* ws, int64_t nslab, double count, int C, const float* weight,
                                   const float* bias, float eps, float* mean, float* invstd, float* A, float* S) {
    __shared__ double sm[kFL][2][kFC];
    const int c = blockIdx.x * kFC + threadIdx.x, b = blockIdx.y;
    const bool valid = c < C;
    Red r = reduce_partials(ws, (int64_t)b * nslab, nslab, C, c, valid, sm);
    if (threadIdx.y == 0 && valid) {
        const double m = r.s / count;
        double var = r.q / count - m * m;
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const size_t o = (size_t)b * C + c;
        mean[o] = (float)m;
        invstd[o] = (float)is;
        const float a = (float)((weight ? weight[c] : 1.f) * is);
        A[o] = a;
        S[o] = (float)((bias ? bias[c] : 0.f) - m * (double)a);
    }
}

// InstanceNorm backward: grid (C / kFC).  The kFL partial lanes are split into kFL / kIL groups of kIL lanes: one group per image,
// kIL lanes over the image's slabs, so that 16 images are reduced in ONE round of loads (they were B sequential rounds of
// reduce_partials: 36 us at B = 16 where this takes a third).  dweight / dbias are summed over the images in image order by
// one lane (fixed order: bitwise reproducible, no atomics).
constexpr int kIL = 8;
__global__ void in_bwd_finalize_kernel(const float* ws, int B, int64_t nslab, double count, int C, const float* weight,
                                       const float* mean, const float* invstd, float* dweight, float* dbias,
                                       float* P, float* Q, float* R) {
    __shared__ double sm[kFL][2][kFC];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c = blockIdx.x * kFC + tx;
    const bool valid = c < C;
    constexpr int G = kFL / kIL;            // images per round
    const int gi = ty / kIL, gl = ty % kIL;
    double dw = 0.0, db = 0.0;
    for (int b0 = 0; b0 < B; b0 += G) {
        const int b = b0 + gi;
        const bool mine = valid && b < B;
        double s = 0.0, q = 0.0;
        if (mine) {
            const float* base = ws + ((size_t)b * nslab * 2) * C + c;
            int64_t p = gl;
            for (; p + 3 * kIL < nslab; p += 4 * kIL) {
                float fs[4], fq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fs[u] = base[((p + u * kIL) * 2 + 0) * C];
                    fq[u] = base[((p + u * kIL) * 2 + 1) * C];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s += (double)fs[u]; q += (double)fq[u]; }
            }
            for (; p < nslab; p += kIL) {
                s += (double)base[(p * 2 + 0) * C];
                q += (double)base[(p * 2 + 1) * C];
            }
        }
        __syncthreads();
        sm[ty][0][tx] = s;
        sm[ty][1][tx] = q;
        __syncthreads();
        if (gl == 0 && mine) {
            double rs = 0.0, rq = 0.0;
#pragma unroll
            for (int k = 0; k < kIL; ++k) { rs += sm[ty + k][0][tx]; rq += sm[ty + k][1][tx]; }
            const size_t o = (size_t)b * C + c;
            norm_bwd_coef(rs, rq, count, weight ? weight[c] : 1.f, mean[o], invstd[o], P[o], Q[o], R[o]);
            sm[ty][0][tx] = rs;                              // (only this lane read rows ty .. ty + kIL - 1)
            sm[ty][1][tx] = rq * (double)invstd[o];
        }
        __syncthreads();
        if (ty == 0 && valid) {
            for (int g = 0; g < G && b0 + g < B; ++g) { db += sm[g * kIL][0][tx]; dw += sm[g * kIL][1][tx]; }
        }
    }
    if (ty == 0 && valid) {
        if (dweight) dweight[c] = (float)dw;
        if (dbias) dbias[c] = (float)db;
    }
}

// plane sums -> fp32 [B][C] scaled by `scale` (mean: 1/count; raw sum: 1)
__global__ void plane_sum_kernel(const float* ws, int64_t nslab, double scale, int C, float* out) {
    __shared__ double sm[kFL][2][kFC];
    const int c = blockIdx.x * kFC + threadIdx.x, b = blockIdx.y;
    const bool valid = c < C;
    Red r = reduce_partials(ws, (int64_t)b * nslab, nslab, C, c, valid, sm);
    if (threadIdx.y == 0 && valid) out[(size_t)b * C + c] = (float)(r.s * scale);
}

template <typename T>
__global__ void cast_out_kernel(const float* in, T* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = from_f<T>(in[i]);
}

// ------------------------------------------------------------------------------------------
// NP+ coefficient kernels: one workgroup (the data is [B][C], a few KB)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (t < s) sm[t] = fmaxf(sm[t], sm[t + s]);
        __syncthreads();
    }
    const float r = sm[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_sum(double v, double* sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (t < s) sm[t] += sm[t + s];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// reference deepv3.py:269-276
__global__ __launch_bounds__(kThreads) void np_coef_kernel(int B, int C, const float* alpha, const float* beta_noise,
                                                           const float* mu, float* sigma, float* A, float* S) {
    __shared__ float smf[kThreads];
    float lmax = -INFINITY;
    for (int c = threadIdx.x; c < C; c += kThreads) {
        double m = 0.0;
        for (int b = 0; b < B; ++b) m += (double)mu[(size_t)b * C + c];
        m /= B;
        double v = 0.0;
        for (int b = 0; b < B; ++b) { const double d = (double)mu[(size_t)b * C + c] - m; v += d * d; }
        const float sg = (float)sqrt(v / (double)(B - 1));   // unbiased; B==1 -> NaN like torch.std
        sigma[c] = sg;
        lmax = fmaxf(lmax, sg);
        if (sg != sg) lmax = sg;                              // propagate NaN
    }
    const float M = block_max(lmax, smf);
    for (int c = threadIdx.x; c < C; c += kThreads) {
        const float scale = sigma[c] / M * 1.5f;
        for (int b = 0; b < B; ++b) {
            const size_t o = (size_t)b * C + c;
            const float beta = 1.f + beta_noise[o] * scale;
            A[o] = alpha[o];
            S[o] = (beta - alpha[o]) * mu[o];
        }
    }
}

// Backward of the same expression.  G[b][c] = sum_hw dy.
//   dL/dbeta[b,c] = mu*G ; ds[c] = sum_b dL/dbeta*noise ; s = 1.5*sigma/M (M = sigma[c*])
//   dL/dsigma[c] = 1.5*ds[c]/M - [c==c*] * sum_c' 1.5*ds[c']*sigma[c']/M^2
//   dL/dmu[b,c]  = (beta-alpha)*G + dL/dsigma[c]*(mu-mbar)/((B-1)*sigma[c])
//   dx = alpha*dy + K,  K = dL/dmu / HW
__global__ __launch_bounds__(kThreads) void np_bwd_coef_kernel(int B, int C, double hw, const float* alpha,
                                                               const float* beta_noise, const float* mu,
                                                               const float* sigma, const float* G, float* K) {
    __shared__ float smf[kThreads];
    __shared__ double smd[kThreads];
    float lmax = -INFINITY;
    for (int c = threadIdx.x; c < C; c += kThreads) lmax = fmaxf(lmax, sigma[c]);
    const float M = block_max(lmax, smf);
    // first channel attaining the maximum (torch.max() sends the gradient to one arg-max)
    int larg = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += kThreads)
        if (sigma[c] == M && c < larg) larg = c;
    const int cstar = -(int)block_max(-(float)larg, smf);
    double lt = 0.0;
    for (int c = threadIdx.x; c < C; c += kThreads) {
        double ds = 0.0;
        for (int b = 0; b < B; ++b) {
            const size_t o = (size_t)b * C + c;
            ds += (double)mu[o] * (double)G[o] * (double)beta_noise[o];
        }
        lt += ds * (double)sigma[c];
    }
    const double T = block_sum(lt, smd);
    const double Md = (double)M;
    for (int c = threadIdx.x; c < C; c += kThreads) {
        double ds = 0.0, mbar = 0.0;
        for (int b = 0; b < B; ++b) {
            const size_t o = (size_t)b * C + c;
            ds += (double)mu[o] * (double)G[o] * (double)beta_noise[o];
            mbar += (double)mu[o];
        }
        mbar /= B;
        double dsig = 1.5 * ds / Md;
        if (c == cstar) dsig -= 1.5 * T / (Md * Md);
        const double scale = (double)sigma[c] / Md * 1.5;
        for (int b = 0; b < B; ++b) {
            const size_t o = (size_t)b * C + c;
            const double beta = 1.0 + (double)beta_noise[o] * scale;
            double dmu = (beta - (double)alpha[o]) * (double)G[o];
            dmu += dsig * ((double)mu[o] - mbar) / ((double)(B - 1) * (double)sigma[c]);
            K[o] = (float)(dmu / hw);
        }
    }
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int mrfp_version(void) { return 100; }
const char* mrfp_last_error(void) { return mrfp::g_err; }

int64_t mrfp_stats_nslab(int64_t B, int64_t Ho) { return lines_per_image(B, Ho); }

int mrfp_stats_fwd(const void* x, int dtype, int64_t B, int64_t Ho, int64_t Wo, int64_t C, int64_t Hs, int64_t Ws,
                   const int32_t* tabH, const int32_t* tabW, float* ws, void* stream) {
    MRFP_CHECK(x && ws && B > 0 && Ho > 0 && Wo > 0 && C > 0, "stats_fwd: bad arguments");
    MRFP_CHECK(C <= 65536 && B * Ho < (1LL << 31), "stats_fwd: shape out of range");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return launch_stats<float, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, B, Ho, Wo, C, Hs, Ws, tabH, tabW, ws, st);
    if (dtype == MRFP_BF16) return launch_stats<bf16, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, B, Ho, Wo, C, Hs, Ws, tabH, tabW, ws, st);
    if (dtype == MRFP_F16) return launch_stats<f16, 0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, B, Ho, Wo, C, Hs, Ws, tabH, tabW, ws, st);
    MRFP_CHECK(false, "stats_fwd: unknown dtype %d", dtype);
}

int mrfp_stats_bwd(const void* dy, const void* x, const void* y, const float* mean, const float* fA, const float* fS,
                   int per_image, int dtype,
                   int64_t B, int64_t Ho, int64_t Wo, int64_t C, int64_t Hs, int64_t Ws, const int32_t* tabH,
                   const int32_t* tabW, float* ws, void* stream) {
    MRFP_CHECK(dy && x && ws && B > 0 && Ho > 0 && Wo > 0 && C > 0, "stats_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return launch_stats<float, 1>(x, dy, y, mean, fA, fS, per_image, B, Ho, Wo, C, Hs, Ws, tabH, tabW, ws, st);
    if (dtype == MRFP_BF16) return launch_stats<bf16, 1>(x, dy, y, mean, fA, fS, per_image, B, Ho, Wo, C, Hs, Ws, tabH, tabW, ws, st);
    if (dtype == MRFP_F16) return launch_stats<f16, 1>(x, dy, y, mean, fA, fS, per_image, B, Ho, Wo, C, Hs, Ws, tabH, tabW, ws, st);
    MRFP_CHECK(false, "stats_bwd: unknown dtype %d", dtype);
}

int mrfp_stats_bwd_mask(const void* dy, const void* x, const void* mask, const float* mean, int per_image, int dtype, int64_t B,
                        int64_t H, int64_t W, int64_t C, float* ws, void* stream) {
    MRFP_CHECK(dy && x && mask && ws && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "stats_bwd_mask: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_BF16) return launch_stats<bf16, 1>(x, dy, mask, mean, nullptr, nullptr, per_image, B, H, W, C, H, W, nullptr, nullptr, ws, st, true);
    if (dtype == MRFP_F16) return launch_stats<f16, 1>(x, dy, mask, mean, nullptr, nullptr, per_image, B, H, W, C, H, W, nullptr, nullptr, ws, st, true);
    MRFP_CHECK(false, "stats_bwd_mask: 16-bit activations only (dtype %d)", dtype);
}

int mrfp_bn_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C, const float* weight,
                     const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                     float* mean, float* invstd, float* A, float* S, void* stream) {
    MRFP_CHECK(ws && mean && invstd && A && S && C > 0 && count > 0, "bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + mrfp::kFC - 1) / mrfp::kFC)), dim3(mrfp::kFC, mrfp::kFL), 0, (hipStream_t)stream, ws,
                       B * nslab, (double)count, (int)C, weight, bias, eps, momentum, running_mean, running_var, mean,
                       invstd, A, S);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_bn_eval_coef(int64_t C, const float* weight, const float* bias, const float* running_mean,
                      const float* running_var, float eps, float* A, float* S, void* stream) {
    MRFP_CHECK(running_mean && running_var && A && S && C > 0, "bn_eval_coef: bad arguments");
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)C,
                       weight, bias, running_mean, running_var, eps, A, S);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_bn_bwd_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C, const float* weight,
                         const float* mean, const float* invstd, float* dweight, float* dbias, float* P, float* Q,
                         float* R, void* stream) {
    MRFP_CHECK(ws && mean && invstd && P && Q && R && C > 0 && count > 0, "bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((C + mrfp::kFC - 1) / mrfp::kFC)), dim3(mrfp::kFC, mrfp::kFL), 0, (hipStream_t)stream, ws,
                       B * nslab, (double)count, (int)C, weight, mean, invstd, dweight, dbias, P, Q, R);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_in_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C, const float* weight,
                     const float* bias, float eps, float* mean, float* invstd, float* A, float* S, void* stream) {
    MRFP_CHECK(ws && mean && invstd && A && S && C > 0 && count > 0 && B > 0 && B < 65536, "in_finalize: bad arguments");
    hipLaunchKernelGGL(in_finalize_kernel, dim3((unsigned)((C + mrfp::kFC - 1) / mrfp::kFC), (unsigned)B), dim3(mrfp::kFC, mrfp::kFL), 0,
                       (hipStream_t)stream, ws, nslab, (double)count, (int)C, weight, bias, eps, mean, invstd, A, S);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_in_bwd_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C, const float* weight,
                         const float* mean, const float* invstd, float* dweight, float* dbias, float* P, float* Q,
                         float* R, void* stream) {
    MRFP_CHECK(ws && mean && invstd && P && Q && R && C > 0 && count > 0, "in_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3((unsigned)((C + mrfp::kFC - 1) / mrfp::kFC)), dim3(mrfp::kFC, mrfp::kFL), 0, (hipStream_t)stream, ws,
                       (int)B, nslab, (double)count, (int)C, weight, mean, invstd, dweight, dbias, P, Q, R);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_np_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C, const float* alpha,
                     const float* beta_noise, float* mu, float* sigma, float* A, float* S, void* stream) {
    MRFP_CHECK(ws && alpha && beta_noise && mu && sigma && A && S && C > 0 && B > 0 && B < 65536, "np_finalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(plane_sum_kernel, dim3((unsigned)((C + mrfp::kFC - 1) / mrfp::kFC), (unsigned)B), dim3(mrfp::kFC, mrfp::kFL), 0, st, ws, nslab,
                       1.0 / (double)count, (int)C, mu);
    MRFP_LAUNCH_CHECK();
    hipLaunchKernelGGL(np_coef_kernel, dim3(1), dim3(kThreads), 0, st, (int)B, (int)C, alpha, beta_noise, mu, sigma, A, S);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_np_bwd_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C, const float* alpha,
                         const float* beta_noise, const float* mu, const float* sigma, float* Gtmp, float* K,
                         void* stream) {
    MRFP_CHECK(ws && alpha && beta_noise && mu && sigma && Gtmp && K && C > 0 && B > 0 && B < 65536, "np_bwd_finalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(plane_sum_kernel, dim3((unsigned)((C + mrfp::kFC - 1) / mrfp::kFC), (unsigned)B), dim3(mrfp::kFC, mrfp::kFL), 0, st, ws, nslab, 1.0,
                       (int)C, Gtmp);
    MRFP_LAUNCH_CHECK();
    hipLaunchKernelGGL(np_bwd_coef_kernel, dim3(1), dim3(kThreads), 0, st, (int)B, (int)C, (double)count, alpha,
                       beta_noise, mu, sigma, Gtmp, K);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_mean_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C, float* tmp, void* out,
                       int dtype, void* stream) {
    MRFP_CHECK(ws && out && tmp && C > 0 && B > 0 && B < 65536, "mean_finalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* dst = dtype == MRFP_F32 ? (float*)out : tmp;
    hipLaunchKernelGGL(plane_sum_kernel, dim3((unsigned)((C + mrfp::kFC - 1) / mrfp::kFC), (unsigned)B), dim3(mrfp::kFC, mrfp::kFL), 0, st, ws, nslab,
                       1.0 / (double)count, (int)C, dst);
    MRFP_LAUNCH_CHECK();
    if (dtype == MRFP_BF16) {
        const int64_t n = B * C;
        hipLaunchKernelGGL((cast_out_kernel<bf16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tmp, (bf16*)out, n);
        MRFP_LAUNCH_CHECK();
    } else if (dtype == MRFP_F16) {
        const int64_t n = B * C;
        hipLaunchKernelGGL((cast_out_kernel<f16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tmp, (f16*)out, n);
        MRFP_LAUNCH_CHECK();
    } else {
        MRFP_CHECK(dtype == MRFP_F32, "mean_finalize: unknown dtype %d", dtype);
    }
    return 0;
}

}  // extern "C"
