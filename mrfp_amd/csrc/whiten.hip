// whiten.hip -- group whitening passes over an NHWC activation (groups of 16 channels): the heavy parts of the
// switchable / instance whitening options (reference network/sync_switchwhiten.py:20-26, 161-170 "in_data.mean",
// "bmm(x, x^T)" per group; 217 "bmm(wm, in_data)"; network/instance_whitening.py).
//
//   group_moments   M[b,g] = sum_p a_g(p) b_g(p)^T  (16x16 per group, fp32) and sum_p a(p)  -- ONE read of a and b
//                   (a == b: the second moments of the forward pass; a = dy, b = x: the gradient of the whitening matrix)
//   group_apply     y(p) = Wm[b,g] x_g(p) (+ Vm[b,g] z_g(p)) + shift[b]                     -- one read (two), one write
//                   (forward: the folded whitening matrix; backward: dx = Wm^T dy + (dM + dM^T) x + dmu/HW in one pass)
//
// Both are HBM-bound streaming kernels.  A thread owns (pixel slot, group g, row quarter q): it loads the 16 channels
// of its group (two 16-byte loads for the 16-bit types; consecutive threads read consecutive 32/64-byte runs, so a wave
// covers whole pixels) and keeps a 4x16 block of the 16x16 product / matrix in registers: 64 FMAs per 16 channels read.
// Reduction over pixels: registers -> LDS across the pixel slots of the workgroup -> one fp32 partial per workgroup ->
// fp64 combination in a finalize kernel (fixed order: bitwise reproducible).
#include "conv_common.hpp"      // MFMA wrappers (Mma16), fragment types

namespace mrfp {

constexpr int kG = 16;                 // channels per whitening group (reference num_pergroup = 16)
constexpr int kPart = 4 * kG + 4;      // floats a thread contributes: its 4x16 block + its 4 channel sums

template <typename T>
__device__ __forceinline__ void load16(const T* p, float (&o)[16]) {
    constexpr int V = FullVec<T>::value;             // 8 (16-bit) / 4 (fp32) elements per 16-byte load
#pragma unroll
    for (int i = 0; i < 16 / V; ++i) {
        float t[V];
        load_f<T, V>(p + i * V, t);
#pragma unroll
        for (int u = 0; u < V; ++u) o[i * V + u] = t[u];
    }
}

struct WhitenGeom {
    int tpp;     // threads per pixel = C / 4
    int ppi;     // pixel slots per workgroup = 256 / tpp
    int nch;     // pixel chunks (workgroups) per image
    int per;     // pixels per chunk (multiple of ppi)
};
static WhitenGeom whiten_geom(int64_t B, int64_t HW, int64_t C) {
    WhitenGeom g;
    g.tpp = (int)(C / 4);
    g.ppi = kThreads / g.tpp;
    int64_t want = 2048 / (B > 0 ? B : 1);                       // ~2048 workgroups over the chip
    if (want < 1) want = 1;
    int64_t per = (HW + want - 1) / want;
    const int64_t minper = 8 * g.ppi;                             // at least 8 iterations per workgroup
    if (per < minper) per = minper;
    per = (per + g.ppi - 1) / g.ppi * g.ppi;
    g.per = (int)per;
    g.nch = (int)((HW + per - 1) / per);
    return g;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void group_moments_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                                  float* __restrict__ part, int HW, int C, int tpp, int ppi,
                                                                  int per) {
    __shared__ float red[kThreads * 17];
    const int slot = threadIdx.x / tpp, tq = threadIdx.x - slot * tpp;
    const int g = tq >> 2, q = tq & 3;
    const int img = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * per, p1 = min(p0 + per, HW);
    const bool live = slot < ppi;
    float acc[4][16], s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    }
    const size_t base = (size_t)img * HW * C + g * kG;
#pragma unroll 2
    for (int p = live ? p0 + slot : p1; p < p1; p += ppi) {
        float av[4], bv[16];
        load_f<T, 4>(a + base + (size_t)p * C + 4 * q, av);
        load16<T>(b + base + (size_t)p * C, bv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ai = av[i];
            s[i] += ai;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = fmaf(ai, bv[j], acc[i][j]);
        }
    }
    // across the pixel slots of the workgroup, 17 values at a time through LDS
    float* out = part + ((size_t)img * gridDim.x + chunk) * (size_t)tpp * kPart;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        __syncthreads();
        if (live) {
#pragma unroll
            for (int j = 0; j < 16; ++j) red[(slot * tpp + tq) * 17 + j] = acc[r][j];
            red[(slot * tpp + tq) * 17 + 16] = s[r];
        }
        __syncthreads();
        for (int v = threadIdx.x; v < tpp * 17; v += kThreads) {
            const int t = v / 17, e = v - t * 17;
            float sum = 0.f;
            for (int sl = 0; sl < ppi; ++sl) sum += red[(sl * tpp + t) * 17 + e];
            out[(size_t)t * kPart + (e < 16 ? r * 16 + e : 64 + r)] = sum;
        }
    }
}

// ---- the same moments on the matrix cores (16-bit activations; round 5) ---------------------------------
// The per-group second moment  M[b,g] = sum_p a_g(p) b_g(p)^T  (reference sync_switchwhiten.py:165, 206-217: bmm over the pixels of a
// 16-channel group) is a 16 x 16 x (pixels) contraction -- one v_mfma_f32_16x16x32 per group and 32 pixels, the operands taken
// from memory as they are (bf16 / f16 products are exact in fp32; fp32 accumulation as before).  The VALU kernel above spends 64
// FMAs per 16 channels read and is bound by their issue (2.0 TB/s bf16, profiles/r01_whitening.md); here the matrix pipe does that
// work in 16 cycles per KB and the kernel streams.
//   * a workgroup = 4 waves walks tiles of 32 pixels x one 256-channel slab of its pixel chunk; the tile goes global -> registers
//     (16-byte loads, one tile ahead) -> LDS rows of 512 + 32 bytes ([pixel][channel]: the memory layout);
//   * the contraction index (pixel) is the SLOW index of that image, so the MFMA fragments -- lane (channel c = l & 15, pixel
//     block l >> 4) holds 8 pixels of one channel -- come from the transposing LDS read ds_read_b64_tr_b16 (4 pixel rows x 16
//     channels per 16-lane group).  Which pixels a lane group takes is free as long as both operands agree: group q reads rows
//     4q .. 4q+3 and 16+4q .. 16+4q+3, so that the eight rows of a 32-lane half differ mod 8 and the 32-byte row segments fall
//     into disjoint bank windows at the 544-byte pitch (136 dwords = 8 mod 64);
//   * a == b (the forward's covariance): ONE fragment serves as both operands;
//   * the channel sums (sum_p a) are a second MFMA against an all-ones operand (every column of its result is the row sum);
//   * partials leave in the layout of the VALU kernel ([tpp][68] per workgroup), so the fp64 finalize kernel is shared.
typedef __attribute__((ext_vector_type(4))) short short4v_w;
typedef __attribute__((address_space(3))) short4v_w lds_short4v_w;
constexpr int kMmPitch = 544;          // bytes per pixel row of the LDS tile: 256 channels x 2 B + 32
constexpr int kMmSlab = 256;           // channels per workgroup

template <typename T> __device__ __forceinline__ uint4 ones_frag();
template <> __device__ __forceinline__ uint4 ones_frag<bf16>() { return make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u); }
template <> __device__ __forceinline__ uint4 ones_frag<f16>() { return make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u); }

__device__ __forceinline__ uint4 mm_frag(const char* tile, int group_in_slab, int lane) {
    const int q = lane >> 4, r4 = (lane & 15) >> 2, c4 = lane & 3;
    const char* p0 = tile + (4 * q + r4) * kMmPitch + group_in_slab * 32 + c4 * 8;
    const short4v_w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v_w*)(p0));
    const short4v_w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v_w*)(p0 + 16 * kMmPitch));
    uint4 r;
    r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
    r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
    r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
    r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
    return r;
}

template <typename T, bool SAME>
__global__ __launch_bounds__(kThreads, (SAME ? 4 : 3)) void group_moments_mfma_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                                       float* __restrict__ part, int HW, int C, int tpp, int per) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ta = smem;
    char* const tb = smem + 32 * kMmPitch;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int img = blockIdx.y, chunk = blockIdx.x, slab = blockIdx.z;
    const int c0 = slab * kMmSlab, cs = min(C - c0, kMmSlab);       // channels of this slab
    const int ng = cs >> 4;                                          // its groups (<= 16)
    const int cpp = cs >> 3;                                         // 16-byte chunks per pixel (<= 32)
    const int nck = 32 * cpp;                                        // chunks per tile (<= 1024)
    const int p0 = chunk * per, p1 = min(p0 + per, HW);
    const int ntile = (p1 - p0 + 31) >> 5;
    const size_t base = (size_t)img * HW * C + c0;
    uint4 ra[2][4], rb[2][SAME ? 1 : 4];      // two tiles ahead in registers (two named sets: compile-time indices only)
    int lds_o[4];
    int px[4], cc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = t + i * kThreads;
        px[i] = ci < nck ? ci / cpp : -1;
        cc[i] = ci < nck ? ci - px[i] * cpp : 0;
        lds_o[i] = px[i] >= 0 ? px[i] * kMmPitch + cc[i] * 16 : 0;
    }
    auto load = [&](int tile, uint4 (&qa)[4], uint4 (&qb)[SAME ? 1 : 4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + tile * 32 + px[i];
            const bool ok = px[i] >= 0 && p < p1;
            const size_t off = base + (size_t)(ok ? p : p0) * C + cc[i] * 8;
            const uint4 va = *reinterpret_cast<const uint4*>(a + off);
            qa[i] = ok ? va : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (!SAME) {
                const uint4 vb = *reinterpret_cast<const uint4*>(b + off);
                qb[i] = ok ? vb : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    f32x4 accM[4], accS[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { accM[k][e] = 0.f; accS[k][e] = 0.f; }
    const uint4 ones = ones_frag<T>();
    auto step = [&](int tile, uint4 (&qa)[4], uint4 (&qb)[SAME ? 1 : 4]) {
        __syncthreads();                      // every wave has read the previous tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (px[i] >= 0) {
                *reinterpret_cast<uint4*>(ta + lds_o[i]) = qa[i];
                if constexpr (!SAME) *reinterpret_cast<uint4*>(tb + lds_o[i]) = qb[i];
            }
        if (tile + 2 < ntile) load(tile + 2, qa, qb);     // two tiles in flight behind this tile's multiplies
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = wave + 4 * k;           // (wave-uniform; the transposing read needs every lane's address: no early exit)
            const int gs = g < ng ? g : 0;
            const uint4 fa = mm_frag(ta, gs, lane);
            uint4 fb = fa;
            if constexpr (!SAME) fb = mm_frag(tb, gs, lane);
            Mma16<T>::run(accM[k], fa, fb);
            Mma16<T>::run(accS[k], fa, ones);
        }
    };
    if (ntile > 0) load(0, ra[0], rb[0]);
    if (ntile > 1) load(1, ra[1], rb[1]);
    for (int tile = 0; tile < ntile; tile += 2) {
        step(tile, ra[0], rb[0]);
        if (tile + 1 < ntile) step(tile + 1, ra[1], rb[1]);
    }
    // D[row = 4 * (lane >> 4) + e][col = lane & 15]: lane (q, j) holds rows 4q .. 4q+3 of column j -- the [4 x 16] block + 4 sums of
    // thread (g, q) of the VALU kernel
    float* out = part + ((size_t)img * gridDim.x + chunk) * (size_t)tpp * kPart;
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g = wave + 4 * k;
        if (g < ng) {
            float* o = out + (size_t)(((c0 >> 4) + g) * 4 + q) * kPart;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e * 16 + j] = accM[k][e];
            if (j == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[64 + e] = accS[k][e];
            }
        }
    }
}

// part [B][nch][tpp][68] -> M [B][C/16][16][16], sum_a [B][C]   (fp64 combination, fixed order)
__global__ __launch_bounds__(kThreads) void group_moments_finalize_kernel(const float* __restrict__ part, int nch, int tpp,
                                                                           int C, float* __restrict__ M, float* __restrict__ sum_a) {
    const int img = blockIdx.y;
    const int v = blockIdx.x * kThreads + threadIdx.x;
    if (v >= tpp * kPart) return;
    const int t = v / kPart, e = v - t * kPart;
    double acc = 0.0;
    const float* src = part + (size_t)img * nch * tpp * kPart + v;
    for (int c = 0; c < nch; ++c) acc += (double)src[(size_t)c * tpp * kPart];
    const int g = t >> 2, q = t & 3;
    if (e < 64) {
        const int i = e >> 4, j = e & 15;
        M[(((size_t)img * (C / kG) + g) * kG + 4 * q + i) * kG + j] = (float)acc;
    } else if (sum_a) {
        sum_a[(size_t)img * C + g * kG + 4 * q + (e - 64)] = (float)acc;
    }
}

template <typename T, bool TWO>
__global__ __launch_bounds__(kThreads) void group_apply_kernel(const T* __restrict__ x, const float* __restrict__ Wm,
                                                                const T* __restrict__ z, const float* __restrict__ Vm,
                                                                const float* __restrict__ shift, T* __restrict__ y, int HW,
                                                                int C, int tpp, int ppi, int per) {
    const int slot = threadIdx.x / tpp, tq = threadIdx.x - slot * tpp;
    const int g = tq >> 2, q = tq & 3;
    const int img = blockIdx.y;
    const int p0 = blockIdx.x * per, p1 = min(p0 + per, HW);
    if (slot >= ppi) return;
    float w[4][16], v[TWO ? 4 : 1][16], sh[4];
    const size_t mb = (((size_t)img * (C / kG) + g) * kG + 4 * q) * kG;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sh[i] = shift ? shift[(size_t)img * C + g * kG + 4 * q + i] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            w[i][j] = Wm[mb + i * kG + j];
            if (TWO) v[i][j] = Vm[mb + i * kG + j];
        }
    }
    const size_t base = (size_t)img * HW * C + g * kG;
    for (int p = p0 + slot; p < p1; p += ppi) {
        float xv[16], zv[16], o[4];
        load16<T>(x + base + (size_t)p * C, xv);
        if (TWO) load16<T>(z + base + (size_t)p * C, zv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r = sh[i];
#pragma unroll
            for (int j = 0; j < 16; ++j) r = fmaf(w[i][j], xv[j], r);
            if (TWO) {
#pragma unroll
                for (int j = 0; j < 16; ++j) r = fmaf(v[i][j], zv[j], r);
            }
            o[i] = r;
        }
        store_f<T, 4>(y + base + (size_t)p * C + 4 * q, o);
    }
}

// ---- inverse square root of the 16x16 group covariances: Newton-Schulz (reference sync_switchwhiten.py:206-215) ----
//   r = 1/tr(S); Sn = S r; P_0 = I; P_{k+1} = 1.5 P_k - 0.5 P_k^3 Sn; wm = P_T sqrt(r)
// One workgroup of 256 threads per matrix, thread (i,j) owns one element, operands in LDS (16 FMAs per product and
// thread).  The backward kernel recomputes the P_k chain (T <= 8 copies in LDS) and walks it in reverse:
//   dQ = -0.5 dP_{k+1};  dP_k = 1.5 dP_{k+1} + dQ (P^2 Sn)^T + P^T dQ (P Sn)^T + (P^2)^T dQ Sn^T;  dSn += (P^3)^T dQ
//   dS = dSn r + dtr I,  dtr = -r^2 (sum(dwm . P_T) / (2 sqrt r) + sum(dSn . S))
constexpr int kNsMaxT = 8;

__device__ __forceinline__ float mm16(const float* A, const float* B, int i, int j) {        // (A B)[i][j]
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r = fmaf(A[i * 17 + k], B[k * 17 + j], r);
    return r;
}
__device__ __forceinline__ float mm16_tn(const float* A, const float* B, int i, int j) {     // (A^T B)[i][j]
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r = fmaf(A[k * 17 + i], B[k * 17 + j], r);
    return r;
}
__device__ __forceinline__ float mm16_nt(const float* A, const float* B, int i, int j) {     // (A B^T)[i][j]
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r = fmaf(A[i * 17 + k], B[j * 17 + k], r);
    return r;
}
__device__ __forceinline__ float block_sum256(float v, float* red) {     // sum over the 256 threads; red: 4 floats
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void group_isqrt_fwd_kernel(const float* __restrict__ cov, float* __restrict__ wm, int T) {
    __shared__ float Sn[16 * 17], P[16 * 17], A[16 * 17], Bm[16 * 17], red[4];
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15, e = i * 17 + j;
    const size_t off = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float s = cov[off];
    const float r = 1.f / block_sum256(i == j ? s : 0.f, red);
    Sn[e] = s * r;
    float p = i == j ? 1.f : 0.f;
    P[e] = p;
    __syncthreads();
    for (int k = 0; k < T; ++k) {
        A[e] = mm16(P, P, i, j);            // P^2
        __syncthreads();
        Bm[e] = mm16(A, P, i, j);           // P^3
        __syncthreads();
        p = 1.5f * p - 0.5f * mm16(Bm, Sn, i, j);
        __syncthreads();
        P[e] = p;
        __syncthreads();
    }
    wm[off] = p * sqrtf(r);
}

__global__ __launch_bounds__(256) void group_isqrt_bwd_kernel(const float* __restrict__ cov, const float* __restrict__ dwm,
                                                               float* __restrict__ dcov, int T) {
    __shared__ float Sn[16 * 17], Pk[kNsMaxT + 1][16 * 17], A[16 * 17], Bm[16 * 17], Cm[16 * 17], G[16 * 17], dQ[16 * 17], red[4];
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15, e = i * 17 + j;
    const size_t off = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float s = cov[off];
    const float r = 1.f / block_sum256(i == j ? s : 0.f, red);
    Sn[e] = s * r;
    float p = i == j ? 1.f : 0.f;
    Pk[0][e] = p;
    __syncthreads();
    for (int k = 0; k < T; ++k) {
        A[e] = mm16(Pk[k], Pk[k], i, j);
        __syncthreads();
        Bm[e] = mm16(A, Pk[k], i, j);
        __syncthreads();
        p = 1.5f * p - 0.5f * mm16(Bm, Sn, i, j);
        Pk[k + 1][e] = p;
        __syncthreads();
    }
    const float sr = sqrtf(r);
    const float g0 = dwm[off];
    const float dr_a = block_sum256(g0 * p, red) * 0.5f / sr;      // wm = P_T sqrt(r)
    float g = g0 * sr;                                             // dP_T
    float dsn = 0.f;
    for (int k = T - 1; k >= 0; --k) {
        const float* Pc = Pk[k];
        __syncthreads();
        dQ[e] = -0.5f * g;
        A[e] = mm16(Pc, Pc, i, j);            // P^2
        Cm[e] = mm16(Pc, Sn, i, j);           // P Sn
        __syncthreads();
        Bm[e] = mm16(A, Sn, i, j);            // P^2 Sn
        __syncthreads();
        const float t1 = mm16_nt(dQ, Bm, i, j);                     // dQ (P^2 Sn)^T
        const float u = mm16_tn(Pc, dQ, i, j);                      // P^T dQ
        const float w = mm16_tn(A, dQ, i, j);                       // (P^2)^T dQ
        __syncthreads();
        Bm[e] = mm16(A, Pc, i, j);            // P^3
        G[e] = u;
        __syncthreads();
        const float t2 = mm16_nt(G, Cm, i, j);                      // (P^T dQ) (P Sn)^T
        dsn += mm16_tn(Bm, dQ, i, j);                               // (P^3)^T dQ
        __syncthreads();
        G[e] = w;
        __syncthreads();
        const float t3 = mm16_nt(G, Sn, i, j);                      // ((P^2)^T dQ) Sn^T
        g = 1.5f * g + t1 + t2 + t3;
    }
    const float dr = dr_a + block_sum256(dsn * s, red);
    const float dtr = -dr * r * r;
    dcov[off] = dsn * r + (i == j ? dtr : 0.f);
}

static bool whiten_shape_ok(int64_t B, int64_t HW, int64_t C) {
    return B > 0 && B < 65536 && HW > 0 && HW < (1ll << 31) && C >= kG && C <= 1024 && C % kG == 0;
}

template <typename T>
static int run_moments(const void* a, const void* b, float* M, float* sum_a, float* ws, int64_t B, int64_t HW, int64_t C,
                       hipStream_t st) {
    const WhitenGeom g = whiten_geom(B, HW, C);
    static int mfma = -1;              // MRFP_WHITEN_MFMA=0: the VALU kernel for 16-bit activations too (A/B runs)
    if (mfma < 0) { const char* e = getenv("MRFP_WHITEN_MFMA"); mfma = e ? atoi(e) : 1; }
    if constexpr (sizeof(T) == 2) {
        if (mfma) {
            // HALF a round of resident workgroups (2 of the 4 slots per CU x 256 CUs; MRFP_WHITEN_WGS) instead of the VALU kernel's ~2 048
            // short ones: pixel chunks of whole 32-pixel tiles, never more chunks than the workspace was sized for (g.nch).  Swept at
            // 16 x 256 x 192^2 (profiles/r05_whitening.md): 256 / 384 / 512 / 640 / 768 / 1024 / 2048 workgroups -> (x,x) 3.9 / 4.7 / 5.1 / 4.7 / 4.5 / 4.2 / 3.2 TB/s
            static int wgs = -1;
            if (wgs < 0) { const char* e = getenv("MRFP_WHITEN_WGS"); wgs = e ? atoi(e) : 2 * kCUs; if (wgs < 64) wgs = 2 * kCUs; }
            const int slabs = (int)((C + kMmSlab - 1) / kMmSlab);
            int64_t want = (a == b ? wgs : wgs * 3 / 4) / (B * slabs);      // (a != b: two register sets of two operands, 3 per CU)
            if (want < 1) want = 1;
            if (want > g.nch) want = g.nch;
            int64_t per = (HW + want - 1) / want;
            per = (per + 31) / 32 * 32;
            const int nch = (int)((HW + per - 1) / per);
            const dim3 grid((unsigned)nch, (unsigned)B, (unsigned)slabs);
            if (a == b)
                hipLaunchKernelGGL((group_moments_mfma_kernel<T, true>), grid, dim3(kThreads), 32 * kMmPitch, st, (const T*)a, (const T*)b,
                                   ws, (int)HW, (int)C, g.tpp, (int)per);
            else
                hipLaunchKernelGGL((group_moments_mfma_kernel<T, false>), grid, dim3(kThreads), 2 * 32 * kMmPitch, st, (const T*)a,
                                   (const T*)b, ws, (int)HW, (int)C, g.tpp, (int)per);
            MRFP_LAUNCH_CHECK();
            const int nv = g.tpp * kPart;
            hipLaunchKernelGGL(group_moments_finalize_kernel, dim3((unsigned)((nv + kThreads - 1) / kThreads), (unsigned)B),
                               dim3(kThreads), 0, st, (const float*)ws, nch, g.tpp, (int)C, M, sum_a);
            MRFP_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL((group_moments_kernel<T>), dim3((unsigned)g.nch, (unsigned)B), dim3(kThreads), 0, st, (const T*)a,
                       (const T*)b, ws, (int)HW, (int)C, g.tpp, g.ppi, g.per);
    MRFP_LAUNCH_CHECK();
    const int nv = g.tpp * kPart;
    hipLaunchKernelGGL(group_moments_finalize_kernel, dim3((unsigned)((nv + kThreads - 1) / kThreads), (unsigned)B),
                       dim3(kThreads), 0, st, (const float*)ws, g.nch, g.tpp, (int)C, M, sum_a);
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int run_apply(const void* x, const float* Wm, const void* z, const float* Vm, const float* shift, void* y, int64_t B,
                     int64_t HW, int64_t C, hipStream_t st) {
    const WhitenGeom g = whiten_geom(B, HW, C);
    const dim3 grid((unsigned)g.nch, (unsigned)B);
    if (z)
        hipLaunchKernelGGL((group_apply_kernel<T, true>), grid, dim3(kThreads), 0, st, (const T*)x, Wm, (const T*)z, Vm, shift,
                           (T*)y, (int)HW, (int)C, g.tpp, g.ppi, g.per);
    else
        hipLaunchKernelGGL((group_apply_kernel<T, false>), grid, dim3(kThreads), 0, st, (const T*)x, Wm, (const T*)nullptr,
                           (const float*)nullptr, shift, (T*)y, (int)HW, (int)C, g.tpp, g.ppi, g.per);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int64_t mrfp_group_moments_ws_bytes(int64_t B, int64_t HW, int64_t C) {
    if (!whiten_shape_ok(B, HW, C)) return 0;
    const WhitenGeom g = whiten_geom(B, HW, C);
    return B * (int64_t)g.nch * g.tpp * kPart * 4;
}

int mrfp_group_moments(const void* a, const void* b, float* M, float* sum_a, void* ws, int dtype, int64_t B, int64_t HW,
                       int64_t C, void* stream) {
    MRFP_CHECK(a && b && M && ws, "group_moments: null argument");
    MRFP_CHECK(whiten_shape_ok(B, HW, C), "group_moments: unsupported shape B=%lld HW=%lld C=%lld (C %% 16 == 0, C <= 1024)",
               (long long)B, (long long)HW, (long long)C);
    MRFP_CHECK(aligned16(a) && aligned16(b), "group_moments: activations must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return run_moments<float>(a, b, M, sum_a, (float*)ws, B, HW, C, st);
    if (dtype == MRFP_BF16) return run_moments<bf16>(a, b, M, sum_a, (float*)ws, B, HW, C, st);
    if (dtype == MRFP_F16) return run_moments<f16>(a, b, M, sum_a, (float*)ws, B, HW, C, st);
    MRFP_CHECK(false, "group_moments: unknown dtype %d", dtype);
}

int mrfp_group_apply(const void* x, const float* Wm, const void* z, const float* Vm, const float* shift, void* y, int dtype,
                     int64_t B, int64_t HW, int64_t C, void* stream) {
    MRFP_CHECK(x && Wm && y && (!z == !Vm), "group_apply: null argument (z and Vm come together)");
    MRFP_CHECK(whiten_shape_ok(B, HW, C), "group_apply: unsupported shape B=%lld HW=%lld C=%lld (C %% 16 == 0, C <= 1024)",
               (long long)B, (long long)HW, (long long)C);
    MRFP_CHECK(aligned16(x) && aligned16(y) && (!z || aligned16(z)), "group_apply: activations must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MRFP_F32) return run_apply<float>(x, Wm, z, Vm, shift, y, B, HW, C, st);
    if (dtype == MRFP_BF16) return run_apply<bf16>(x, Wm, z, Vm, shift, y, B, HW, C, st);
    if (dtype == MRFP_F16) return run_apply<f16>(x, Wm, z, Vm, shift, y, B, HW, C, st);
    MRFP_CHECK(false, "group_apply: unknown dtype %d", dtype);
}

int mrfp_group_isqrt_fwd(const float* cov, float* wm, int64_t n, int T, void* stream) {
    MRFP_CHECK(cov && wm && n > 0 && n < (1ll << 31) && T >= 0 && T <= kNsMaxT, "group_isqrt_fwd: bad arguments (T <= %d)", kNsMaxT);
    hipLaunchKernelGGL(group_isqrt_fwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, cov, wm, T);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_group_isqrt_bwd(const float* cov, const float* dwm, float* dcov, int64_t n, int T, void* stream) {
    MRFP_CHECK(cov && dwm && dcov && n > 0 && n < (1ll << 31) && T >= 0 && T <= kNsMaxT, "group_isqrt_bwd: bad arguments (T <= %d)", kNsMaxT);
    hipLaunchKernelGGL(group_isqrt_bwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, cov, dwm, dcov, T);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
