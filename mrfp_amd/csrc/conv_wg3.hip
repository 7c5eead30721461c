#include "conv_common.hpp"

// =============================================================================================
// Weight gradient of the 3x3 / stride 1 / pad = dilation (1 or 2) convolutions, 16-bit activations, ACCUMULATOR-STATIONARY
// (round 5).    dW[n, (r,s), c] = sum_{b,oh,ow} dY[b,oh,ow,n] * X[b, oh + (r-1)d, ow + (s-1)d, c]
//
// conv_wgrad_kernel (conv_wgrad.hip) treats the nine taps as nine independent column tiles: each re-fetches the dY rows of its
// K' range and its own shifted copy of X (18 column tiles x 28 splits on the decoder layer: HBM reads 2.0 - 2.7x the algorithmic bytes,
// 85 FLOP per byte through the L2 -> LDS fill path).  Here a workgroup owns dW[64 n][9 taps][64 c] -- 36 864 fp32 = 144 accumulator
// registers per lane -- and streams the pixels past it ONCE: the rolling window of image-row strips of conv_c64.hip (one new input row
// per output row by LDS-DMA, every tap a shifted read of the same window rows) plus the dY strip of the row.  Per 64 output pixels
// 8.5 KB of X + 8 KB of dY are filled for 4.7 MFLOP: 287 FLOP per fill byte.  Both operands are "k-strided" (the pixel is the slow
// index), so the MFMA fragments (v_mfma_f32_16x16x32: A = 16 channels of X x 32 pixels, B = 16 channels of dY x 32 pixels) come from
// the transposing LDS read ds_read_b64_tr_b16 over [pixel][64 channels] rows (128 B per pixel, 32-byte blocks XOR-swizzled by
// (pixel >> 1) & 3: the 8 pixels of a 32-lane access fall into 8 disjoint bank windows, for every tap shift).
//
// Work: a "class" = (problem, 64-channel block of N, 64-channel block of C); a "unit" = RU output rows of one SW-pixel strip (RU * SW
// a multiple of 32: one or two rows); units of one (image, strip, dilation class) are consecutive rows, so the window rolls.  The
// workgroups of one problem are dealt so that the classes walk the SAME unit range side by side on one XCD (x is shared by the classes
// of a C block, dY by those of an N block: they meet in L2), every workgroup gets the same number of units, and each writes its
// partial dW to a slab slot that the shared wgrad_reduce_kernel sums in a fixed order (bitwise reproducible).
// =============================================================================================
namespace mrfp {

typedef __attribute__((ext_vector_type(4))) short wg3_short4;
typedef __attribute__((address_space(3))) wg3_short4 wg3_lds_short4;

constexpr int kWg3MaxGroup = 32;
struct Wg3Group {
    const char* x[kWg3MaxGroup];
    const char* dy[kWg3MaxGroup];
};
struct Wg3P {
    float* slab;             // [problem][splits][N][Q] fp32
    int B, H, W, C, N, ldn, d;
    int strips, ups;         // strips per image row, units per segment (= one (image, strip, dilation class): H / (d * RU))
    int U;                   // units per class = B * strips * d * ups
    int ncb, ncls;           // 64-channel blocks of C; classes per problem
    int Wp, a, L;            // workgroups per problem; main chunks per class; units per main chunk
    int R, Wr;               // units per class left to the remainder workgroups; their number per problem
    int splits, Q;
    unsigned xbytes, dybytes;
};

__device__ __forceinline__ uint4 wg3_frag(const char* lo, const char* hi) {
    const wg3_short4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg3_lds_short4*)(lo));
    const wg3_short4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg3_lds_short4*)(hi));
    uint4 r;
    r.x = (unsigned)(unsigned short)a[0] | ((unsigned)(unsigned short)a[1] << 16);
    r.y = (unsigned)(unsigned short)a[2] | ((unsigned)(unsigned short)a[3] << 16);
    r.z = (unsigned)(unsigned short)b[0] | ((unsigned)(unsigned short)b[1] << 16);
    r.w = (unsigned)(unsigned short)b[2] | ((unsigned)(unsigned short)b[3] << 16);
    return r;
}

// SW: strip width in pixels (a multiple of 16); RU: output rows per unit (RU * SW a multiple of 32)
template <typename T, int SW, int RU>
__global__ __launch_bounds__(256, 2) void conv_wg3_kernel(Wg3P p, Wg3Group grp) {
    constexpr int HL = 2;                          // halo pixels either side of a window row (the largest dilation)
    constexpr int NPX = (SW + 2 * HL + 7) / 8;     // 1 KiB pieces (8 pixels x 128 B) per window row
    constexpr int XSLOT = NPX * 1024;
    constexpr int NSLOT = 2 * RU + 2;              // window rows: RU + 2 in use, RU arriving
    constexpr int NPD = RU * SW / 8;               // pieces of one unit's dY strip
    constexpr int DYB = NPD * 1024;
    constexpr int NKS = RU * SW / 32;              // k steps (32 pixels) per unit
    constexpr int HPR = SW / 16;                   // 16-pixel half tiles per row
    static_assert(SW % 16 == 0 && (RU * SW) % 32 == 0, "a k step is two 16-pixel half tiles, each inside one image row");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int d = p.d;

    const int w = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int prob = w / p.Wp, v = w - prob * p.Wp;
    const i32x4 xw = rsrc_words(grp.x[prob], p.xbytes);
    const i32x4 yw = rsrc_words(grp.dy[prob], p.dybytes);

    // fragment offsets of this lane: pixel 4q + r4 (+ 16 for the second read) of a half tile, 8 bytes = 4 channels at c4
    const int lpx = 4 * (lane >> 4) + ((lane & 15) >> 2), c4 = lane & 3;
    unsigned xo[3], yo[4];
#pragma unroll
    for (int s = 0; s < 3; ++s) {                  // X: this wave's 16 channels (32-byte block `wave`), tap column s
        const int px = lpx + HL + (s - 1) * d;
        xo[s] = (unsigned)(px * 128 + (((wave ^ (px >> 1)) & 3) << 5) + c4 * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) yo[j] = (unsigned)(lpx * 128 + (((j ^ (lpx >> 1)) & 3) << 5) + c4 * 8);
    // transfer source of this lane inside a piece: pixel lane >> 3 of the piece, the 16-byte chunk whose 32-byte block is swizzled
    const int ppx = lane >> 3, pch = lane & 7;

    f32x4 acc[9][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[tp][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // ---- one class piece: units [ua, ub) of class (nb, cb) ----
    auto run_piece = [&](int nb, int cb, int ua, int ub) {
        int u = ua;
        while (u < ub) {
            const int seg = u / p.ups, k0 = u - seg * p.ups;
            int k1 = k0 + (ub - u);
            if (k1 > p.ups) k1 = p.ups;
            const int q = seg % d, sb = seg / d;
            const int strip = sb % p.strips, b = sb / p.strips;
            const int Hc = p.H / d;                                    // rows of a dilation class (H % d == 0)
            const unsigned ximg = (unsigned)b * (unsigned)p.H * (unsigned)p.W;
            const int col0 = strip * SW;
            // one window row: class row xr -> image row q + d * xr, pixels col0 - HL ..., into slot `slot`
            auto issue_x = [&](int xr, int slot, int pi) {
                const int ih = q + d * xr;
                const bool rok = xr >= 0 && xr < Hc;
                const int px = 8 * pi + ppx, iw = col0 - HL + px;
                const bool ok = rok && iw >= 0 && iw < p.W;
                const unsigned chunk = (unsigned)(((((pch >> 1) ^ (px >> 1)) & 3) << 1) | (pch & 1));
                const unsigned src = ((ximg + (unsigned)(rok ? ih : 0) * (unsigned)p.W + (unsigned)(ok ? iw : 0)) * (unsigned)p.C + (unsigned)(cb * 64)) * 2u + chunk * 16u;
                dma16_async(xw, lds0 + (unsigned)(slot * XSLOT + pi * 1024), ok ? src : kOOB);
            };
            auto issue_dy = [&](int k, int stage, int pi) {           // unit k of the segment: class rows k * RU ...
                const int pxd = 8 * pi + ppx;
                const int j = pxd / SW, col = pxd - j * SW;
                const int oh = q + d * (k * RU + j);
                const unsigned m = ximg + (unsigned)oh * (unsigned)p.W + (unsigned)(col0 + col);
                const unsigned chunk = (unsigned)(((((pch >> 1) ^ (pxd >> 1)) & 3) << 1) | (pch & 1));
                dma16_async(yw, lds0 + (unsigned)(NSLOT * XSLOT + stage * DYB + pi * 1024), (m * (unsigned)p.ldn + (unsigned)(nb * 64)) * 2u + chunk * 16u);
            };
            // the pieces of `nrows` window rows starting at class row xr0 / slot s, then (k >= 0) the dY strip of unit k: piece i of the list
            // goes to wave i % 4 (wave-uniform: the transfer's LDS address is a scalar)
            auto issue = [&](int xr0, int slot0, int nrows, int k, int stage) {
                const int nx = nrows * NPX, tot = nx + (k >= 0 ? NPD : 0);
                for (int i = wave; i < tot; i += 4) {
                    if (i < nx) {
                        const int rr = i / NPX, pi = i - rr * NPX;
                        int sl = slot0 + rr;
                        if (sl >= NSLOT) sl -= NSLOT;
                        issue_x(xr0 + rr, sl, pi);
                    } else {
                        issue_dy(k, stage, i - nx);
                    }
                }
            };
            const int ci0 = k0 * RU;
            issue(ci0 - 1, 0, RU + 2, k0, 0);        // warm-up: the window of the first unit and its dY strip
            int s0 = 0;                              // slot of class row (first row of the unit) - 1
            for (int k = k0; k < k1; ++k) {
                dma_wait<0>();
                __builtin_amdgcn_s_barrier();        // the unit has landed everywhere; every wave is done with the previous unit
                if (k + 1 < k1) {
                    int sn = s0 + RU + 2;
                    if (sn >= NSLOT) sn -= NSLOT;
                    issue(k * RU + RU + 1, sn, RU, k + 1, (k + 1 - k0) & 1);
                }
                unsigned sl[RU + 2];                 // byte offsets of the window rows (first row - 1) ... (last row + 1)
#pragma unroll
                for (int i = 0; i < RU + 2; ++i) {
                    int s = s0 + i;
                    if (s >= NSLOT) s -= NSLOT;
                    sl[i] = (unsigned)(s * XSLOT);
                }
                const char* dyb = smem + NSLOT * XSLOT + ((k - k0) & 1) * DYB;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int h0 = 2 * ks, h1 = 2 * ks + 1;
                    const int j0 = h0 / HPR, c0 = (h0 % HPR) * 16, j1 = h1 / HPR, c1 = (h1 % HPR) * 16;
                    uint4 fy[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) fy[j] = wg3_frag(dyb + h0 * 2048 + yo[j], dyb + h1 * 2048 + yo[j]);
                    uint4 fx[2][3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) fx[0][s] = wg3_frag(smem + sl[j0] + c0 * 128 + xo[s], smem + sl[j1] + c1 * 128 + xo[s]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (r + 1 < 3) {
#pragma unroll
                            for (int s = 0; s < 3; ++s)
                                fx[(r + 1) & 1][s] = wg3_frag(smem + sl[j0 + r + 1] + c0 * 128 + xo[s], smem + sl[j1 + r + 1] + c1 * 128 + xo[s]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < 3; ++s)
#pragma unroll
                            for (int j = 0; j < 4; ++j) Mma16<T>::run(acc[r * 3 + s][j], fx[r & 1][s], fy[j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                s0 += RU;
                if (s0 >= NSLOT) s0 -= NSLOT;
            }
            u += k1 - k0;
            // the next segment refills the window from slot 0: every wave must be done reading this one first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    };
    // ---- partial dW of a class -> slab slot; lane holds c = 4 * (lane >> 4) .. + 3 (rows of D), n = lane & 15 (column) ----
    auto flush = [&](int nb, int cb, int slot, bool zeros) {
        float* out = p.slab + ((size_t)prob * p.splits + slot) * (size_t)p.N * p.Q;
        const int cc = cb * 64 + wave * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nb * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!zeros) o = make_float4(acc[tp][j][0], acc[tp][j][1], acc[tp][j][2], acc[tp][j][3]);
                *reinterpret_cast<float4*>(out + (size_t)n * p.Q + tp * p.C + cc) = o;
            }
        }
    };

    if (v < p.a * p.ncls) {
        // main chunk i of class cls: the p.ncls workgroups of one chunk are consecutive (one XCD) and walk the same units
        const int i = v / p.ncls, cls = v - i * p.ncls;
        const int nb = cls / p.ncb, cb = cls - nb * p.ncb;
        int ua = i * p.L, ub = ua + p.L;
        const int lim = p.U - p.R;
        if (ua > lim) ua = lim;
        if (ub > lim) ub = lim;
        run_piece(nb, cb, ua, ub);
        flush(nb, cb, i, false);
    } else if (p.R > 0) {
        // remainder: a contiguous range of the class-major list of the units the main chunks leave ([U - R, U) of every class).
        // A range is longer than R, so a class is touched by at most two workgroups: the one that enters it at its first unit
        // writes slot a, one that enters it later slot a + 1; whoever finishes the class zero-fills what is left of the two.
        const int r = v - p.a * p.ncls;
        const long long TR = (long long)p.ncls * p.R;
        long long g0 = (long long)r * TR / p.Wr;
        const long long g1 = (long long)(r + 1) * TR / p.Wr;
        while (g0 < g1) {
            const int cls = (int)(g0 / p.R), off = (int)(g0 - (long long)cls * p.R);
            int len = p.R - off;
            if ((long long)len > g1 - g0) len = (int)(g1 - g0);
            const int nb = cls / p.ncb, cb = cls - nb * p.ncb;
            run_piece(nb, cb, p.U - p.R + off, p.U - p.R + off + len);
            const int ord = off == 0 ? 0 : 1;
            flush(nb, cb, p.a + ord, false);
            if (off + len == p.R && ord == 0) flush(nb, cb, p.a + 1, true);
            zero_acc();
            g0 += len;
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int g_wg3 = -1;
static int wg3_mode() {
    if (g_wg3 < 0) {
        const char* e = getenv("MRFP_WGRAD3");      // 0: never; 1 (default): where the rule below says; 2: wherever it is legal (tests, A/B runs)
        g_wg3 = e ? atoi(e) : 1;
    }
    return g_wg3;
}
// strip width / rows per unit for image width W: 64 x 1, 96 x 1 or 48 x 2 (0: none)
static int wg3_strip(int64_t W, int& ru) {
    ru = 1;
    if (W % 64 == 0) return 64;
    if (W % 96 == 0) return 96;
    if (W % 48 == 0) { ru = 2; return 48; }
    return 0;
}
struct Wg3Plan {
    int sw, ru, ncb, ncls, Wp, a, L, R, Wr, splits, ups, strips, U;
};
static bool wg3_plan(int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t dil, int64_t count, Wg3Plan& pl) {
    pl.sw = wg3_strip(W, pl.ru);
    if (!pl.sw || C % 64 || N % 64 || dil < 1 || dil > 2 || H % (dil * pl.ru)) return false;
    pl.ncb = (int)(C / 64);
    pl.ncls = (int)(N / 64) * pl.ncb;
    pl.strips = (int)(W / pl.sw);
    pl.ups = (int)(H / (dil * pl.ru));
    const int64_t U = B * pl.strips * dil * pl.ups;
    if (U >= (1LL << 30)) return false;
    pl.U = (int)U;
    pl.Wp = (int)(kGrid2PerCU / count);          // two workgroups per CU, one round
    if ((int64_t)pl.ncls * U < pl.Wp) pl.Wp = (int)((int64_t)pl.ncls * U);
    pl.a = pl.Wp / pl.ncls;
    if (pl.a < 1) return false;                  // fewer workgroups than classes: no side-by-side walk
    pl.L = (int)(((int64_t)pl.ncls * U + pl.Wp - 1) / pl.Wp);
    const int64_t main = (int64_t)pl.a * pl.L < U ? (int64_t)pl.a * pl.L : U;
    pl.R = (int)(U - main);
    pl.Wr = pl.Wp - pl.a * pl.ncls;
    if (pl.R > 0 && pl.Wr == 0) return false;    // (cannot happen: Wr == 0 means a * ncls == Wp, so a * L >= U)
    pl.splits = pl.a + (pl.R > 0 ? 2 : 0);
    return true;
}
// upper bound of the slab slots per problem such a launch may use, from what mrfp_conv_wgrad*_ws_bytes knows (N, Q, count)
int64_t wg3_splits_bound(int64_t N, int64_t Q, int64_t count) {
    if (wg3_mode() == 0 || Q % (9 * 64) || N % 64 || count < 1) return 0;
    const int64_t ncls = (N / 64) * (Q / (9 * 64));
    const int64_t a = (kGrid2PerCU / count) / ncls;
    return a < 1 ? 0 : a + 2;
}
bool wg3_applicable(int dtype_size, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t ldn, int64_t R, int64_t S, int64_t Ho,
                    int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t count) {
    if (wg3_mode() == 0 || dtype_size != 2) return false;
    if (R != 3 || S != 3 || stride != 1 || Ho != H || Wo != W || pad_h != dil || pad_w != dil || (ldn & 7)) return false;
    Wg3Plan pl;
    if (!wg3_plan(B, H, W, C, N, dil, count, pl)) return false;
    if (B * H * W * C * 2 >= (int64_t)kOOB || B * H * W * ldn * 2 >= (int64_t)kOOB) return false;
    if (wg3_mode() >= 2) return true;
    // every workgroup first fills a window (RU + 2 rows) and ends with 144 KB of slab stores: it needs a K' loop behind them
    return pl.L >= 24;
}

template <typename T, int SW, int RU>
static int wg3_launch(const Wg3P& q, const Wg3Group& g, int grid, hipStream_t st) {
    constexpr int NPX = (SW + 4 + 7) / 8;
    const int lds = (2 * RU + 2) * NPX * 1024 + 2 * (RU * SW / 8) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wg3_kernel<T, SW, RU>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_wg3_kernel<T, SW, RU>), dim3((unsigned)grid), dim3(256), lds, st, q, g);
    MRFP_LAUNCH_CHECK();
    return 0;
}
template <typename T>
static int wg3_pick(const Wg3Plan& pl, const Wg3P& q, const Wg3Group& g, int grid, hipStream_t st) {
    if (pl.sw == 64) return wg3_launch<T, 64, 1>(q, g, grid, st);
    if (pl.sw == 96) return wg3_launch<T, 96, 1>(q, g, grid, st);
    return wg3_launch<T, 48, 2>(q, g, grid, st);
}

// the caller (wgrad_run, conv_wgrad.hip) has checked wg3_applicable(); returns the slab slots per problem in *splits
int wg3_run(const void* const* xs, const void* const* dys, int64_t count, float* slab, bool is_f16, int64_t B, int64_t H, int64_t W, int64_t C,
            int64_t N, int64_t ldn, int64_t dil, unsigned xbytes, unsigned dybytes, int* splits, hipStream_t st) {
    Wg3Plan pl;
    if (!wg3_plan(B, H, W, C, N, dil, count, pl) || count > kWg3MaxGroup) return -1;
    // the slab the caller sized through wg3_splits_bound() holds that many slots per problem
    MRFP_CHECK(pl.splits <= wg3_splits_bound(N, 9 * C, count) && (int64_t)count * pl.Wp <= kGrid2PerCU,
               "conv_wg3: %d slab slots per problem / %lld workgroups exceed the workspace rule (%lld)", pl.splits,
               (long long)(count * pl.Wp), (long long)wg3_splits_bound(N, 9 * C, count));
    Wg3P q;
    q.slab = slab;
    q.B = (int)B; q.H = (int)H; q.W = (int)W; q.C = (int)C; q.N = (int)N; q.ldn = (int)ldn; q.d = (int)dil;
    q.strips = pl.strips; q.ups = pl.ups; q.U = pl.U; q.ncb = pl.ncb; q.ncls = pl.ncls;
    q.Wp = pl.Wp; q.a = pl.a; q.L = pl.L; q.R = pl.R; q.Wr = pl.Wr; q.splits = pl.splits; q.Q = (int)(9 * C);
    q.xbytes = xbytes; q.dybytes = dybytes;
    Wg3Group g;
    for (int i = 0; i < kWg3MaxGroup; ++i) {
        g.x[i] = (const char*)xs[i < count ? i : 0];
        g.dy[i] = (const char*)dys[i < count ? i : 0];
    }
    *splits = pl.splits;
    const int grid = (int)(count * pl.Wp);
    return is_f16 ? wg3_pick<f16>(pl, q, g, grid, st) : wg3_pick<bf16>(pl, q, g, grid, st);
}

}  // namespace mrfp
