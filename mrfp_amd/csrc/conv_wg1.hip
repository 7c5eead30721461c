#include "conv_common.hpp"

// =============================================================================================
// Weight gradient of the POINTWISE convolutions with C % 256 == 0 and N % 256 == 0, 16-bit activations (round 5):
//     dW[n, c] = sum_m dY[m, n] * X[m, c]
// the grouped layer-3 pair (23 x 256 -> 1024, 22 x 1024 -> 256), 512 <-> 2048, 1024 -> 2048, the ASPP / decoder projections.
// conv_wgrad_kernel's 256 x 128 tile re-fetches X four times and dY twice per problem (2.4x the algorithmic bytes through the L2 -> LDS
// fill path, 9 of its ~13 TB/s) behind a single LDS buffer, and streams HBM at 3.7 TB/s.  Here a 512-thread workgroup owns
// dW[256 n][256 c] (128 accumulator registers per lane: 1.6x fill bytes) and the pixels stream past it through a ring of four 32 KB stages
// (32 pixels x (256 c + 256 n)) with counted waits -- the pipeline of conv_pwk.hip, the operand layout, transposing fragment reads, work
// division (classes walked side by side, equal units per workgroup, slab slots summed by wgrad_reduce_kernel) of conv_wg3.hip.
// =============================================================================================
namespace mrfp {

typedef __attribute__((ext_vector_type(4))) short wg1_short4;
typedef __attribute__((address_space(3))) wg1_short4 wg1_lds_short4;

constexpr int kWg1MaxGroup = 32;
struct Wg1Group {
    const char* x[kWg1MaxGroup];
    const char* dy[kWg1MaxGroup];
};
struct Wg1P {
    float* slab;             // [problem][splits][N][C] fp32
    int C, N, ldn;
    int U;                   // units (32 pixels) per class = M / 32
    int ncb, ncls;           // 256-channel blocks of C; classes per problem
    int Wp, a, L;            // workgroups per problem; main chunks per class; units per main chunk
    int R, Wr;               // units per class left to the remainder workgroups; their number per problem
    int splits;
    unsigned xbytes, dybytes;
};

__device__ __forceinline__ uint4 wg1_frag(const char* lo) {        // pixels 4q + r4 and 16 + 4q + r4 of the stage (+ 2048 B)
    const wg1_short4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg1_lds_short4*)(lo));
    const wg1_short4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg1_lds_short4*)(lo + 2048));
    uint4 r;
    r.x = (unsigned)(unsigned short)a[0] | ((unsigned)(unsigned short)a[1] << 16);
    r.y = (unsigned)(unsigned short)a[2] | ((unsigned)(unsigned short)a[3] << 16);
    r.z = (unsigned)(unsigned short)b[0] | ((unsigned)(unsigned short)b[1] << 16);
    r.w = (unsigned)(unsigned short)b[2] | ((unsigned)(unsigned short)b[3] << 16);
    return r;
}

template <typename T>
__global__ __launch_bounds__(512, 2) void conv_wg1_kernel(Wg1P p, Wg1Group grp) {
    constexpr int HALF = 4 * 32 * 128;           // one operand of a stage: [4 blocks of 64 channels][32 pixels][128 B] = 16 KB
    constexpr int STAGE = 2 * HALF;
    constexpr int NST = 4;
    constexpr int NP = 4;                        // pieces per wave and stage (32 / 8): two of X, two of dY
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave & 3, wc = wave >> 2;     // this wave's 64 output channels (dY block wn), its 128 input channels (X blocks 2 wc, 2 wc + 1)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int w = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int prob = w / p.Wp, v = w - prob * p.Wp;
    const i32x4 xw = rsrc_words(grp.x[prob], p.xbytes);
    const i32x4 yw = rsrc_words(grp.dy[prob], p.dybytes);

    const int lpx = 4 * (lane >> 4) + ((lane & 15) >> 2), c4 = lane & 3;
    unsigned fo[4];                              // fragment offset of 16-channel sub-block j inside a 64-channel block
#pragma unroll
    for (int j = 0; j < 4; ++j) fo[j] = (unsigned)(lpx * 128 + (((j ^ (lpx >> 1)) & 3) << 5) + c4 * 8);

    f32x4 acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    auto run_piece = [&](int nb, int cb, int ua, int ub) {
        // piece j of this wave: pi = j * 8 + wave; 0..15: X (block pi >> 2, 8-pixel group pi & 3), 16..31: dY
        unsigned src[NP], dst[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pi = j * 8 + wave, isy = j >> 1, wi = pi & 15, blk = wi >> 2, r8 = wi & 3;      // (j < 2: X, else dY -- in every wave)
            const int px = r8 * 8 + (lane >> 3), pch = lane & 7;
            const unsigned chunk = (unsigned)(((((pch >> 1) ^ (px >> 1)) & 3) << 1) | (pch & 1));
            const unsigned rowb = (unsigned)(isy ? p.ldn : p.C) * 2u;
            src[j] = (unsigned)px * rowb + (unsigned)(((isy ? nb : cb) * 256 + blk * 64) * 2) + chunk * 16u;
            dst[j] = (unsigned)(isy * HALF + blk * 4096 + r8 * 1024);
        }
        const unsigned xstep = 32u * (unsigned)p.C * 2u, ystep = 32u * (unsigned)p.ldn * 2u;
        const int total = ub - ua;
        auto issue = [&](int h) {                // unit ua + h into ring slot h % NST
            const unsigned sbase = lds0 + (unsigned)((h % NST) * STAGE);
            const unsigned u = (unsigned)(ua + h);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (j < 2) dma16_async(xw, sbase + dst[j], u * xstep + src[j]);
                else dma16_async(yw, sbase + dst[j], u * ystep + src[j]);
            }
        };
#pragma unroll
        for (int h = 0; h < NST - 1; ++h)
            if (h < total) issue(h);
        for (int h = 0; h < total; ++h) {
            if (h + NST - 1 <= total) dma_wait<(NST - 2) * NP>();
            else dma_wait<0>();
            __builtin_amdgcn_s_barrier();        // the stage has landed everywhere; every wave is done with the slot about to be refilled
            if (h + NST - 1 < total) issue(h + NST - 1);
            const char* st = smem + (h % NST) * STAGE;
            uint4 fy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fy[j] = wg1_frag(st + HALF + wn * 4096 + fo[j]);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {     // the wave's two 64-channel X blocks
                uint4 fx[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fx[i] = wg1_frag(st + (wc * 2 + hb) * 4096 + fo[i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma16<T>::run(acc[hb * 4 + i][j], fx[i], fy[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the next piece refills the ring from slot 0: every wave must be done reading first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // lane holds c = 4 * (lane >> 4) .. + 3 (rows of D) of sub-block i, n = lane & 15 (column) of sub-block j
    auto flush = [&](int nb, int cb, int slot, bool zeros) {
        float* out = p.slab + ((size_t)prob * p.splits + slot) * (size_t)p.N * p.C;
        const int cc = cb * 256 + wc * 128 + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nb * 256 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!zeros) o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                *reinterpret_cast<float4*>(out + (size_t)n * p.C + cc + i * 16) = o;
            }
        }
    };

    if (v < p.a * p.ncls) {
        const int i = v / p.ncls, cls = v - i * p.ncls;
        const int nb = cls / p.ncb, cb = cls - nb * p.ncb;
        int ua = i * p.L, ub = ua + p.L;
        const int lim = p.U - p.R;
        if (ua > lim) ua = lim;
        if (ub > lim) ub = lim;
        run_piece(nb, cb, ua, ub);
        flush(nb, cb, i, false);
    } else if (p.R > 0) {
        // remainder workgroups: conv_wg3.hip (a range is longer than R: at most two workgroups per class, slots a and a + 1)
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
static int g_wg1 = -1;
static int wg1_mode() {
    if (g_wg1 < 0) {
        const char* e = getenv("MRFP_WGRAD1");      // 0: never; 1 (default): where the rule below says; 2: wherever it is legal (tests, A/B runs)
        g_wg1 = e ? atoi(e) : 1;
    }
    return g_wg1;
}
struct Wg1Plan {
    int ncb, ncls, Wp, a, L, R, Wr, splits, U;
};
static bool wg1_plan(int64_t M, int64_t C, int64_t N, int64_t count, Wg1Plan& pl) {
    if (C % 256 || N % 256 || M % 32 || count < 1 || count > kWg1MaxGroup) return false;
    pl.ncb = (int)(C / 256);
    pl.ncls = (int)(N / 256) * pl.ncb;
    const int64_t U = M / 32;
    if (U >= (1LL << 30)) return false;
    pl.U = (int)U;
    pl.Wp = (int)(kGrid1PerCU / count);          // one 512-thread workgroup per CU, one round
    if ((int64_t)pl.ncls * U < pl.Wp) pl.Wp = (int)((int64_t)pl.ncls * U);
    pl.a = pl.Wp / pl.ncls;
    if (pl.a < 1) return false;
    pl.L = (int)(((int64_t)pl.ncls * U + pl.Wp - 1) / pl.Wp);
    const int64_t main = (int64_t)pl.a * pl.L < U ? (int64_t)pl.a * pl.L : U;
    pl.R = (int)(U - main);
    pl.Wr = pl.Wp - pl.a * pl.ncls;
    if (pl.R > 0 && pl.Wr == 0) return false;
    pl.splits = pl.a + (pl.R > 0 ? 2 : 0);
    return true;
}
int64_t wg1_splits_bound(int64_t N, int64_t Q, int64_t count) {
    if (wg1_mode() == 0 || Q % 256 || N % 256 || count < 1) return 0;
    const int64_t ncls = (N / 256) * (Q / 256);
    const int64_t a = (kGrid1PerCU / count) / ncls;
    return a < 1 ? 0 : a + 2;
}
bool wg1_applicable(int dtype_size, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t ldn, int64_t R, int64_t S, int64_t Ho,
                    int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t count) {
    if (wg1_mode() == 0 || dtype_size != 2) return false;
    if (R != 1 || S != 1 || stride != 1 || Ho != H || Wo != W || pad_h != 0 || pad_w != 0 || (ldn & 7) || ldn < N) return false;
    Wg1Plan pl;
    const int64_t M = B * H * W;
    if (!wg1_plan(M, C, N, count, pl)) return false;
    if (M * C * 2 >= (int64_t)kOOB || M * ldn * 2 >= (int64_t)kOOB) return false;
    if (wg1_mode() >= 2) return true;
    return pl.L >= 48;          // a workgroup ends with 256 KB of slab stores: it needs a K' loop in front of them
}
template <typename T>
static int wg1_launch(const Wg1P& q, const Wg1Group& g, int grid, hipStream_t st) {
    const int lds = 4 * 32768;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wg1_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_wg1_kernel<T>), dim3((unsigned)grid), dim3(512), lds, st, q, g);
    MRFP_LAUNCH_CHECK();
    return 0;
}
// the caller (wgrad_run, conv_wgrad.hip) has checked wg1_applicable(); returns the slab slots per problem in *splits
int wg1_run(const void* const* xs, const void* const* dys, int64_t count, float* slab, bool is_f16, int64_t M, int64_t C, int64_t N, int64_t ldn,
            unsigned xbytes, unsigned dybytes, int* splits, hipStream_t st) {
    Wg1Plan pl;
    if (!wg1_plan(M, C, N, count, pl)) return -1;
    // the slab the caller sized through wg1_splits_bound() holds that many slots per problem
    MRFP_CHECK(pl.splits <= wg1_splits_bound(N, C, count) && (int64_t)count * pl.Wp <= kGrid1PerCU,
               "conv_wg1: %d slab slots per problem / %lld workgroups exceed the workspace rule (%lld)", pl.splits,
               (long long)(count * pl.Wp), (long long)wg1_splits_bound(N, C, count));
    Wg1P q;
    q.slab = slab;
    q.C = (int)C; q.N = (int)N; q.ldn = (int)ldn;
    q.U = pl.U; q.ncb = pl.ncb; q.ncls = pl.ncls;
    q.Wp = pl.Wp; q.a = pl.a; q.L = pl.L; q.R = pl.R; q.Wr = pl.Wr; q.splits = pl.splits;
    q.xbytes = xbytes; q.dybytes = dybytes;
    Wg1Group g;
    for (int i = 0; i < kWg1MaxGroup; ++i) {
        g.x[i] = (const char*)xs[i < count ? i : 0];
        g.dy[i] = (const char*)dys[i < count ? i : 0];
    }
    *splits = pl.splits;
    const int grid = (int)(count * pl.Wp);
    return is_f16 ? wg1_launch<f16>(q, g, grid, st) : wg1_launch<bf16>(q, g, grid, st);
}

}  // namespace mrfp
