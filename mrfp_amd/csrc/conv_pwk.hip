#include "conv_common.hpp"

namespace mrfp {

// Phase clock of the two-group kernel (DIAGNOSTIC build only, -DMRFP_CLOCK_STAMP=1, tools/phase_stamp.py; in the product build every PH_*
// macro is empty and no stamp executes): wave 0 and wave 4 of a workgroup add up the shader cycles (s_memtime) they spend in each phase of
// the K loop and write {cycles, samples} per phase to a buffer of their own that nothing else reads.  Phases: 0 prologue (first transfers
// issued, this wave's weights in registers), 1 wait + barrier at the top of a step, 2 issuing the step's four transfers, 3 fragment reads +
// multiplies, 4 exchange / statistics / stores of a tile, 5 the whole kernel, 6 the whole kernel in 100 MHz ticks (s_memrealtime).
// What it showed (profiles/r06_experiments.md 6): 8.4 - 9.2 of a workgroup's 23.5 - 25 us pass before all its weights are in registers
// (every CU starts 96 KB of cold X transfers and 256 KB of weight loads at once: the start-up burst of the memory system, not a
// bandwidth), the rest streams X at what HBM gives; finishing a tile behind the next barrier and starting on the first weight quarter --
// both built, both bit-identical -- moved time between the phases and left the sum where it was (tools/experiments/conv_pwk_*.patch).
#if MRFP_CLOCK_STAMP
__device__ unsigned long long g_phase_pwk[kStampSlots][2];
int stamps_pwk(unsigned long long* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_pwk), (size_t)n * 16) == hipSuccess ? 0 : -1; }
struct PhaseClock {
    unsigned long long acc[8];
    unsigned long long cnt[8];
    unsigned long long last, t0, r0;
    static __device__ __forceinline__ unsigned long long now() {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        return t;
    }
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc[k] = 0; cnt[k] = 0; }
        r0 = __builtin_amdgcn_s_memrealtime();
        last = t0 = now();
    }
    template <int K> __device__ __forceinline__ void lap() {
        const unsigned long long t = now();
        acc[K] += t - last;
        cnt[K] += 1;
        last = t;
    }
    __device__ __forceinline__ void flush(int wave) {
        const unsigned long long t1 = now(), r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        acc[5] = t1 - t0; cnt[5] = 1;
        acc[6] = r1 - r0; cnt[6] = 1;
        if ((wave == 0 || wave == 4) && (threadIdx.x & 63) == 0) {
            const int slot = ((blockIdx.x % (kStampSlots / 16)) * 2 + (wave ? 1 : 0)) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) { g_phase_pwk[slot + k][0] = acc[k]; g_phase_pwk[slot + k][1] = cnt[k]; }
        }
    }
};
#define PH_START() PhaseClock ph__; ph__.start()
#define PH_LAP(k) ph__.template lap<k>()
#define PH_FLUSH(w) ph__.flush(w)
#else
int stamps_pwk(unsigned long long*, int) { return -1; }
#define PH_START()
#define PH_LAP(k)
#define PH_FLUSH(w)
#endif

// =============================================================================================
// Weight-stationary kernel for the LONG-K pointwise convolutions (round 5): 1x1, stride 1, K = C in {512, 1024, 1280}, 16-bit.
//
// Reference: the bottleneck "reduce" convolutions and the dgrads of the "expand" ones (Resnet.py:156-161, 202-216: 1024 -> 256 at
// 48^2 forty-five times per step of the ResNet-101 trunk; 512 -> 2048, 512 -> 128, 1024 -> 2048, deepv3.py:117-121 bot_aspp 1280 -> 256).
// On the implicit-GEMM tiles a 1024 -> 256 launch at M = 36 864 is ONE round of 768 workgroups that each walk 16 K tiles with a
// single LDS buffer: 33 us = 16 x 2 us of exposed L2 -> LDS fill latency for 3 us of matrix work, and every 96-row tile re-fetches
// its 128 x 1024 weight tile (profiles/r05_experiments.md section 8).  conv_pw.hip removes both for K <= 256; this is the same idea
// where the weights need more registers than two waves per SIMD leave:
//   * the WEIGHTS of a 128-column panel stay in registers for the whole launch -- a wave owns 32 columns x K as K/32 x 2 MFMA A
//     fragments: 256 VGPRs at K = 1 024 (ONE wave per SIMD, one workgroup per CU), 128 at K = 512 (two);
//   * X streams through a ring of [48 rows x 256 K] stages by asynchronous LDS-DMA, NST - 1 stages ahead (120 KB in flight per CU
//     at K = 1 024), one barrier per stage, counted waits;
//   * 48-row M tiles (M = 36 864 = 768 tiles = 6 per workgroup at two panels); the MFMA runs transposed, so a lane stores 8
//     consecutive channels of a pixel straight from its accumulators; fused per-channel statistics (from the fp32 accumulators,
//     as conv_pw.hip) or a (gated) skip-gradient addend.
// =============================================================================================

struct PkP {
    const char* x;       // [M][K] dense
    const char* w;       // forward pack [N][K]
    char* y;             // [M][ldy]
    const char* addend;  // [M][ldy] or null
    const unsigned char* addend_mask;   // 1 bit per addend element or null (ConvP::addend_mask)
    float* colstats;     // [chunks][2][ldy] or null
    int M, N, ldy;
    int tiles;           // ceil(M / 48)
    int panels;          // ceil(N / 128)
    int chunks;          // M-tile ranges per panel (grid = panels * chunks rounded up to 8)
    unsigned xbytes, wbytes, ybytes;
};

constexpr int kPkRows = 48;

template <typename T, int KQ, int NST, bool STATS, bool ADD>
__global__ __launch_bounds__(256, (KQ <= 2 ? 2 : 1)) void conv1x1_longk_kernel(PkP p) {
    constexpr int ROWB = KQ * 512;               // bytes of one row of X
    constexpr int STAGE = 4 * kPkRows * 128;     // [4 blocks of 64 channels][48 rows][128 B] = 24 KB
    constexpr int NP = 6;                        // DMA pieces (8 rows x 128 B) per wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 xw = rsrc_words(p.x, p.xbytes);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend ? p.addend : p.y), 0, (int)p.ybytes, 0x00020000);

    // work: block b -> (chunk, panel), the panels of one chunk (same rows of X) on one XCD (conv_pw.hip)
    const int b = blockIdx.x, xcd = b & 7, rest = b >> 3;
    const int panel = rest % p.panels, chunk = xcd + 8 * (rest / p.panels);
    if (chunk >= p.chunks) return;
    const int per = (p.tiles + p.chunks - 1) / p.chunks;
    const int t0 = chunk * per, t1 = min(p.tiles, t0 + per);
    if (t0 >= t1) return;
    const int n0 = panel * 128 + wave * 32, nl = n0 + 8 * lq;
    const int total = (t1 - t0) * KQ;            // stages this workgroup consumes

    // ---- stage transfers: piece j of this wave = block kb, rows r8 * 8 .. + 7 (source chunk swizzled: the LDS image is lane-linear)
    unsigned src[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int pi = j * 4 + wave, kb = pi / 6, r8 = pi - kb * 6;
        const int row = r8 * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        src[j] = (unsigned)row * (unsigned)ROWB + (unsigned)(kb * 128 + ch * 16);
    }
    auto issue = [&](int g) {                    // stage g = (tile t0 + g / KQ, K quarter g % KQ) into ring slot g % NST
        const int tile = t0 + g / KQ, kq = g - (g / KQ) * KQ;
        const unsigned base = (unsigned)tile * (unsigned)(kPkRows * ROWB) + (unsigned)(kq * 512);     // rows beyond M lie beyond xbytes: zero fill
        const unsigned sbase = lds0 + (unsigned)((g % NST) * STAGE);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pi = j * 4 + wave, kb = pi / 6, r8 = pi - kb * 6;
            dma16_async(xw, sbase + (unsigned)(kb * (kPkRows * 128) + r8 * 1024), base + src[j]);
        }
    };
#pragma unroll
    for (int g = 0; g < NST - 1; ++g)
        if (g < total) issue(g);

    // ---- the weights: 32 columns x K as A fragments (row l15 of block j = channel n0 + 8*(l15>>2) + 4*j + (l15&3), conv_pw.hip) ----
    uint4 fw[KQ * 8][2];
#pragma unroll
    for (int ks = 0; ks < KQ * 8; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 8 * (l15 >> 2) + 4 * j + (l15 & 3);
            fw[ks][j] = bload(wr, n < p.N ? (unsigned)n * (unsigned)ROWB + (unsigned)(ks * 64 + lq * 16) : kOOB);
        }
#pragma unroll
    for (int ks = 0; ks < KQ * 8; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) settle(fw[ks][j]);      // arrived BEFORE the loop (else the compiler's waits for them drain the ring)

    float cs[8], cq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { cs[u] = 0.f; cq[u] = 0.f; }

    int g = 0;
    for (int tile = t0; tile < t1; ++tile) {
        f32x4 acc[3][2];
        uint4 av[3];
        unsigned am[3] = {0xffu, 0xffu, 0xffu};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq, ++g) {
            // all but the NST - 2 younger stages' transfers done (stores and addend loads issued in between only make this stricter)
            if (g + NST - 1 <= total) dma_wait<(NST - 2) * NP>();
            else dma_wait<0>();
            __builtin_amdgcn_s_barrier();         // stage g landed everywhere; stage g - 1 fully consumed
            if (g + NST - 1 < total) issue(g + NST - 1);
            if constexpr (ADD) {
                if (kq == 0) {
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb) {
                        const int m = tile * kPkRows + rb * 16 + l15;
                        const bool ok = m < p.M && nl < p.N;
                        av[rb] = bload(ar, ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)nl) * 2u : kOOB);
                        am[rb] = (p.addend_mask && ok) ? p.addend_mask[((size_t)m * p.ldy + nl) >> 3] : 0xffu;
                    }
                }
            }
            const char* st = smem + (g % NST) * STAGE;
            uint4 fx[2][3];
            auto read_x = [&](int ks, uint4 (&f)[3]) {
                const char* ab = st + (ks >> 1) * (kPkRows * 128);
                const int ch = (ks & 1) * 4 + lq;
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) f[rb] = *reinterpret_cast<const uint4*>(ab + lds_off(rb * 16 + l15, ch));
            };
            read_x(0, fx[0]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) read_x(ks + 1, fx[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (kq == 0 && ks == 0) {
                            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                            acc[rb][j] = z;
                        }
                        Mma16<T>::run(acc[rb][j], fw[kq * 8 + ks][j], fx[ks & 1][rb]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue of this tile: 8 consecutive channels of one pixel per lane and row block ----------------------------------
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int m = tile * kPkRows + rb * 16 + l15;
            const bool ok = m < p.M && nl < p.N;
            if constexpr (STATS) {       // from the fp32 accumulators (rows beyond M were zero-filled: they add exactly 0)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cs[u] += acc[rb][0][u];
                    cs[4 + u] += acc[rb][1][u];
                    cq[u] = __builtin_fmaf(acc[rb][0][u], acc[rb][0][u], cq[u]);
                    cq[4 + u] = __builtin_fmaf(acc[rb][1][u], acc[rb][1][u], cq[4 + u]);
                }
            }
            uint4 v;
            if constexpr (ADD) {
                const uint4 gt = gate_chunk16(av[rb], am[rb]);
                float a[8];
                unpack2<T>(gt.x, a[0], a[1]);
                unpack2<T>(gt.y, a[2], a[3]);
                unpack2<T>(gt.z, a[4], a[5]);
                unpack2<T>(gt.w, a[6], a[7]);
                v.x = pack2<T>(acc[rb][0][0] + a[0], acc[rb][0][1] + a[1]);
                v.y = pack2<T>(acc[rb][0][2] + a[2], acc[rb][0][3] + a[3]);
                v.z = pack2<T>(acc[rb][1][0] + a[4], acc[rb][1][1] + a[5]);
                v.w = pack2<T>(acc[rb][1][2] + a[6], acc[rb][1][3] + a[7]);
            } else {
                v.x = pack2<T>(acc[rb][0][0], acc[rb][0][1]);
                v.y = pack2<T>(acc[rb][0][2], acc[rb][0][3]);
                v.z = pack2<T>(acc[rb][1][0], acc[rb][1][1]);
                v.w = pack2<T>(acc[rb][1][2], acc[rb][1][3]);
            }
            u32x4 dv;
            dv.x = v.x; dv.y = v.y; dv.z = v.z; dv.w = v.w;
            __builtin_amdgcn_raw_buffer_store_b128(dv, yr, (int)(ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)nl) * 2u : kOOB), 0, 0);
        }
    }
    if constexpr (STATS) {
        float* out = p.colstats + (size_t)chunk * 2 * p.ldy;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cs[u] = row16_sum(cs[u]);
            cq[u] = row16_sum(cq[u]);
        }
        if (l15 == 0 && nl < p.N) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                out[nl + u] = cs[u];
                out[p.ldy + nl + u] = cq[u];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K = 1 024 with TWO waves per SIMD.  The one-wave form above pays for every stage twice: a wave issues its 6 transfers (100 - 185
// cycles each beside fragment reads: MI355X_MICROARCH.md) and then multiplies 48 MFMAs (768 cycles), and with one wave per SIMD nothing
// runs beside either -- 1024 -> 256 at M = 36 864 took 34 us on it, exactly what the implicit-GEMM tile takes (r05_experiments.md 11).
// Here the 8 waves of a 512-thread workgroup split K: waves 0-3 hold the weights of K quarters 0 and 1 (128 registers), waves 4-7 those
// of quarters 2 and 3, so every SIMD has a wave of each group and one multiplies while the other issues.  A step = one quarter per
// group of a 32-row tile (a 32 KB pair of stages, ring of four pairs); after its two steps group 1 hands its partial sums to group 0
// through LDS (16 KB, fp32, lane-matched), group 0 adds them and runs the epilogue.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kPk2Rows = 32;

template <typename T, bool STATS, bool ADD>
__global__ __launch_bounds__(512, 2) void conv1x1_longk2_kernel(PkP p) {
    constexpr int ROWB = 2048;                   // K = 1 024 elements
    constexpr int QTR = 4 * kPk2Rows * 128;      // one K quarter of a tile: [4 blocks][32 rows][128 B] = 16 KB
    constexpr int PAIR = 2 * QTR;                // a step's two quarters
    constexpr int NST = 4;                       // pairs in the ring
    constexpr int NP = 4;                        // DMA pieces per wave and step (32 pieces / 8 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const exch = smem + NST * PAIR;        // group 1 -> group 0 partial sums
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 xw = rsrc_words(p.x, p.xbytes);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend ? p.addend : p.y), 0, (int)p.ybytes, 0x00020000);

    const int b = blockIdx.x, xcd = b & 7, rest = b >> 3;
    const int panel = rest % p.panels, chunk = xcd + 8 * (rest / p.panels);
    if (chunk >= p.chunks) return;
    const int per = (p.tiles + p.chunks - 1) / p.chunks;
    const int t0 = chunk * per, t1 = min(p.tiles, t0 + per);
    if (t0 >= t1) return;
    const int n0 = panel * 128 + wq * 32, nl = n0 + 8 * lq;
    const int total = (t1 - t0) * 2;             // steps
    PH_START();

    // piece j of this wave: pi = j * 8 + wave -> quarter slot pi / 16 (0: group 0's, 1: group 1's), block kb, rows r8 * 8 .. + 7
    unsigned src[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int pi = j * 8 + wave, qs = pi >> 4, wi = pi & 15, kb = wi >> 2, r8 = wi & 3;
        const int row = r8 * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        src[j] = (unsigned)row * (unsigned)ROWB + (unsigned)(qs * 1024 + kb * 128 + ch * 16);      // (+ step * 512: quarter s / 2 + s)
    }
    auto issue = [&](int h) {                    // step h = (tile t0 + h / 2, s = h & 1) into ring slot h % NST
        const int tile = t0 + (h >> 1), sst = h & 1;
        const unsigned base = (unsigned)tile * (unsigned)(kPk2Rows * ROWB) + (unsigned)(sst * 512);
        const unsigned sbase = lds0 + (unsigned)((h % NST) * PAIR);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pi = j * 8 + wave, qs = pi >> 4, wi = pi & 15, kb = wi >> 2, r8 = wi & 3;
            dma16_async(xw, sbase + (unsigned)(qs * QTR + kb * (kPk2Rows * 128) + r8 * 1024), base + src[j]);
        }
    };
#pragma unroll
    for (int h = 0; h < NST - 1; ++h)
        if (h < total) issue(h);

    // weights of this group's two K quarters: in step s group 0 multiplies quarter s, group 1 quarter 2 + s
    uint4 fw[16][2];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 8 * (l15 >> 2) + 4 * j + (l15 & 3);
            fw[ks][j] = bload(wr, n < p.N ? (unsigned)n * (unsigned)ROWB + (unsigned)(grp * 1024 + ks * 64 + lq * 16) : kOOB);
        }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) settle(fw[ks][j]);
    PH_LAP(0);

    float cs[8], cq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { cs[u] = 0.f; cq[u] = 0.f; }

    int h = 0;
    for (int tile = t0; tile < t1; ++tile) {
        f32x4 acc[2][2];
        uint4 av[2];
        unsigned am[2] = {0xffu, 0xffu};
#pragma unroll
        for (int sst = 0; sst < 2; ++sst, ++h) {
            if (h + NST - 1 <= total) dma_wait<(NST - 2) * NP>();
            else dma_wait<0>();
            __builtin_amdgcn_s_barrier();
            PH_LAP(1);
            if (h + NST - 1 < total) issue(h + NST - 1);
            PH_LAP(2);
            if constexpr (ADD) {
                if (sst == 0 && grp == 0) {
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const int m = tile * kPk2Rows + rb * 16 + l15;
                        const bool ok = m < p.M && nl < p.N;
                        av[rb] = bload(ar, ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)nl) * 2u : kOOB);
                        am[rb] = (p.addend_mask && ok) ? p.addend_mask[((size_t)m * p.ldy + nl) >> 3] : 0xffu;
                    }
                }
            }
            const char* st = smem + (h % NST) * PAIR + grp * QTR;
            uint4 fx[2][2];
            auto read_x = [&](int ks, uint4 (&f)[2]) {
                const char* ab = st + (ks >> 1) * (kPk2Rows * 128);
                const int ch = (ks & 1) * 4 + lq;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) f[rb] = *reinterpret_cast<const uint4*>(ab + lds_off(rb * 16 + l15, ch));
            };
            read_x(0, fx[0]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) read_x(ks + 1, fx[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (sst == 0 && ks == 0) {
                            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                            acc[rb][j] = z;
                        }
                        Mma16<T>::run(acc[rb][j], fw[sst * 8 + ks][j], fx[ks & 1][rb]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            PH_LAP(3);
        }
        // ---- group 1 hands its partial sums over (lane-matched 16-byte slots), group 0 adds them and stores ----------------------
        f32x4* ex = reinterpret_cast<f32x4*>(exch) + (wq * 4) * 64 + lane;
        if (grp == 1) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int j = 0; j < 2; ++j) ex[(rb * 2 + j) * 64] = acc[rb][j];
        }
        // (NOT __syncthreads(): its fence would also drain the vector-memory counter -- the three pairs of transfers in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 o = ex[(rb * 2 + j) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[rb][j][e] += o[e];
                }
                const int m = tile * kPk2Rows + rb * 16 + l15;
                const bool ok = m < p.M && nl < p.N;
                if constexpr (STATS) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        cs[u] += acc[rb][0][u];
                        cs[4 + u] += acc[rb][1][u];
                        cq[u] = __builtin_fmaf(acc[rb][0][u], acc[rb][0][u], cq[u]);
                        cq[4 + u] = __builtin_fmaf(acc[rb][1][u], acc[rb][1][u], cq[4 + u]);
                    }
                }
                uint4 v;
                if constexpr (ADD) {
                    const uint4 gt = gate_chunk16(av[rb], am[rb]);
                    float a[8];
                    unpack2<T>(gt.x, a[0], a[1]);
                    unpack2<T>(gt.y, a[2], a[3]);
                    unpack2<T>(gt.z, a[4], a[5]);
                    unpack2<T>(gt.w, a[6], a[7]);
                    v.x = pack2<T>(acc[rb][0][0] + a[0], acc[rb][0][1] + a[1]);
                    v.y = pack2<T>(acc[rb][0][2] + a[2], acc[rb][0][3] + a[3]);
                    v.z = pack2<T>(acc[rb][1][0] + a[4], acc[rb][1][1] + a[5]);
                    v.w = pack2<T>(acc[rb][1][2] + a[6], acc[rb][1][3] + a[7]);
                } else {
                    v.x = pack2<T>(acc[rb][0][0], acc[rb][0][1]);
                    v.y = pack2<T>(acc[rb][0][2], acc[rb][0][3]);
                    v.z = pack2<T>(acc[rb][1][0], acc[rb][1][1]);
                    v.w = pack2<T>(acc[rb][1][2], acc[rb][1][3]);
                }
                u32x4 dv;
                dv.x = v.x; dv.y = v.y; dv.z = v.z; dv.w = v.w;
                __builtin_amdgcn_raw_buffer_store_b128(dv, yr, (int)(ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)nl) * 2u : kOOB), 0, 0);
            }
        }
        PH_LAP(4);
    }
    PH_FLUSH(wave);
    if constexpr (STATS) {
        if (grp == 0) {
            float* out = p.colstats + (size_t)chunk * 2 * p.ldy;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cs[u] = row16_sum(cs[u]);
                cq[u] = row16_sum(cq[u]);
            }
            if (l15 == 0 && nl < p.N) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    out[nl + u] = cs[u];
                    out[p.ldy + nl + u] = cq[u];
                }
            }
        }
    }
}

static int g_pwk = -1;
bool pwk_applicable(const ConvP& p, int esz) {
    if (g_pwk < 0) {
        const char* e = getenv("MRFP_CONV_PWK");
        g_pwk = e ? atoi(e) : 1;
    }
    if (!g_pwk || esz != 2 || p.bias != nullptr || (p.colstats && p.addend) || p.rowweight) return false;
    if (p.R != 1 || p.S != 1 || p.stride != 1 || p.sstride != 1 || p.pad_h != 0 || p.pad_w != 0) return false;
    if (p.Ho != p.H || p.Wo != p.W || p.N < 128 || (p.N & 7) != 0) return false;
    if ((int64_t)p.M * p.ldy * esz >= (int64_t)kOOB) return false;
    const int rowb = p.C * esz;
    if (rowb != 1024 && rowb != 2048 && rowb != 2560) return false;
    // a workgroup first loads 128 - 320 registers of weights per wave: at least four 48-row tiles each (MRFP_CONV_PWK=2: always)
    const int rows = rowb == 2048 ? kPk2Rows : kPkRows;
    const int64_t tiles = (p.M + rows - 1) / rows, panels = (p.N + 127) / 128;
    const int64_t slots = rowb == 1024 ? 512 : 256;
    return g_pwk >= 2 || tiles * panels >= 4 * slots;
}
// M-tile ranges per panel: one (K >= 1024) or two (K = 512) workgroups per CU, a multiple of 8 (the XCD mapping), no range empty
static int pwk_rows(int rowb) { return rowb == 2048 ? kPk2Rows : kPkRows; }      // tile height: K = 1 024 runs the two-group kernel
static int pwk_chunks(int M, int N, int rowb) {
    const int tiles = (M + pwk_rows(rowb) - 1) / pwk_rows(rowb), panels = (N + 127) / 128;
    int chunks = (rowb == 1024 ? 512 : 256) / panels;
    chunks = chunks / 8 * 8;
    if (chunks < 8) chunks = 8;
    if (chunks > tiles) chunks = (tiles + 7) / 8 * 8;
    const int per = (tiles + chunks - 1) / chunks;
    return (tiles + per - 1) / per;
}
int64_t pwk_stats_blocks(const ConvP& p) { return (int64_t)pwk_chunks(p.M, p.N, p.C * 2); }
int64_t pwk_stats_block_rows(const ConvP& p) {
    const int rows = pwk_rows(p.C * 2);
    const int tiles = (p.M + rows - 1) / rows, chunks = pwk_chunks(p.M, p.N, p.C * 2);
    return (int64_t)((tiles + chunks - 1) / chunks) * rows;
}

template <typename T, bool STATS, bool ADD>
static int pwk2_launch(const ConvP& c, hipStream_t st) {
    constexpr int lds = 4 * 2 * 4 * kPk2Rows * 128 + kPk2Rows * 128 * 4;      // ring of four 32 KB pairs + the 16 KB exchange
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_longk2_kernel<T, STATS, ADD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    PkP p;
    p.x = c.x; p.w = c.w; p.y = c.y; p.addend = c.addend; p.addend_mask = c.addend_mask; p.colstats = c.colstats;
    p.M = c.M; p.N = c.N; p.ldy = c.ldy;
    p.tiles = (c.M + kPk2Rows - 1) / kPk2Rows;
    p.panels = (c.N + 127) / 128;
    p.chunks = pwk_chunks(c.M, c.N, 2048);
    const int chunks = (p.chunks + 7) / 8 * 8;
    p.xbytes = c.xbytes; p.wbytes = c.wbytes; p.ybytes = (unsigned)((int64_t)c.M * c.ldy * 2);
    hipLaunchKernelGGL((conv1x1_longk2_kernel<T, STATS, ADD>), dim3((unsigned)(p.panels * chunks)), dim3(512), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T, int KQ, bool STATS, bool ADD>
static int pwk_launch(const ConvP& c, hipStream_t st) {
    constexpr int NST = KQ <= 2 ? 3 : 6;
    constexpr int lds = NST * 4 * kPkRows * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_longk_kernel<T, KQ, NST, STATS, ADD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    PkP p;
    p.x = c.x; p.w = c.w; p.y = c.y; p.addend = c.addend; p.addend_mask = c.addend_mask; p.colstats = c.colstats;
    p.M = c.M; p.N = c.N; p.ldy = c.ldy;
    p.tiles = (c.M + kPkRows - 1) / kPkRows;
    p.panels = (c.N + 127) / 128;
    p.chunks = pwk_chunks(c.M, c.N, c.C * 2);
    const int chunks = (p.chunks + 7) / 8 * 8;
    p.xbytes = c.xbytes; p.wbytes = c.wbytes; p.ybytes = (unsigned)((int64_t)c.M * c.ldy * 2);
    hipLaunchKernelGGL((conv1x1_longk_kernel<T, KQ, NST, STATS, ADD>), dim3((unsigned)(p.panels * chunks)), dim3(256), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}
template <typename T, int KQ>
static int pwk_pick(const ConvP& p, hipStream_t st) {
    if (p.colstats) return pwk_launch<T, KQ, true, false>(p, st);
    if (p.addend) return pwk_launch<T, KQ, false, true>(p, st);
    return pwk_launch<T, KQ, false, false>(p, st);
}
template <typename T>
static int pwk_run_t(const ConvP& p, hipStream_t st) {
    const int kq = p.C * 2 / 512;
    static int two = -1;               // MRFP_CONV_PWK2=0: K = 1 024 on the one-wave-per-SIMD form (A/B runs)
    if (two < 0) { const char* e = getenv("MRFP_CONV_PWK2"); two = e ? atoi(e) : 1; }
    if (kq == 4 && two) {
        if (p.colstats) return pwk2_launch<T, true, false>(p, st);
        if (p.addend) return pwk2_launch<T, false, true>(p, st);
        return pwk2_launch<T, false, false>(p, st);
    }
    return kq == 2 ? pwk_pick<T, 2>(p, st) : kq == 4 ? pwk_pick<T, 4>(p, st) : pwk_pick<T, 5>(p, st);
}
int pwk_run(const ConvP& p, bool is_f16, hipStream_t st) { return is_f16 ? pwk_run_t<f16>(p, st) : pwk_run_t<bf16>(p, st); }

}  // namespace mrfp
