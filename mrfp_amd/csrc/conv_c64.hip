#include "conv_common.hpp"

namespace mrfp {

// =============================================================================================
// Weight-stationary 3x3 kernel for the 64-input-channel layers of the HRFP branch (round 5).
//
// Reference: the random over-complete encoder / decoder of MRFP+ (deepv3.py:221-237, 320-327): 3x3 convolutions, dilation 1 or 2,
// 64 / 128 channels on maps of 192^2 .. 384^2 (odd sizes 231, 277, 321 included), forward and dgrad.  On the implicit-GEMM tiles
// these launches are bound by the L2 -> LDS fill path (profiles/r05_experiments.md section 2: 52 - 122 FLOP per fill byte, 600 - 870
// TFLOP/s): with K = 9 x 64 every tile re-fetches its pixels once per tap (or per filter row) AND the whole 74 - 147 KB weight
// tensor.  Here nothing is fetched twice:
//   * the WEIGHTS live in registers for the whole launch: a wave owns 32 output channels x K = 576 as 18 x 2 MFMA A fragments
//     (144 VGPRs), loaded once per (persistent) workgroup;
//   * the PIXELS stream through a rolling window of four image-row strips in LDS (LDS-DMA, one new input row per output row:
//     an input row is fetched once per column strip, + 2 dil halo pixels); the nine taps are nine shifted reads of that window;
//   * a workgroup walks a contiguous span of output row strips (balanced to one row over the grid), rows of one dilation class
//     (oh = q mod dil) in sequence so that the three filter rows are always neighbours in the window;
//   * the MFMA runs transposed (accumulator rows = channels, columns = pixels; the weight rows are permuted at load time) so that
//     a lane stores 8 consecutive channels of its pixel straight from the accumulators: no LDS transposition, no epilogue barrier;
//     bias enters as the initial accumulator; per-channel statistics (optionally weighted per pixel: ConvP::rowweight) are summed
//     from the fp32 accumulators and leave as ONE partial row per (workgroup, pixel sub-strip).
// Fill bytes per output: one 128-byte pixel per 64 / 128 outputs x 576 MACs -- 8 B/clk per CU at full matrix rate, a third of what
// the path delivers beside compute: the kernel is bound by the matrix pipe / HBM, not by the fill path.
// Geometry: N = 64: waves = 2 channel groups x 2 pixel sub-strips of 64 (strip of 128 pixels); N = 128: 4 channel groups x 64 pixels.
// CB = 2 (128 input channels: K = 1 152, the 128 -> 64 / 128 / 256 layers): 288 weight registers per wave, so ONE wave per SIMD
// (512 registers) and one workgroup per CU; a window pixel is two 128-byte halves kept as two [pixel][128 B] planes (each with the
// conflict-free swizzle of the 64-channel window); N = 256 runs as two column groups of 128 (grid.y).
// =============================================================================================

struct C64P {
    const char* x;       // [B][H][W][C] 16-bit, C = 64 * CB
    const char* w;       // forward-form pack [N][3][3][C]
    char* y;             // [B][H][W][ldy]
    const float* bias;   // [N] or null
    const char* addend;  // [M][ldy] or null
    float* colstats;     // [B][PSN][spi][2][ldy] or null: per IMAGE, per pixel sub-strip, one row per workgroup that touches the image
    const unsigned char* rowweight;   // [M] or null (weighted statistics)
    int B, H, W, N, ldy, dil;
    int strips;          // column strips per image row
    int units;           // B * strips * H  (output row strips)
    int spi;             // statistics row slots per image and sub-strip (>= workgroups that can touch one image)
    unsigned xbytes, wbytes, ybytes;
};

constexpr int kC64Slots = 4;

template <typename T, int CB, int NCG, bool STATS, bool ADD>
__global__ __launch_bounds__(256, (CB == 1 ? 2 : 1)) void conv3x3_c64_kernel(C64P p) {
    constexpr int PSN = 4 / NCG;                 // pixel sub-strips per workgroup
    constexpr int SW = 64 * PSN;                 // strip width
    constexpr int NPIECE = (SW + 4 + 7) / 8;     // 8-pixel DMA pieces per window row and 64-channel half (dilation <= 2: 2 halo pixels either side)
    constexpr int NPC = CB * NPIECE;             // pieces per window row
    constexpr int PWV = (NPC + 3) / 4;           // rounds of piece issue per row (wave w takes pieces w, w + 4, ...)
    constexpr int SLOT = NPC * 1024;
    constexpr int KS = 18 * CB;                  // k steps of 32 channels: (filter row, tap, half, step)
    constexpr int PIXB = 128 * CB;               // bytes per pixel of x
    constexpr int ST = 4;                        // output stores per wave and row (unconditional: counted by vmcnt)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int cg = wave % NCG, ps = wave / NCG;
    const int n0 = (int)blockIdx.y * 128 + cg * 32, nl = n0 + 8 * lq;
    const int d = p.dil;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 xw = rsrc_words(p.x, p.xbytes);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend ? p.addend : p.y), 0, (int)p.ybytes, 0x00020000);

    // ---- weights: this wave's 32 channels x 576, as MFMA A fragments (row l15 of block j = channel n0 + 8*(l15>>2) + 4*j + (l15&3)) ----
    uint4 fw[KS][2];      // (the pack is [N][tap][C]: k step ks = ((tap * CB + half) * 2 + step) is 64 contiguous bytes at ks * 64)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 8 * (l15 >> 2) + 4 * j + (l15 & 3);
            fw[ks][j] = bload(wr, n < p.N ? (unsigned)n * (unsigned)(1152 * CB) + (unsigned)(ks * 64 + lq * 16) : kOOB);
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) settle(fw[ks][j]);
    float bv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = (p.bias && nl + 4 * j + e < p.N) ? p.bias[nl + 4 * j + e] : 0.f;

    // ---- fragment read offsets inside a window row: tap s, k step kk; pixel blocks add 2 KB each --------------------------------
    // window pixel of strip pixel q under tap s = q + s * d (window pixel 0 = image column ow0 - d); chunk (kk*4 + lq) ^ (pixel & 7)
    unsigned foff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int px = ps * 64 + l15 + s * d;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) foff[s][kk] = (unsigned)(px * 128 + ((((kk * 4 + lq) ^ px) & 7) << 4));
    }

    float cs[8], cq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { cs[u] = 0.f; cq[u] = 0.f; }

    // ---- this workgroup's span of output row strips ----------------------------------------------------------------------------
    const long long G = gridDim.x;
    int u = (int)((long long)blockIdx.x * p.units / G);
    const int u1 = (int)((long long)(blockIdx.x + 1) * p.units / G);
    const int Hc0 = (p.H + d - 1) / d;            // rows of dilation class 0 (class 1, d = 2: H / 2)
    // Statistics leave PER IMAGE (an InstanceNorm consumer -- reference Resnet.py:534-536, the stem -- needs plane sums; a BatchNorm
    // consumer adds all rows): row (image b, sub-strip ps, slot) with slot = this workgroup's rank among the workgroups whose span
    // touches image b; the last of them zeroes the unused slots, so every row of the buffer is written in every launch.
    const int upi = p.strips * p.H;               // row strips per image
    int cur_b = -1;
    auto flush = [&](int b) {
        const long long lo = (long long)b * upi, hi = lo + upi;      // units of image b
        int wf = (int)(lo * G / p.units);         // first workgroup whose span reaches into [lo, hi)
        while ((long long)(wf + 1) * p.units / G <= lo) ++wf;
        while (wf > 0 && (long long)wf * p.units / G > lo) --wf;
        const int slot = (int)blockIdx.x - wf;
        float* out = p.colstats + ((size_t)(b * PSN + ps) * p.spi + slot) * 2 * p.ldy;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cs[k] = row16_sum(cs[k]);
            cq[k] = row16_sum(cq[k]);
        }
        if (l15 == 0 && nl < p.N) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                out[nl + k] = cs[k];
                out[p.ldy + nl + k] = cq[k];
            }
            if ((long long)u1 >= hi) {            // the last workgroup of this image: the slots nobody owns
                for (int z = slot + 1; z < p.spi; ++z) {
                    float* oz = p.colstats + ((size_t)(b * PSN + ps) * p.spi + z) * 2 * p.ldy;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        oz[nl + k] = 0.f;
                        oz[p.ldy + nl + k] = 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { cs[k] = 0.f; cq[k] = 0.f; }
    };
    while (u < u1) {
        // decode: (image, strip) block of H rows, enumerated class-major
        const int blk_id = u / p.H, tt = u - blk_id * p.H;
        const int b = blk_id / p.strips, strip = blk_id - b * p.strips;
        const int q = (d == 2 && tt >= Hc0) ? 1 : 0;
        const int i0 = tt - q * Hc0;
        const int Hc = q ? p.H - Hc0 : Hc0;
        int i1 = i0 + (u1 - u);
        if (i1 > Hc) i1 = Hc;
        const int ow0 = strip * SW;
        if constexpr (STATS) {
            if (b != cur_b) {
                if (cur_b >= 0) flush(cur_b);
                cur_b = b;
            }
        }
        const unsigned imgbase = (unsigned)b * (unsigned)p.H * (unsigned)p.W * (unsigned)PIXB;
        const int col0 = ow0 - d + (lane >> 3);   // image column of this lane's pixel in piece 0
        auto issue_row = [&](int i) {            // class row i -> image row q + d * i, into slot (i + 4) & 3
            const int ih = q + d * i;
            const bool rok = i >= 0 && ih < p.H;
            const unsigned rbase = imgbase + (unsigned)(rok ? ih : 0) * (unsigned)p.W * (unsigned)PIXB;
            const unsigned sbase = lds0 + (unsigned)(((i + 4) & 3) * SLOT);
#pragma unroll
            for (int j = 0; j < PWV; ++j) {
                const int pi = j * 4 + wave;          // (wave-uniform: the transfer's LDS address is a scalar)
                // (a wave without a piece in this round issues nothing: the counted wait below leaves only the output STORES
                //  outstanding, so the number of transfers per wave need not be uniform)
                if (pi >= NPC) continue;
                const int hf = pi / NPIECE, pp = pi - hf * NPIECE;      // 64-channel half, 8-pixel piece inside it
                // (the column offsets are recomputed per row -- a dozen vector instructions -- rather than held in registers across
                //  the multiplies of a row: the weights take 144 of the 256 registers)
                const int col = col0 + 8 * pp, px = 8 * pp + (lane >> 3);
                const bool cok = col >= 0 && col < p.W;
                const unsigned cb = (unsigned)(col * PIXB + hf * 128 + (((lane & 7) ^ (px & 7)) << 4));
                dma16_async(xw, sbase + (unsigned)(pi * 1024), (rok && cok) ? rbase + cb : kOOB);
            }
        };
        // warm-up: the window rows of the first output row
        issue_row(i0 - 1);
        issue_row(i0);
        issue_row(i0 + 1);
        for (int i = i0; i < i1; ++i) {
            if (i == i0) dma_wait<0>();
            else dma_wait<ST>();                  // all but the previous row's stores: row i + 1 has landed
            __builtin_amdgcn_s_barrier();         // ... everywhere; and every wave is done with row i - 2's slot
            if (i + 1 < i1) issue_row(i + 2);
            const int oh = q + d * i;
            // addend / statistics weights of this row: fetched before the multiplies, used behind them
            uint4 av[4];
            unsigned wt[4];
            const int owl = ow0 + ps * 64 + l15;                                  // this lane's column in pixel block 0
            const unsigned ml = (unsigned)((b * p.H + oh) * p.W + owl);           // ... and its output row index
            auto out_off = [&](int bk) {
                return (owl + bk * 16 < p.W && nl < p.N) ? ((ml + (unsigned)(bk * 16)) * (unsigned)p.ldy + (unsigned)nl) * 2u : kOOB;
            };
#pragma unroll
            for (int bk = 0; bk < 4; ++bk) {
                if constexpr (ADD) av[bk] = bload(ar, out_off(bk));
                if constexpr (STATS) {
                    const bool ok = owl + bk * 16 < p.W;
                    wt[bk] = ok ? (p.rowweight ? (unsigned)p.rowweight[ml + bk * 16] : 1u) : 0u;
                }
            }
            f32x4 acc[4][2];
            const unsigned sb[3] = {lds0 + (unsigned)(((i + 3) & 3) * SLOT), lds0 + (unsigned)((i & 3) * SLOT),
                                    lds0 + (unsigned)(((i + 1) & 3) * SLOT)};
            uint4 fx[2][4];
            auto read_x = [&](int ks, uint4 (&f)[4]) {
                const int r = ks / (6 * CB), s = (ks / (2 * CB)) % 3, hf = (ks >> 1) % CB, kk = ks & 1;
                const char* base = smem + (sb[r] - lds0) + hf * (NPIECE * 1024) + foff[s][kk];
#pragma unroll
                for (int bk = 0; bk < 4; ++bk) f[bk] = *reinterpret_cast<const uint4*>(base + bk * 2048);
            };
            read_x(0, fx[0]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) read_x(ks + 1, fx[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bk = 0; bk < 4; ++bk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (ks == 0) {
                            const f32x4 z = {bv[j][0], bv[j][1], bv[j][2], bv[j][3]};     // bias as the initial accumulator
                            acc[bk][j] = z;
                        }
                        Mma16<T>::run(acc[bk][j], fw[ks][j], fx[ks & 1][bk]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: 8 consecutive channels of one pixel per lane and pixel block
#pragma unroll
            for (int bk = 0; bk < 4; ++bk) {
                uint4 v;
                if constexpr (ADD) {
                    float a[8];
                    unpack2<T>(av[bk].x, a[0], a[1]);
                    unpack2<T>(av[bk].y, a[2], a[3]);
                    unpack2<T>(av[bk].z, a[4], a[5]);
                    unpack2<T>(av[bk].w, a[6], a[7]);
                    v.x = pack2<T>(acc[bk][0][0] + a[0], acc[bk][0][1] + a[1]);
                    v.y = pack2<T>(acc[bk][0][2] + a[2], acc[bk][0][3] + a[3]);
                    v.z = pack2<T>(acc[bk][1][0] + a[4], acc[bk][1][1] + a[5]);
                    v.w = pack2<T>(acc[bk][1][2] + a[6], acc[bk][1][3] + a[7]);
                } else {
                    v.x = pack2<T>(acc[bk][0][0], acc[bk][0][1]);
                    v.y = pack2<T>(acc[bk][0][2], acc[bk][0][3]);
                    v.z = pack2<T>(acc[bk][1][0], acc[bk][1][1]);
                    v.w = pack2<T>(acc[bk][1][2], acc[bk][1][3]);
                }
                if constexpr (STATS) {
                    // statistics of the STORED (rounded) values, each pixel counted wt times: what a statistics pass over the
                    // (resized) output would sum -- the expression order of the generic kernels' weighted epilogue
                    float f[8];
                    unpack2<T>(v.x, f[0], f[1]);
                    unpack2<T>(v.y, f[2], f[3]);
                    unpack2<T>(v.z, f[4], f[5]);
                    unpack2<T>(v.w, f[6], f[7]);
                    const float wb = (float)wt[bk];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float wf = f[k] * wb;
                        cs[k] += wf;
                        cq[k] += wf * f[k];
                    }
                }
                u32x4 dv;
                dv.x = v.x; dv.y = v.y; dv.z = v.z; dv.w = v.w;
                __builtin_amdgcn_raw_buffer_store_b128(dv, yr, (int)out_off(bk), 0, 0);
            }
        }
        u += i1 - i0;
        // the next segment refills the window from scratch: every wave must be done reading this one's rows first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (STATS) {
        if (cur_b >= 0) flush(cur_b);
    }
}

static int g_c64 = -1;
bool c64_applicable(const ConvP& p, int esz) {
    if (g_c64 < 0) {
        const char* e = getenv("MRFP_CONV_C64");
        g_c64 = e ? atoi(e) : 1;
    }
    if (!g_c64 || esz != 2) return false;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.sstride != 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.dil < 1 || p.dil > 2 || p.pad_h != p.dil || p.pad_w != p.dil) return false;
    if ((p.ldy & 7) != 0) return false;
    if (p.C == 64) { if (p.N != 64 && p.N != 128) return false; }
    else if (p.C == 128) {        // MRFP_CONV_C128=0: the 128-channel layers stay on the implicit-GEMM tiles (A/B runs)
        static int c128 = -1;
        if (c128 < 0) { const char* e = getenv("MRFP_CONV_C128"); c128 = e ? atoi(e) : 1; }
        if (!c128 || (p.N != 64 && p.N != 128 && p.N != 256)) return false;
        // one workgroup per CU that first loads 288 registers of weights per wave: it pays from ~32 output row strips per workgroup on
        // (measured in the step: 128 -> 64 @256^2 188 -> 133 us, @192^2 105 -> 130 us; 128 -> 128 @96^2 52 -> 64 us).  MRFP_CONV_C128=2: always (tests)
        const int SW = p.N == 64 ? 128 : 64;
        if (c128 < 2 && (int64_t)p.B * ((p.W + SW - 1) / SW) * p.H < 32 * kGrid1PerCU) return false;
    } else return false;
    if (p.addend_mask || (p.colstats && p.addend)) return false;
    if ((int64_t)p.M * p.ldy * esz >= (int64_t)kOOB || p.H < 2 * p.dil) return false;
    return true;
}
static int c64_grid(const ConvP& p) {
    const int SW = p.N == 64 ? 128 : 64;
    const int64_t units = (int64_t)p.B * ((p.W + SW - 1) / SW) * p.H;
    const int64_t cap = p.C == 64 ? kGrid2PerCU : kGrid1PerCU;     // two workgroups per CU (64 channels) / one (128: 512 registers per wave)
    return (int)(units < cap ? units : cap);       // every workgroup owns at least one row strip
}
// statistics row slots per image and pixel sub-strip: an upper bound of the workgroups whose span touches one image
static int c64_spi(const ConvP& p) {
    const int SW = p.N == 64 ? 128 : 64;
    const int64_t upi = (int64_t)((p.W + SW - 1) / SW) * p.H, units = upi * p.B;
    const int64_t lmin = units / c64_grid(p);      // shortest span (>= 1)
    return (int)(upi / lmin + 2);
}
int64_t c64_stats_blocks(const ConvP& p) { return (int64_t)p.B * (p.N == 64 ? 2 : 1) * c64_spi(p); }
int64_t c64_stats_block_rows(const ConvP& p) { return -(int64_t)(p.N == 64 ? 2 : 1) * c64_spi(p); }   // < 0: -(rows per image)

template <typename T, int CB, int NCG, bool STATS, bool ADD>
static int c64_launch(const C64P& q, int grid, hipStream_t st) {
    constexpr int SW = 64 * (4 / NCG), NPIECE = (SW + 4 + 7) / 8;
    const int lds = kC64Slots * CB * NPIECE * 1024 + 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel<T, CB, NCG, STATS, ADD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_c64_kernel<T, CB, NCG, STATS, ADD>), dim3((unsigned)grid, (unsigned)((q.N + 127) / 128)), dim3(256), lds, st, q);
    MRFP_LAUNCH_CHECK();
    return 0;
}
template <typename T, int CB, int NCG>
static int c64_pick(const ConvP& p, const C64P& q, int grid, hipStream_t st) {
    if (p.colstats) return c64_launch<T, CB, NCG, true, false>(q, grid, st);
    if (p.addend) return c64_launch<T, CB, NCG, false, true>(q, grid, st);
    return c64_launch<T, CB, NCG, false, false>(q, grid, st);
}

template <typename T>
static int c64_run_t(const ConvP& p, hipStream_t st) {
    C64P q;
    q.x = p.x; q.w = p.w; q.y = p.y; q.bias = p.bias; q.addend = p.addend; q.colstats = p.colstats; q.rowweight = p.rowweight;
    q.B = p.B; q.H = p.H; q.W = p.W; q.N = p.N; q.ldy = p.ldy; q.dil = p.dil;
    const int SW = p.N == 64 ? 128 : 64;
    q.strips = (p.W + SW - 1) / SW;
    q.units = p.B * q.strips * p.H;
    q.spi = c64_spi(p);
    q.xbytes = p.xbytes; q.wbytes = p.wbytes; q.ybytes = (unsigned)((int64_t)p.M * p.ldy * 2);
    const int grid = c64_grid(p);
    // the statistics rows the caller sized through c64_stats_blocks() hold spi slots per image and sub-strip: the longest run of
    // workgroups whose spans touch one image must fit (spans are units / grid or one more long)
    const int64_t upi = (int64_t)q.strips * p.H, lmin = q.units / grid;
    MRFP_CHECK(grid >= 1 && lmin >= 1 && upi / lmin + 2 <= q.spi && (int64_t)p.B * (p.N == 64 ? 2 : 1) * q.spi == c64_stats_blocks(p),
               "conv_c64: grid %d / %d statistics slots per image do not match the workspace rule", grid, q.spi);
    if (p.C == 64) return p.N == 64 ? c64_pick<T, 1, 2>(p, q, grid, st) : c64_pick<T, 1, 4>(p, q, grid, st);
    return p.N == 64 ? c64_pick<T, 2, 2>(p, q, grid, st) : c64_pick<T, 2, 4>(p, q, grid, st);
}
int c64_run(const ConvP& p, bool is_f16, hipStream_t st) { return is_f16 ? c64_run_t<f16>(p, st) : c64_run_t<bf16>(p, st); }

}  // namespace mrfp
