#include "conv_common.hpp"

// =============================================================================================
// wgrad:  dW[n, (r,s,c)] = sum_m dY[m, n] * X[pix(m; r,s), c]          (reduction over pixels)
//
// GEMM with M' = output channels, N' = R*S*C, K' = B*Ho*Wo.  Both operands are stored with the
// reduction index (the pixel) as the SLOW dimension, i.e. they are "k-strided": the LDS tiles
// keep the natural [pixel][channel] layout (filled with 16-byte loads along the channels) and the
// MFMA fragments are formed with the gfx950 transposing LDS read ds_read_b64_tr_b16 (bf16) or
// with plain strided ds_read_b32 (fp32).  Row pitch = row bytes + 64 so that the four pixel rows
// of one transposed read fall into four disjoint 16-bank windows.
// K' is split over gridDim.y workgroups; every split writes an fp32 slab, a second kernel sums
// the slabs in a fixed order (bitwise reproducible) and emits the OIHW fp32 gradient.
// =============================================================================================
namespace mrfp {

MRFP_STAMP_DECL(g_stamps_wgrad)
int stamps_wgrad(unsigned long long* out, int n) { return MRFP_STAMP_READ(g_stamps_wgrad, out, n); }

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((address_space(3))) short4v lds_short4v;

struct FastDiv {     // exact n / d for 0 <= n < 2^31:  q = (n * m) >> (31 + l),  m = floor(2^(31+l)/d) + 1
    unsigned m, sh;
};
static FastDiv make_fastdiv(unsigned d) {
    unsigned l = 0;
    while ((1u << l) < d) ++l;
    FastDiv f;
    f.m = (unsigned)(((1ull << (31 + l)) / d) + 1);
    f.sh = 31 + l;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) { return (int)(((unsigned long long)(unsigned)n * f.m) >> f.sh); }

struct WgP {
    const char* x;    // [B,H,W,C]
    const char* dy;   // [M][ldn]
    float* slab;      // [splits][N][Q]
    int B, H, W, C;
    int N, ldn;       // logical output channels, physical pitch of dy (elements)
    int R, S, Ho, Wo, stride, pad_h, pad_w, dil;
    int M, Q;         // pixels, R*S*C
    int klen;         // pixels per split (multiple of the K' tile)
    int tiles;        // output tiles per split
    int splits;       // K' splits per problem
    int ngroup;       // problems in this launch (1: x / dy / slab above; > 1: WgGroup below, slabs back to back)
    unsigned xbytes, dybytes;
    FastDiv div_hw, div_w;   // by Ho*Wo and by Wo
};

// GROUPED launch: up to kWgMaxGroup weight-gradient problems of ONE geometry (the repeated blocks of a ResNet stage:
// reference network/Resnet.py:579-585 _make_layer) in one grid.  The operand pointers travel in the kernel arguments (no
// device-side table, no copy, graph-capturable); the work index is problem-major, so with the XCD-chunked order the tiles of
// one problem share one L2.  22x the tiles of a layer-3 launch means 2-3 K' splits instead of 21-32: K' loops of ~190 tiles
// per workgroup instead of 12 behind the same prologue, and 1/10 of the fp32 slab traffic.
constexpr int kWgMaxGroup = 32;
struct WgGroup {
    const char* x[kWgMaxGroup];
    const char* dy[kWgMaxGroup];
};
struct WgOut {
    float* dw[kWgMaxGroup];
};

template <typename T> struct WgFrag;
template <> struct WgFrag<bf16> {
    // 32(rows along the lane) x 16(k) operand from a [pixel][channel] tile; col0 = first channel of
    // the 32-column block, krow0 = first pixel row of this 16-deep k step
    static __device__ __forceinline__ uint4 read(const char* tile, int pitch, int col0, int krow0, int lane) {
        const int g = lane >> 4;
        const int col = col0 + 16 * (g & 1) + 4 * (lane & 3);
        const int row = krow0 + 8 * (g >> 1) + ((lane & 15) >> 2);
        const char* p0 = tile + row * pitch + col * 2;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v*)(p0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v*)(p0 + 4 * pitch));
        uint4 r;
        r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
        r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
        r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
        r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
        return r;
    }
    static constexpr int KSTEP = 16;   // pixels consumed per Mma<bf16>::run
};
template <> struct WgFrag<f16> : WgFrag<bf16> {};     // same 16-bit transposing LDS read

// LDS-DMA tile layout of the 16-bit wgrad operands: rows of NB 64-byte blocks with NO padding (a DMA piece is 1 KiB of
// contiguous LDS); block lb of row r sits at physical block lb ^ key(r), key = r & 3 (NB >= 4) or (r >> 1) & 1 (NB == 2),
// so that the four pixel rows of one transposing read still fall into four disjoint 64-byte bank windows.
template <int NB> __device__ __forceinline__ int wg_key(int row) { return NB >= 4 ? (row & 3) : NB == 2 ? ((row >> 1) & 1) : 0; }
template <int NB>
__device__ __forceinline__ uint4 wg_read_sw(const char* tile, int col0, int krow0, int lane) {
    const int g = lane >> 4;
    const int row = krow0 + 8 * (g >> 1) + ((lane & 15) >> 2);
    const char* p0 = tile + row * (NB * 64) + (((col0 >> 5) ^ wg_key<NB>(row)) << 6) + 32 * (g & 1) + 8 * (lane & 3);
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v*)(p0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v*)(p0 + 4 * NB * 64));   // row + 4: same key
    uint4 r;
    r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
    r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
    r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
    r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
    return r;
}
template <> struct WgFrag<float> {
    // 4 MFMA 32x32x2 per call: element j of lane-half h is pixel krow0 + 2*j + h
    static __device__ __forceinline__ uint4 read(const char* tile, int pitch, int col0, int krow0, int lane) {
        const int c = col0 + (lane & 31), h = lane >> 5;
        uint4 r;
        r.x = *reinterpret_cast<const unsigned*>(tile + (krow0 + 0 + h) * pitch + c * 4);
        r.y = *reinterpret_cast<const unsigned*>(tile + (krow0 + 2 + h) * pitch + c * 4);
        r.z = *reinterpret_cast<const unsigned*>(tile + (krow0 + 4 + h) * pitch + c * 4);
        r.w = *reinterpret_cast<const unsigned*>(tile + (krow0 + 6 + h) * pitch + c * 4);
        return r;
    }
    static constexpr int KSTEP = 8;
};

#ifndef MRFP_WGRAD_HOLD
#define MRFP_WGRAD_HOLD 2      // k steps (of 4 per K' tile) multiplied after the next tile's transfer has been issued
#endif
#ifndef MRFP_WGRAD_HOLD_BIG
#define MRFP_WGRAD_HOLD_BIG 2  // the same for the 256 x 128 tile
#endif
template <typename T> struct WgTile { static constexpr int BKP = 64; };   // pixels per K' tile
template <> struct WgTile<float> { static constexpr int BKP = 32; };

// DENSE: pointwise convolution (1x1, stride 1, no padding): X is a dense [pixel][channel] matrix like dY, so its slots advance
// by a constant and need no (ih, iw) bookkeeping -- 60 of the 85 VALU and 40 of the 64 SALU instructions of a K' tile in
// the general kernel, on layers (M = 36 864 bottleneck 1x1) that are bound by exactly that instruction stream
// (profiles/r02_experiments.md section 4: 31 us with or without any global traffic, MFMA time 11 us).
// TMB: 32-row accumulator blocks per wave along the output channels.  2 = the 128 x 128 workgroup tile (64 FLOP per fill byte, four
// workgroups per CU); 4 = a 256 x 128 tile (85 FLOP per fill byte, 6 fragment reads per 8 MFMAs instead of 4 per 4; 128 accumulator
// registers, two workgroups per CU) for the layers with N % 256 == 0 and long K' loops -- the grouped launches above all.
template <typename T, int WM, int WN, bool DMA, bool DENSE = false, int TMB = 2>
__global__ __launch_bounds__(256, (TMB == 4 ? 2 : DMA ? 4 : 3)) void conv_wgrad_kernel(WgP p, WgGroup grp) {   // 2nd = waves per SIMD
    static_assert(WM * WN == 4, "4 waves");
    static_assert(!DMA || sizeof(T) == 2, "LDS-DMA layout is for the 16-bit types");
    static_assert(TMB == 2 || (TMB == 4 && DMA && WM == 2), "the 256 x 128 tile is an LDS-DMA kernel");
    constexpr int BKP = WgTile<T>::BKP;
    constexpr int EPC = 16 / (int)sizeof(T);                 // elements per 16-byte chunk
    constexpr int BNN = 32 * TMB * WM;                       // output channels per workgroup tile
    constexpr int CY = BNN / EPC, CX = 64 * WN / EPC;        // chunks per tile row
    constexpr int SY = BKP * CY / 256, SX = BKP * CX / 256;  // slots per thread
    constexpr int PY = BNN * (int)sizeof(T) + (DMA ? 0 : 64), PX = 64 * WN * (int)sizeof(T) + (DMA ? 0 : 64);   // row pitches
    constexpr int NBY = BNN * (int)sizeof(T) / 64, NBX = 2 * WN;    // 64-byte blocks per row (DMA layout)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ty = smem;               // single LDS buffer: the next tile waits in registers
    char* const tx = smem + BKP * PY;
    MRFP_STAMP_BEGIN();

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    typedef __attribute__((address_space(3))) void lds_void;
    const int wave1k = __builtin_amdgcn_readfirstlane(wave) * 1024;      // this wave's first DMA piece (1 KiB each)
    const int ntq = (p.Q + 64 * WN - 1) / (64 * WN);
    // 1-D grid over (split, tile), split-major, dealt to the XCDs in contiguous chunks: the tiles of one split read the
    // same pixel range of x and dy, so they share one L2 instead of pulling those rows into all eight
    int work = xcd_remap(blockIdx.x, gridDim.x);
    const char* xbase = p.x;
    const char* dybase = p.dy;
    int prob = 0;
    if (p.ngroup > 1) {              // (uniform) problem-major work order
        const int per = p.tiles * p.splits;
        prob = work / per;
        work -= prob * per;
        xbase = grp.x[prob];
        dybase = grp.dy[prob];
    }
    const int split = work / p.tiles, tile = work - split * p.tiles;
    const int n0 = (tile / ntq) * BNN, q0 = (tile % ntq) * 64 * WN;
    const int kbeg = split * p.klen;
    const int kend = min(p.M, kbeg + p.klen);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)dybase, 0, (int)p.dybytes, 0x00020000);

    // dY slots: dense rows; column fixed per thread
    const int yrow = t / CY;
    // DMA: this thread's LDS slot is fixed (piece base + lane * 16); the SOURCE chunk it fetches is the swizzled one
    const int ychunk = DMA ? ((((t % CY) >> 2) ^ wg_key<NBY>(yrow)) << 2) | ((t % CY) & 3) : t % CY;
    const int yn = n0 + ychunk * EPC;
    const unsigned ycol = yn < p.ldn ? (unsigned)yn * (unsigned)sizeof(T) : kOOB;
    const unsigned yrowbytes = (unsigned)p.ldn * (unsigned)sizeof(T);
    // X slots: fixed tap / channel per thread; the pixel moves by one K' tile per trip.  Its source coordinates
    // (ih, iw) and byte offset are advanced incrementally with adds / selects only (no multiply, no divide).
    const int xrow = t / CX;
    const int xchunk = DMA ? ((((t % CX) >> 2) ^ wg_key<NBX>(xrow)) << 2) | ((t % CX) & 3) : t % CX;
    const int q = q0 + xchunk * EPC;
    const int rs = q / p.C, c = q - rs * p.C;
    const int r = rs / p.S, s = rs - r * p.S;
    const int dh = r * p.dil - p.pad_h, dw = s * p.dil - p.pad_w;
    const bool xcol_ok = q < p.Q;
    const int pixbytes = p.C * (int)sizeof(T);
    const int st = p.stride;
    const int qh = BKP / p.Wo, rw = BKP - qh * p.Wo;                       // one K' tile = qh rows + rw pixels
    const int d_iw = rw * st, d_ih = qh * st;
    const unsigned D0 = (unsigned)((d_ih * p.W + d_iw) * pixbytes);        // plain advance
    const unsigned D1 = (unsigned)((st * p.W - p.Wo * st) * pixbytes);     // output-row wrap
    const unsigned D2 = (unsigned)((p.H * p.W - p.Ho * st * p.W) * pixbytes);   // image wrap
    const int iw_lim = p.Wo * st + dw, ih_lim = p.Ho * st + dh, WoSt = p.Wo * st, HoSt = p.Ho * st;
    int x_ih[SX], x_iw[SX];
    unsigned x_off[SX];
#pragma unroll
    for (int i = 0; i < SX; ++i) {
        const int m = kbeg + xrow + i * (256 / CX);
        if constexpr (DENSE) {
            x_ih[i] = 0;
            x_iw[i] = 0;
            x_off[i] = (unsigned)m * (unsigned)pixbytes + (unsigned)c * (unsigned)sizeof(T);
        } else {
            const int b = m / (p.Ho * p.Wo), rem = m - b * (p.Ho * p.Wo);
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            x_ih[i] = oh * st + dh;
            x_iw[i] = ow * st + dw;
            x_off[i] = (unsigned)((b * p.H + x_ih[i]) * p.W + x_iw[i]) * (unsigned)pixbytes + (unsigned)c * (unsigned)sizeof(T);
        }
    }
    const unsigned x_step = (unsigned)BKP * (unsigned)pixbytes;
    unsigned y_off[SY];
#pragma unroll
    for (int i = 0; i < SY; ++i) y_off[i] = (unsigned)(kbeg + yrow + i * (256 / CY)) * yrowbytes + ycol;
    const unsigned y_step = (unsigned)BKP * yrowbytes;

    auto load_tile = [&](int k0, uint4 (&ry)[SY], uint4 (&rx)[SX]) {
#pragma unroll
        for (int i = 0; i < SY; ++i) {
            const int m = k0 + yrow + i * (256 / CY);
            const unsigned voff = (m < kend && ycol < kOOB) ? y_off[i] : kOOB;
            if (DMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void*)(ty + (wave1k + i * 4096)), 16, (int)voff, 0, 0, 0);
            else
                ry[i] = bload(yr, voff);
            y_off[i] += y_step;
        }
#pragma unroll
        for (int i = 0; i < SX; ++i) {
            const int m = k0 + xrow + i * (256 / CX);
            const bool ok = DENSE ? (xcol_ok && m < kend)
                                  : (xcol_ok && m < kend && (unsigned)x_ih[i] < (unsigned)p.H && (unsigned)x_iw[i] < (unsigned)p.W);
            if (DMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(tx + (wave1k + i * 4096)), 16,
                                                         (int)(ok ? x_off[i] : kOOB), 0, 0, 0);
            else
                rx[i] = bload(xr, ok ? x_off[i] : kOOB);
            // advance this slot by one K' tile
            if constexpr (DENSE) {
                x_off[i] += x_step;
            } else {
                x_iw[i] += d_iw;
                x_ih[i] += d_ih;
                x_off[i] += D0;
                if (x_iw[i] >= iw_lim) { x_iw[i] -= WoSt; x_ih[i] += st; x_off[i] += D1; }
                while (x_ih[i] >= ih_lim) { x_ih[i] -= HoSt; x_off[i] += D2; }
            }
        }
    };
    auto store_tile = [&](const uint4 (&ry)[SY], const uint4 (&rx)[SX]) {
#pragma unroll
        for (int i = 0; i < SY; ++i) *reinterpret_cast<uint4*>(ty + (yrow + i * (256 / CY)) * PY + ychunk * 16) = ry[i];
#pragma unroll
        for (int i = 0; i < SX; ++i) *reinterpret_cast<uint4*>(tx + (xrow + i * (256 / CX)) * PX + xchunk * 16) = rx[i];
    };

    f32x16 acc[TMB][2];
#pragma unroll
    for (int i = 0; i < TMB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = (kend - kbeg + BKP - 1) / BKP;
    // one K' tile of register prefetch (a second register set was measured: it costs a wave of occupancy and
    // runs 35 % slower -- three co-resident workgroups per CU hide the load latency better)
    uint4 ry[SY], rx[SX];
    if (DMA) {
        // single LDS buffer filled by LDS-DMA (as the forward kernel, mode 3): no staging registers, no ds_write
        // EARLY ISSUE (as in conv_igemm_kernel): the fragments of the last HOLD k steps go to registers, a barrier frees the
        // buffer, the next tile's transfer is issued and the held k steps are multiplied inside its latency.
        constexpr int KSN = BKP / 16;
        constexpr int HOLD = (!DENSE && WM == 1) ? 1 : TMB == 4 ? MRFP_WGRAD_HOLD_BIG : MRFP_WGRAD_HOLD;     // (the 64x256 gather variant spills with two held k steps)
        if (nkt > 0) load_tile(kbeg, ry, rx);
        for (int kt = 0; kt < nkt; ++kt) {
            dma_wait<0>();            // explicit: across the back edge the compiler's own wait lands behind the barrier
            __syncthreads();          // the tile has landed everywhere
#pragma unroll
            for (int ks = 0; ks < KSN - HOLD; ++ks) {
                uint4 fa[TMB], fb[2];
#pragma unroll
                for (int i = 0; i < TMB; ++i) fa[i] = wg_read_sw<NBY>(ty, wm * 32 * TMB + i * 32, ks * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = wg_read_sw<NBX>(tx, wn * 64 + j * 32, ks * 16, lane);
#pragma unroll
                for (int i = 0; i < TMB; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
            }
            uint4 ha[HOLD > 0 ? HOLD : 1][TMB], hb[HOLD > 0 ? HOLD : 1][2];
#pragma unroll
            for (int h = 0; h < HOLD; ++h) {
#pragma unroll
                for (int i = 0; i < TMB; ++i) ha[h][i] = wg_read_sw<NBY>(ty, wm * 32 * TMB + i * 32, (KSN - HOLD + h) * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) hb[h][j] = wg_read_sw<NBX>(tx, wn * 64 + j * 32, (KSN - HOLD + h) * 16, lane);
            }
            __syncthreads();          // lgkmcnt(0) + barrier: everybody is done reading, the buffer is free
            if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * BKP, ry, rx);
#pragma unroll
            for (int h = 0; h < HOLD; ++h)
#pragma unroll
                for (int i = 0; i < TMB; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], ha[h][i], hb[h][j]);
        }
    } else {
    if (nkt > 0) {
        load_tile(kbeg, ry, rx);
        store_tile(ry, rx);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * BKP, ry, rx);
#pragma unroll
        for (int ks = 0; ks < BKP / WgFrag<T>::KSTEP; ++ks) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = WgFrag<T>::read(ty, PY, wm * 64 + i * 32, ks * WgFrag<T>::KSTEP, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = WgFrag<T>::read(tx, PX, wn * 64 + j * 32, ks * WgFrag<T>::KSTEP, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
        }
        __syncthreads();
        if (kt + 1 < nkt) store_tile(ry, rx);
        __syncthreads();
    }
    }

    MRFP_STAMP_END(g_stamps_wgrad);
    float* out = p.slab + ((size_t)prob * p.splits + split) * p.N * p.Q;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qq = q0 + wn * 64 + j * 32 + lr;
        if (qq >= p.Q) continue;
#pragma unroll
        for (int i = 0; i < TMB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 32 * TMB + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < p.N) out[(size_t)n * p.Q + qq] = acc[i][j][e];
            }
    }
}

// dW[n][c][r][s] (OIHW fp32, c < Ctrue) = sum_z slab[z][n][(r*S+s)*C + c]
// One workgroup = 64 consecutive float4 of the slab order (1 KB runs: the slabs are ~20-30x the size of dW, so their reads are
// the ones that must coalesce) x 4 slab groups: wave g sums the slabs z = g, g + 4, g + 8, ... in ascending order with eight
// independent 16-byte loads in flight per lane, the four group sums are combined through LDS as (g0 + g1) + (g2 + g3) -- a fixed
// association, so the result is bitwise reproducible.  (Round 2 walked all slabs of an output in ONE thread, four loads in
// flight: 14 us per launch for 30-60 MB that the chip reads in 6; a last-arriving-workgroup reduction inside conv_wgrad_kernel
// was considered and not built -- the arriver of a tile would pull splits x 64 KB through ONE compute unit behind a ~3.5 us
// device-scope fence, MI355X_MICROARCH.md -- DESIGN.md section 4.)  The OIHW stores are 4-byte scattered but few.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int N, int Q, int C, int Ctrue, int RS,
                                                            float* __restrict__ dw, int accumulate, WgOut outs) {
    __shared__ float4 part[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t total4 = (int64_t)N * Q / 4, NQ = (int64_t)N * Q;
    if (gridDim.y > 1) {             // grouped launch: problem blockIdx.y, its slabs back to back
        slab += (size_t)blockIdx.y * splits * NQ;
        dw = outs.dw[blockIdx.y];
    }
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4) {
        const float* base = slab + i * 4;
        int z = g;
        for (; z + 28 < splits; z += 32) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (size_t)(z + 4 * u) * NQ);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; z < splits; z += 4) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)z * NQ);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[g][o] = acc;
    __syncthreads();
    if (g != 0 || i >= total4) return;
    const float4 p0 = part[0][o], p1 = part[1][o], p2 = part[2][o], p3 = part[3][o];
    acc.x = (p0.x + p1.x) + (p2.x + p3.x);
    acc.y = (p0.y + p1.y) + (p2.y + p3.y);
    acc.z = (p0.z + p1.z) + (p2.z + p3.z);
    acc.w = (p0.w + p1.w) + (p2.w + p3.w);
    const int64_t j = i * 4;
    const int n = (int)(j / Q), q = (int)(j - (int64_t)n * Q);
    const int rs = q / C, c = q - rs * C;
    if (c >= Ctrue) return;
    float* d = dw + ((size_t)n * Ctrue + c) * RS + rs;
    if (accumulate) {            // a later batch range of an activation that is read in several launches
        acc.x += d[0];
        if (c + 1 < Ctrue) acc.y += d[RS];
        if (c + 2 < Ctrue) acc.z += d[2 * RS];
        if (c + 3 < Ctrue) acc.w += d[3 * RS];
    }
    d[0] = acc.x;
    if (c + 1 < Ctrue) d[RS] = acc.y;
    if (c + 2 < Ctrue) d[2 * RS] = acc.z;
    if (c + 3 < Ctrue) d[3 * RS] = acc.w;
}

template <typename T, int WM, int WN, bool DMA, bool DENSE = false, int TMB = 2>
static int launch_wgrad_v(const WgP& p, int splits, hipStream_t st, const WgGroup* grp) {
    constexpr int BNN = 32 * TMB * WM;
    constexpr int PY = BNN * (int)sizeof(T) + (DMA ? 0 : 64), PX = 64 * WN * (int)sizeof(T) + (DMA ? 0 : 64);
    // MRFP_WGRAD_LDS=<bytes> (experiments: tools/overlap_micro.py): ask for at least that much LDS per workgroup, i.e. cap the
    // workgroups per CU below what the registers allow (81920: one per CU) -- how much of a concurrent HBM-bound kernel's time
    // a weight-gradient launch can hide in when it leaves register file and wave slots free
    static int lds_min = -1;
    if (lds_min < 0) { const char* e = getenv("MRFP_WGRAD_LDS"); lds_min = e ? atoi(e) : 0; }
    const int lds_need = WgTile<T>::BKP * (PY + PX);
    const int lds = lds_need > lds_min ? lds_need : lds_min;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, WM, WN, DMA, DENSE, TMB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    WgP q = p;
    q.tiles = ((p.N + BNN - 1) / BNN) * ((p.Q + 64 * WN - 1) / (64 * WN));
    q.splits = splits;
    hipLaunchKernelGGL((conv_wgrad_kernel<T, WM, WN, DMA, DENSE, TMB>), dim3((unsigned)(q.tiles * splits * q.ngroup)), dim3(256), lds, st, q, *grp);
    MRFP_LAUNCH_CHECK();
    return 0;
}

// MRFP_WGRAD_DMA=0 keeps register staging for the 16-bit types (A/B measurements); fp32 always stages in registers
template <typename T, int WM, int WN>
static int launch_wgrad(const WgP& p, int splits, hipStream_t st, const WgGroup* grp, int tmb = 2) {
    static int dma = -1;
    if (dma < 0) { const char* e = getenv("MRFP_WGRAD_DMA"); dma = e ? atoi(e) : 1; }
    if (sizeof(T) == 2 && dma) {
        static int dense = -1;
        if (dense < 0) { const char* e = getenv("MRFP_WGRAD_DENSE"); dense = e ? atoi(e) : 1; }
        const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 && p.H == p.Ho && p.W == p.Wo;
        if constexpr (WM == 2 && sizeof(T) == 2) {
            if (tmb == 4) {
                if (dense && pointwise) return launch_wgrad_v<T, WM, WN, true, true, 4>(p, splits, st, grp);
                return launch_wgrad_v<T, WM, WN, true, false, 4>(p, splits, st, grp);
            }
        }
        if (dense && pointwise) return launch_wgrad_v<T, WM, WN, sizeof(T) == 2, true>(p, splits, st, grp);
        return launch_wgrad_v<T, WM, WN, sizeof(T) == 2>(p, splits, st, grp);
    }
    return launch_wgrad_v<T, WM, WN, false>(p, splits, st, grp);
}

// MRFP_WGRAD_BIG: 0 = never the 256 x 128 tile, 1 (default) = where the cost model prefers it, 2 = wherever it is legal (A/B runs)
static int wgrad_big_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("MRFP_WGRAD_BIG");
        mode = e ? atoi(e) : 1;
        const char* d = getenv("MRFP_WGRAD_DMA");      // (=0: the register-staged A/B path of the 16-bit kernels has no 256 x 128 instance)
        if (d && atoi(d) == 0) mode = 0;
    }
    return mode;
}

// Split count from a small cost model (times in us, constants fitted to the bench workload's per-launch timings):
//   a CU that holds w = ceil(tiles*sp/256) workgroups needs w * (K' tiles per split) tile-steps of `step` us, divided
//   by a latency-hiding efficiency (fewer co-resident workgroups than the kernel's occupancy hide less of each other's fill
//   latency); every split adds an fp32 slab of dW that is written once and read once by the reduction (~3 TB/s).
// MRFP_WGRAD_WGS=<n> replaces the model by "about n workgroups" (A/B measurements).
static double wgrad_cost(int64_t tiles, int64_t nkt, int64_t group, double nq, double step, int occ, int64_t& sp_out) {
    static int target = -1;
    if (target < 0) {
        const char* e = getenv("MRFP_WGRAD_WGS");
        target = e ? atoi(e) : 0;
    }
    int64_t sp = 1;
    double best = 1e30;
    if (target > 0) {
        sp = target / tiles;
    } else {
        int64_t smax = 1024 / tiles > 96 ? 1024 / tiles : 96;      // few tiles: enough splits to fill the chip
        if (smax > nkt) smax = nkt;
        for (int64_t c = 1; c <= smax; ++c) {
            const int64_t w = (tiles * c + 255) / 256, iters = (nkt + c - 1) / c;
            const double eff = occ == 4 ? (w >= 3 ? 1.0 : w == 2 ? 0.85 : 0.6) : (w >= 2 ? 1.0 : 0.7);
            const double cost = (double)w * (double)iters * step / eff + (double)c * (double)group * (nq * 8.0 / 3.0e6);
            if (cost < best * 0.999) { best = cost; sp = c; }
        }
    }
    sp_out = sp;
    return best;
}

// wm: wave rows (1: the 64 x 256 tile for N <= 64; 2: 128 x 128 or, tmb = 4, 256 x 128); splits / klen: K' splits and pixels per split
static void wgrad_plan(int64_t M, int64_t N, int64_t Q, int bkp, int& wm, int& splits, int& klen, int64_t cap = 0, int64_t group = 1,
                       int* tmb_out = nullptr, bool pointwise = false) {
    wm = N <= 64 ? 1 : 2;
    const int wn = 4 / wm;
    // (a grouped launch: `group` problems of this geometry fill the chip together; the split count is per problem)
    const int64_t tiles = group * ((N + 64 * wm - 1) / (64 * wm)) * ((Q + 64 * wn - 1) / (64 * wn));
    const int64_t nkt = (M + bkp - 1) / bkp;
    int64_t sp = 1;
    (void)wgrad_cost(tiles, nkt, group, (double)N * (double)Q, 0.84, 4, sp);
    int tmb = 2;
    // The 256 x 128 tile: 16-bit LDS-DMA kernels (K' tile of 64 pixels), whole 256-channel tiles.  Two workgroups per CU hide less
    // fill latency than four, so it needs LONG K' loops to pay -- measured per launch class of the bench step (tools/wgrad_micro.py,
    // MRFP_WGRAD_BIG=0 / 2, same box): the grouped layer-3 pointwise launches 28.7 -> 26.0 us per problem, 512 <-> 2048 grouped
    // 88 -> 83-86, the 3x3 layers at 192^2 697 -> 634 and 890 -> 802 us; but single pointwise launches with few tiles 37 -> 44 us
    // (every split gets ~10 K' tiles), and the 3x3 layers at 48^2 47.2 -> 49.2 us per problem even grouped.  Rule: enough K' tile-steps
    // per CU (`load`, in 256 x 128 units), and for the gather (non-pointwise) form only the large images.
    if (bkp == 64 && wm == 2 && N % 256 == 0 && wgrad_big_mode() > 0) {
        const int64_t tiles4 = group * (N / 256) * ((Q + 127) / 128);
        const int64_t load = tiles4 * nkt / 256;
        const bool want = load >= 100 && (pointwise || M >= 65536);
        if (wgrad_big_mode() >= 2 || want) {
            int64_t sp4 = 1;
            (void)wgrad_cost(tiles4, nkt, group, (double)N * (double)Q, 1.30, 2, sp4);
            tmb = 4;
            sp = sp4;
        }
    }
    if (tmb_out) *tmb_out = tmb;
    if (sp < 1) sp = 1;
    if (sp > nkt) sp = nkt;
    if (cap > 0 && sp > cap) sp = cap;         // (a batch range of a larger call: the workspace was sized for the whole call)
    int64_t per = (nkt + sp - 1) / sp;        // K' tiles per split
    sp = (nkt + per - 1) / per;
    splits = (int)sp;
    klen = (int)(per * bkp);
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int64_t mrfp_conv_wgrad_ws_bytes(int64_t M, int64_t N, int64_t Q) {
    int wm, s32, s64, klen;
    int s64p;
    mrfp::wgrad_plan(M, N, Q, 32, wm, s32, klen);         // fp32 K' tile
    mrfp::wgrad_plan(M, N, Q, 64, wm, s64, klen);         // bf16 K' tile
    mrfp::wgrad_plan(M, N, Q, 64, wm, s64p, klen, 0, 1, nullptr, true);      // ... of a pointwise launch (its tile rule differs)
    if (s64p > s64) s64 = s64p;
    const int64_t s3 = mrfp::wg3_splits_bound(N, Q, 1);       // the accumulator-stationary 3x3 kernel (conv_wg3.hip), if Q could be 9 * C
    if (s3 > s64) s64 = (int)s3;
    const int64_t s1 = mrfp::wg1_splits_bound(N, Q, 1);       // ... and the pointwise one (conv_wg1.hip)
    if (s1 > s64) s64 = (int)s1;
    return (int64_t)(s32 > s64 ? s32 : s64) * N * Q * 4;
}

static int wgrad_run(const void* const* xs, const void* const* dys, float* const* dws, int64_t count, void* ws, int dtype, int64_t B,
                     int64_t H, int64_t W, int64_t C, int64_t Ctrue, int64_t N, int64_t ldn, int64_t R, int64_t S, int64_t Ho,
                     int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, void* stream) {
    MRFP_CHECK(xs && dys && dws && ws && count > 0 && count <= kWgMaxGroup && B > 0 && H > 0 && W > 0 && C > 0 && N > 0 && R > 0 && S > 0 &&
               Ho > 0 && Wo > 0, "conv_wgrad: bad arguments");
    MRFP_CHECK(dtype == MRFP_F32 || dtype == MRFP_BF16 || dtype == MRFP_F16, "conv_wgrad: unknown dtype %d", dtype);
    const int esz = dtype == MRFP_F32 ? 4 : 2;
    MRFP_CHECK((C * esz) % 16 == 0 && (ldn * esz) % 16 == 0 && ldn >= N && Ctrue <= C,
               "conv_wgrad: channel counts must make 16-byte chunks (C=%lld ldn=%lld)", (long long)C, (long long)ldn);
    for (int64_t g = 0; g < count; ++g)
        MRFP_CHECK(xs[g] && dys[g] && dws[g] && aligned16(xs[g]) && aligned16(dys[g]), "conv_wgrad: x / dy must be 16-byte aligned, dw non-null");
    MRFP_CHECK(B * Ho * Wo < (1LL << 31), "conv_wgrad: tensor too large");
    WgP p;
    p.x = (const char*)xs[0]; p.dy = (const char*)dys[0]; p.slab = (float*)ws;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.N = (int)N; p.ldn = (int)ldn;
    p.R = (int)R; p.S = (int)S; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.stride = (int)stride; p.pad_h = (int)pad_h; p.pad_w = (int)pad_w; p.dil = (int)dil;
    p.Q = (int)(R * S * C);
    p.ngroup = (int)count;
    WgGroup grp;
    WgOut outs;
    for (int g = 0; g < kWgMaxGroup; ++g) {
        grp.x[g] = (const char*)xs[g < count ? g : 0];
        grp.dy[g] = (const char*)dys[g < count ? g : 0];
        outs.dw[g] = dws[g < count ? g : 0];
    }
    // Both operands are read through 32-bit buffer-descriptor offsets: an activation above kOOB bytes is walked in batch
    // ranges, every range one wgrad + reduction pair on the stream, the later ones adding to dw (fixed order: reproducible)
    const int64_t ximg = H * W * C * esz, yimg = Ho * Wo * ldn * esz;
    MRFP_CHECK(ximg < (int64_t)kOOB && yimg < (int64_t)kOOB, "conv_wgrad: one image exceeds the 3.75 GB buffer-descriptor range");
    int64_t bmax = (int64_t)(kOOB - 1) / (ximg > yimg ? ximg : yimg);
    MRFP_CHECK(count == 1 || bmax >= B, "conv_wgrad_grouped: the activations of a grouped launch must fit one 3.75 GB buffer range each");
    int dbg_drop = 0;
    {   // timing-only diagnostics (see mrfp_conv_fwd)
        static int dbg = -1;
        if (dbg < 0) { const char* e = getenv("MRFP_DEBUG_DROP"); dbg = e ? atoi(e) : 0; }
        dbg_drop = dbg;
    }
    p.div_hw = make_fastdiv((unsigned)(Ho * Wo)); p.div_w = make_fastdiv((unsigned)Wo);
    hipStream_t st = (hipStream_t)stream;
    int wm0, cap, klen0;
    const bool pointwise = R == 1 && S == 1 && stride == 1 && pad_h == 0 && pad_w == 0 && H == Ho && W == Wo;
    wgrad_plan(B * Ho * Wo, N, p.Q, dtype == MRFP_F32 ? 32 : 64, wm0, cap, klen0, 0, count, nullptr, pointwise);     // what `ws` was sized for
    for (int64_t b0 = 0; b0 < B; b0 += bmax) {
        const int64_t bc = B - b0 < bmax ? B - b0 : bmax;
        p.B = (int)bc;
        p.M = (int)(bc * Ho * Wo);
        p.x = (const char*)xs[0] + b0 * ximg;
        p.dy = (const char*)dys[0] + b0 * yimg;
        p.xbytes = (dbg_drop & 1) ? 0u : (unsigned)(bc * ximg);
        p.dybytes = (dbg_drop & 2) ? 0u : (unsigned)(bc * yimg);
        int wm, splits, tmb = 2;
        wgrad_plan(p.M, N, p.Q, dtype == MRFP_F32 ? 32 : 64, wm, splits, p.klen, cap, count, &tmb, pointwise);
        int rc;
        if (bc == B && !dbg_drop && wg3_applicable(esz, B, H, W, C, N, ldn, R, S, Ho, Wo, stride, pad_h, pad_w, dil, count))
            rc = wg3_run(xs, dys, count, (float*)ws, dtype == MRFP_F16, B, H, W, C, N, ldn, dil, p.xbytes, p.dybytes, &splits, st);
        else if (bc == B && !dbg_drop && wg1_applicable(esz, B, H, W, C, N, ldn, R, S, Ho, Wo, stride, pad_h, pad_w, count))
            rc = wg1_run(xs, dys, count, (float*)ws, dtype == MRFP_F16, B * H * W, C, N, ldn, p.xbytes, p.dybytes, &splits, st);
        else if (dtype == MRFP_F32) rc = wm == 1 ? launch_wgrad<float, 1, 4>(p, splits, st, &grp) : launch_wgrad<float, 2, 2>(p, splits, st, &grp);
        else if (dtype == MRFP_F16) rc = wm == 1 ? launch_wgrad<f16, 1, 4>(p, splits, st, &grp) : launch_wgrad<f16, 2, 2>(p, splits, st, &grp, tmb);
        else rc = wm == 1 ? launch_wgrad<bf16, 1, 4>(p, splits, st, &grp) : launch_wgrad<bf16, 2, 2>(p, splits, st, &grp, tmb);
        if (rc) return rc;
        const int64_t total4 = N * (int64_t)p.Q / 4;          // Q = R*S*C and C*esz % 16 == 0  =>  Q % 4 == 0
        const int64_t blocks = (total4 + 63) / 64;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, st, (const float*)ws, splits, (int)N,
                           p.Q, (int)C, (int)Ctrue, (int)(R * S), dws[0], b0 > 0 ? 1 : 0, outs);
        MRFP_LAUNCH_CHECK();
    }
    return 0;
}

int mrfp_conv_wgrad(const void* x, const void* dy, float* dw, void* ws, int dtype, int64_t B, int64_t H, int64_t W,
                    int64_t C, int64_t Ctrue, int64_t N, int64_t ldn, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                    int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, void* stream) {
    return wgrad_run(&x, &dy, &dw, 1, ws, dtype, B, H, W, C, Ctrue, N, ldn, R, S, Ho, Wo, stride, pad_h, pad_w, dil, stream);
}

int64_t mrfp_conv_wgrad_group_max(void) { return kWgMaxGroup; }

int64_t mrfp_conv_wgrad_grouped_ws_bytes(int64_t M, int64_t N, int64_t Q, int64_t count) {
    int wm, s32, s64, klen;
    int s64p;
    mrfp::wgrad_plan(M, N, Q, 32, wm, s32, klen, 0, count);
    mrfp::wgrad_plan(M, N, Q, 64, wm, s64, klen, 0, count);
    mrfp::wgrad_plan(M, N, Q, 64, wm, s64p, klen, 0, count, nullptr, true);
    if (s64p > s64) s64 = s64p;
    const int64_t s3 = mrfp::wg3_splits_bound(N, Q, count);
    if (s3 > s64) s64 = (int)s3;
    const int64_t s1 = mrfp::wg1_splits_bound(N, Q, count);
    if (s1 > s64) s64 = (int)s1;
    return (int64_t)(s32 > s64 ? s32 : s64) * count * N * Q * 4;
}

int mrfp_conv_wgrad_grouped(const void* const* xs, const void* const* dys, float* const* dws, int64_t count, void* ws, int dtype,
                            int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ctrue, int64_t N, int64_t ldn, int64_t R, int64_t S,
                            int64_t Ho, int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, void* stream) {
    return wgrad_run(xs, dys, dws, count, ws, dtype, B, H, W, C, Ctrue, N, ldn, R, S, Ho, Wo, stride, pad_h, pad_w, dil, stream);
}

}  // extern "C"
