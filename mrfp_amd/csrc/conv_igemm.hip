// conv_igemm.hip -- implicit-GEMM convolution on the gfx950 matrix cores (MFMA), NHWC activations: forward and dgrad.
// (Pointwise short-K layers: conv_pw.hip; weight gradients: conv_wgrad.hip; weight packs: conv_pack.hip.)
//
//   forward :  Y[m, n]  = sum_{r,s,c} X[pix(m; r,s), c] * Wf[n, (r,s,c)]         m = (b, oh, ow)
//   dgrad   :  the same kernel on dY with the flipped/transposed pack Wd[c, (r',s',n)] and a
//              "source stride" for strided convolutions (a tap exists only where the position
//              divides the stride)
//
// Tiling (one workgroup = WM x WN waves, every wave owns TM x TN MFMA 32x32 accumulators: 128x128, 192x128, 96x128 or
// 256x64 outputs per 4-wave workgroup, chosen per launch by run_igemm): the K dimension is walked in 128-BYTE steps
// (64 bf16 / 32 fp32 = 8 chunks of 16 bytes).  A chunk never straddles an (r,s) tap because the channel count is a
// multiple of the chunk, so every 16-byte global load is either a contiguous run of input channels of one pixel or zero
// (padding) -- im2col happens in the address computation, the matrix is never materialised.  Staging (launch_igemm):
// LDS-DMA (buffer_load ... lds) straight into ONE LDS buffer per workgroup -- no staging registers, no ds_write; the 3-5
// co-resident workgroups of a CU hide each other's fill latency.  LDS rows are 128 B with the 16-byte slot
// XOR-swizzled by (row>>1)&7 (applied on the SOURCE side for LDS-DMA) so that the ds_read_b128 fragment reads are
// bank-conflict free.
//
//   bf16 / f16: v_mfma_f32_16x16x32_{bf16,f16} (fp32 accumulate; 32x32x16 in the weight-gradient kernel)  -- bench dtype
//   fp32:       v_mfma_f32_32x32x2_f32   (exact fp32 FMA chains)                                        -- parity dtype
// All share the byte geometry, so there is one kernel template.
//
// Replaces (reference): every nn.Conv2d on the hot path -- Resnet.py:156-161 (Bottleneck),
// deepv3.py:96-112 (ASPP), 200-219 (decoder), 221-237 (HRFP), and their autograd backward.
#include "conv_common.hpp"
#include <type_traits>

// MRFP_WREG (build switch, default 0; round 6 experiment, profiles/r06_experiments.md): the WEIGHT fragments of the 16-bit ALIGNED
// kernels go global -> registers (one buffer_load_dwordx4 per 16x32 fragment, issued where the K tile's LDS-DMA is issued) instead of
// global -> LDS -> registers: the activation tile alone passes through LDS, so the LDS-DMA landing traffic and the ds_read_b128 count
// of a K tile both drop by BN / (BM + BN) (tools/fill_micro.hip: fragment reads beside the DMA halve the DMA's rate).
// bit 0: the plain single-buffer loop, bit 1: the row-reuse loop.
#ifndef MRFP_WREG
#define MRFP_WREG 0
#endif

namespace mrfp {

MRFP_STAMP_DECL(g_stamps_igemm)
int stamps_igemm(unsigned long long* out, int n) { return MRFP_STAMP_READ(g_stamps_igemm, out, n); }

template <typename T, int WM, int WN, bool ALIGNED, bool STRIDED, int TM, int TN, bool RR = false>
__global__ __launch_bounds__(64 * WM * WN, (((RR || ((MRFP_WREG & 1) && ALIGNED && sizeof(T) == 2)) && TM * TN >= 6) ? 2 : 3)) void conv_igemm_kernel(ConvP p) {   // 2nd = waves per SIMD
    constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int SA = BM * 8 / NT, SB = BN * 8 / NT, RSTEP = NT / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA0 = smem;
    char* const sB0 = smem + BM * 128;
    MRFP_STAMP_BEGIN();

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int rbase = t >> 3;
    const int chunk = (t & 7) ^ ((rbase >> 1) & 7);   // SOURCE chunk of this thread's slots (the LDS image of a DMA piece is lane-linear)
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 8;          // first tile row of this wave's DMA pieces
    const int pixbytes = p.C * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);

    // CLASSED (dgrad of a stride-2 convolution, host: conv_fwd_impl): an output pixel (oh, ow) only has the taps r = (pad_h + oh) mod 2
    // (+2, ...), s likewise -- a quarter of them on average, and which ones depends on the pixel's PARITY CLASS (oh & 1, ow & 1).  With
    // rows enumerated pixel by pixel every tile mixes the four classes and walks all R*S taps, zero-filling three quarters of its
    // fills and multiplies.  Here the row index is class-major -- m = class * Mc + ((b * Ho/2 + i) * Wo/2 + j), pixel (2i + ph, 2j + pw),
    // Mc a multiple of the tile height -- so a tile's rows share their taps: the K loop walks only those (none at all for three of
    // the four classes of a 1x1 stride-2 convolution: those tiles just store zeros + addend), and the epilogue scatters the rows back.
    // p.classed == 2 (pointwise stride-2 convolution: only the even-even pixels have a tap at all): the rows are the even-even pixels
    // alone (M = Mc), and the epilogue stores each row's 2 x 2 output block -- its value and three zeros (+ addend) -- so that the
    // output is written in two-pixel runs and no tile exists only to store zeros.
    const bool classed = STRIDED && p.classed != 0;
    const bool quad = STRIDED && p.classed == 2;
    const int cHc = p.Ho >> 1, cWc = p.Wo >> 1, cMc = p.B * cHc * cWc;
    const int cls = (classed && !quad) ? m0 / cMc : 0, cph = cls >> 1, cpw = cls & 1;
    auto pixel_of = [&](int m, int& b, int& oh, int& ow) {
        if (classed) {
            const int mm = m - cls * cMc;
            b = mm / (cHc * cWc);
            const int rem = mm - b * (cHc * cWc), i2 = rem / cWc;
            oh = 2 * i2 + cph;
            ow = 2 * (rem - i2 * cWc) + cpw;
        } else {
            b = m / (p.Ho * p.Wo);
            const int rem = m - b * (p.Ho * p.Wo);
            oh = rem / p.Wo;
            ow = rem - oh * p.Wo;
        }
    };
    // fixed per-thread gather state for its SA rows of the A tile
    int a_ih0[SA], a_iw0[SA];
    unsigned a_base[SA];   // !STRIDED: byte offset of pixel (b, ih0, iw0) (mod 2^32); STRIDED: offset of image b
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int m = m0 + rbase + i * RSTEP;
        if (m < p.M) {
            int b, oh, ow;
            pixel_of(m, b, oh, ow);
            a_ih0[i] = oh * p.stride - p.pad_h;
            a_iw0[i] = ow * p.stride - p.pad_w;
            a_base[i] = STRIDED ? (unsigned)(b * p.H * p.W) * (unsigned)pixbytes
                                : (unsigned)((b * p.H + a_ih0[i]) * p.W + a_iw0[i]) * (unsigned)pixbytes;
        } else {
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = -(1 << 28);
            a_base[i] = 0;
        }
    }
    unsigned b_base[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int n = n0 + rbase + i * RSTEP;
        b_base[i] = n < p.N ? (unsigned)n * (unsigned)p.kchunks * 16u : kOOB;
    }

    // tap tracking: scalar (r, s, tile-in-tap) when ALIGNED, per-thread (r, s, chunk-in-tap) otherwise
    // (CLASSED: the first tap of this tile's class along each axis, every second one after it)
    // (Round 5 tried skipping, per tile, the filter rows that lie outside the image for every row of the tile -- with a dilation of
    //  6 / 12 / 18 on a 48-row map a third of the ASPP tiles' K tiles are fills of zeros.  The three launches it helped gained 5 %
    //  (1 219 -> 1 289 TFLOP/s); the run-time loop bounds cost EVERY instance of this kernel 26 scalar and 4 - 10 vector registers and
    //  the 192x128 tile 60 bytes of scratch: 332^2 256 -> 128 0.90 -> 1.59 ms, conv family +3.3 ms.  Reverted; profiles/r05_experiments.md 10.)
    const int tr0 = classed ? ((p.pad_h + cph) & 1) : 0, ts0 = classed ? ((p.pad_w + cpw) & 1) : 0, tstep = classed ? 2 : 1;
    int tr = tr0, ts = ts0, tc = 0;
    if (!ALIGNED) {
        const int rs = chunk / p.cpr;
        tc = chunk - rs * p.cpr;
        tr = rs / p.S;
        ts = rs - tr * p.S;
    }

    typedef __attribute__((address_space(3))) void lds_void;
    constexpr bool M16 = kM16 && sizeof(T) == 2;
    constexpr bool WREG = (MRFP_WREG & 1) && M16 && ALIGNED && !RR;      // weight fragments through registers (plain loop)
    const int l15 = lane & 15, lq = lane >> 4;
    // WREG: byte offset of weight row n = (this wave's column block j) * 16 + l15 in the pack, + this lane quarter's chunk
    unsigned wq_base[2 * TN];
    uint4 fbn[2][2 * TN];          // the NEXT K tile's weight fragments [k step][column block], in flight beside its LDS-DMA
    if constexpr (WREG) {
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) {
            const int n = n0 + wn * 32 * TN + j * 16 + l15;
            wq_base[j] = n < p.N ? (unsigned)n * (unsigned)p.kchunks * 16u + (unsigned)lq * 16u : kOOB;
        }
    }
    auto load_tile = [&](int kt) {
        const int dh = tr * p.dil, dw = ts * p.dil;
        const int cc = ALIGNED ? tc * 8 + chunk : tc;
        const bool qok = ALIGNED ? true : tr < p.R;
        const unsigned tap = (unsigned)((dh * p.W + dw) * pixbytes + cc * 16);
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            int ih = a_ih0[i] + dh, iw = a_iw0[i] + dw;
            unsigned voff;
            if (STRIDED) {
                bool ok = qok && ih >= 0 && iw >= 0 && (ih % p.sstride == 0) && (iw % p.sstride == 0);
                ih /= p.sstride;
                iw /= p.sstride;
                ok = ok && ih < p.H && iw < p.W;
                voff = ok ? a_base[i] + (unsigned)((ih * p.W + iw) * pixbytes + cc * 16) : kOOB;
            } else {
                const bool ok = qok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                voff = ok ? a_base[i] + tap : kOOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(sA0 + (wrow + i * RSTEP) * 128), 16, (int)voff, 0, 0, 0);
        }
        // weight-pack chunk of this K tile: taps are the INNER loop when ALIGNED (see the advance below)
        const unsigned qoff = !qok ? kOOB
                              : ALIGNED ? (unsigned)((tr * p.S + ts) * p.cpr + tc * 8 + chunk) * 16u
                                        : (unsigned)(kt * 8 + chunk) * 16u;
        if constexpr (WREG) {
            // (ALIGNED) the K tile's 8 chunks of weight row n start at chunk (tap * cpr + tc * 8); lane quarter lq takes chunk 4 kk + lq
            const unsigned t0 = (unsigned)((tr * p.S + ts) * p.cpr + tc * 8) * 16u;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j)
                    fbn[kk][j] = bload(wr, wq_base[j] >= kOOB ? kOOB : wq_base[j] + t0 + (unsigned)(kk * 64));
        } else {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const unsigned voff = (b_base[i] >= kOOB || !qok) ? kOOB : b_base[i] + qoff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(sB0 + (wrow + i * RSTEP) * 128), 16, (int)voff, 0, 0, 0);
        }
        }
        // advance to the next K tile.  ALIGNED: channel chunk OUTER, filter tap INNER -- the R*S taps of one 64-channel
        // slab re-read (shifted) the same input pixels back to back, so 8 of 9 reads of a 3x3 convolution are served
        // by the XCD's L2 instead of the fabric (the working set of a tap-outer order, 3 image rows x all channels x
        // 32 workgroups, does not fit the 4 MiB L2; measured: the 256x256 kernel was fill-bound at 6.3 TB/s).
        if (ALIGNED) {
            ts += tstep;
            if (ts >= p.S) {
                ts = ts0;
                tr += tstep;
                if (tr >= p.R) { tr = tr0; ++tc; }
            }
        } else {
            tc += 8;
            while (tc >= p.cpr) {
                tc -= p.cpr;
                if (++ts == p.S) { ts = 0; ++tr; }
            }
        }
    };

    f32x16 acc[TM][TN];
    f32x4 acc16[2 * TM][2 * TN];       // M16: 16x16 blocks, D[row = 4*(lane>>4) + e][col = lane&15]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;

    const int nkt = classed ? ((p.R - tr0 + 1) >> 1) * ((p.S - ts0 + 1) >> 1) * (p.cpr >> 3) : (p.kchunks + 7) >> 3;
    const int lr = lane & 31, lh = lane >> 5;
    uint4 fbc[2][2 * TN];          // WREG: the CURRENT K tile's weight fragments
    auto compute = [&](int kk0 = 0, int kk1 = 2) {
        const char* a = sA0;
        const char* b = sB0;
        if constexpr (M16) {
#pragma unroll
            for (int kk = kk0; kk < kk1; ++kk) {      // K step 32 = 4 chunks, one per lane quarter
                const int ch = kk * 4 + lq;
                uint4 fa[2 * TM], fb[2 * TN];
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(a + lds_off(wm * 32 * TM + i * 16 + l15, ch));
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) {
                    if constexpr (WREG) fb[j] = fbc[kk][j];
                    else fb[j] = *reinterpret_cast<const uint4*>(b + lds_off(wn * 32 * TN + j * 16 + l15, ch));
                }
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j) Mma16<T>::run(acc16[i][j], fa[i], fb[j]);
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + lh;
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(a + lds_off(wm * 32 * TM + i * 32 + lr, ch));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(b + lds_off(wn * 32 * TN + j * 32 + lr, ch));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
        }
    };
    if constexpr (RR) {
        // ---- ROW REUSE (3x3, stride 1, pad = dil): the three taps of one filter row read the SAME input pixels shifted by one
        // (dil) column, so ONE fill of a haloed pixel patch serves all three -- the A bytes through the fill path drop to a
        // third (+ halo) and the tile's FLOP per fill byte goes from 77 to 128 (192x128) without a larger register tile
        // (profiles/r02_experiments.md section 5: these kernels are bound by the bytes they pull through that path).
        // The M tile is PW = min(W, BM) consecutive pixels of RT = BM / PW image rows (host: W % 16 == 0, H*W % BM == 0, so
        // a 16-pixel fragment block never straddles an image row and a tile never straddles an image).  LDS patch: RT row
        // segments of PW + 2*dil pixels, 128 B (64 channels) each; tap s of pixel x reads patch column x + s*dil.  K order:
        // 64-channel chunk (outer), filter row r, tap s (inner): A is filled per (chunk, r), B (one tap's 128 x 64 weights) per
        // tap, each transfer issued as early as the single buffers allow (see the early-issue loop below).
        static_assert(M16 && ALIGNED && !STRIDED, "row-reuse kernels");
        constexpr int ARR = BM + 32, SAR = ARR / RSTEP;       // LDS rows of the patch (whole DMA pieces), pieces per wave
        char* const sBr = smem + ARR * 128;       // three weight tiles of BN x 64 channels
        const int dil = p.dil;
        const int PW = p.W < BM ? p.W : BM, RT = BM / PW, PWH = PW + 2 * dil;
        const int hw = p.H * p.W;
        const int b0 = m0 / hw, rem0 = m0 - b0 * hw, oh0 = rem0 / p.W, ow0 = rem0 - oh0 * p.W;
        // The patch is read at EVERY row offset (tap shifts), not only at multiples of 16: its chunk swizzle is keyed by row & 7
        // (conflict-free for ds_read_b128 of 16 consecutive rows from any start row; the tiles' (row >> 1) & 7 key is so only from
        // multiples of 4 -- SQ_LDS_BANK_CONFLICT was 5 % of the wave cycles with it).
        const int chunk_a = (t & 7) ^ (rbase & 7);
        auto lds_off_a = [](int row, int ch) { return row * 128 + (((ch ^ row) & 7) << 4); };
        unsigned long long r_j = 0;    // per piece i: image row j of its patch row (4 bits each: 13 pieces on the 384-row tile)
        unsigned r_okm = 0;            // ... and its "inside the image row" bit
        unsigned r_base[SAR];   // byte offset of its pixel at filter row r = 1 (the centre row), chunk included
#pragma unroll
        for (int i = 0; i < SAR; ++i) {
            const int L = rbase + i * RSTEP;
            const int j = L / PWH, x = L - j * PWH - dil, col = ow0 + x;
            const bool ok = j < RT && col >= 0 && col < p.W;
            r_j |= (unsigned long long)(j & 15) << (4 * i);
            r_okm |= (ok ? 1u : 0u) << i;
            r_base[i] = (unsigned)((b0 * p.H + oh0 + j) * p.W + col) * (unsigned)pixbytes + (unsigned)(chunk_a * 16);
        }
        auto fill_a = [&](int cc, int r) {
            const int dh = (r - 1) * dil;
            const unsigned add = (unsigned)(dh * p.W * pixbytes + cc * 128);
#pragma unroll
            for (int i = 0; i < SAR; ++i) {
                const int ih = oh0 + (int)((r_j >> (4 * i)) & 15u) + dh;
                const bool ok = ((r_okm >> i) & 1u) && (unsigned)ih < (unsigned)p.H;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(sA0 + (wrow + i * RSTEP) * 128), 16,
                                                         (int)(ok ? r_base[i] + add : kOOB), 0, 0, 0);
            }
        };
        // the weights of the three taps of filter row r: three 128 x 64 tiles side by side
        auto fill_b3 = [&](int cc, int r) {
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const unsigned qoff = (unsigned)((r * 3 + s_) * p.cpr + cc * 8 + chunk) * 16u;
#pragma unroll
                for (int i = 0; i < SB; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(sBr + s_ * (BN * 128) + (wrow + i * RSTEP) * 128), 16,
                                                             (int)(b_base[i] >= kOOB ? kOOB : b_base[i] + qoff), 0, 0, 0);
            }
        };
        int arow[2 * TM];       // patch row of the first pixel of fragment block i, tap 0 (wave-uniform: scalar registers)
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) {
            const int pb = __builtin_amdgcn_readfirstlane(wm) * 32 * TM + i * 16;
            const int j = pb / PW;
            arow[i] = __builtin_amdgcn_readfirstlane(j * PWH + (pb - j * PW));
        }
        auto read_frags = [&](int kk, int s_, uint4 (&fa)[2 * TM], uint4 (&fb)[2 * TN]) {
            const int ch = kk * 4 + lq;
            const int sh = s_ * dil + l15;
            const char* bt = sBr + s_ * (BN * 128);
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(sA0 + lds_off_a(arow[i] + sh, ch));
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(bt + lds_off(wn * 32 * TN + j * 16 + l15, ch));
        };
        auto mma_frags = [&](const uint4 (&fa)[2 * TM], const uint4 (&fb)[2 * TN]) {
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) Mma16<T>::run(acc16[i][j], fa[i], fb[j]);
        };
        const int ncc = p.cpr >> 3;
        fill_a(0, 0);
        fill_b3(0, 0);
        for (int cc = 0; cc < ncc; ++cc)
            for (int r = 0; r < 3; ++r) {
                uint4 fa[2 * TM], fb[2 * TN], ga[2 * TM], gb[2 * TN];
                dma_wait<0>();                // (explicit: see the early-issue loop below)
                __syncthreads();              // the patch and the three weight tiles have landed
#pragma unroll
                for (int it = 0; it < 6 - MRFP_RR_HOLD; ++it) {      // (tap, k step) = (0,0) (0,1) (1,0) (1,1) [(2,0)]: no barrier in between
                    read_frags(it & 1, it >> 1, fa, fb);
                    mma_frags(fa, fb);
                }
                if constexpr (MRFP_RR_HOLD == 2) read_frags(0, 2, ga, gb);
                read_frags(1, 2, fa, fb);
                __syncthreads();              // every wave holds its last fragments: the buffers are free
                const int rn = r == 2 ? 0 : r + 1, cn = r == 2 ? cc + 1 : cc;
                if (cn < ncc) {
                    fill_a(cn, rn);
                    fill_b3(cn, rn);
                }
                if constexpr (MRFP_RR_HOLD == 2) mma_frags(ga, gb);
                mma_frags(fa, fb);
            }
        __syncthreads();                      // the epilogue reuses the buffers
    } else {
        // single LDS buffer filled by LDS-DMA: no register staging and no ds_write at all; the fill latency of a
        // workgroup is exposed and hidden only by the other workgroups of the CU (more of them fit: fewer registers)
        if constexpr (M16 && ALIGNED) {
            // EARLY ISSUE: the fragments of the LAST k step go to registers, a barrier says "every wave has read the tile",
            // the transfer of tile kt+1 is issued, and only then the last k step is multiplied -- half of a K tile's matrix
            // work runs inside the fill latency of the next tile.  (The single buffer bounds the bytes in flight per
            // workgroup to one tile and only while it is not computing: tools/fill_micro.hip, profiles/r02_experiments.md
            // section 5 -- the fill path delivers 22-26 TB/s beside an MFMA stream, these kernels draw 13.)  ALIGNED kernels
            // only: with per-thread tap tracking in the address computation (C = 304) the same reordering costs 29 %.
            // HOLD = 2 (both k steps of the K tile held: the whole multiply runs inside the next fill) where the register budget
            // of the tile's occupancy allows it (MRFP_EARLY_FULL, bit 0: 96x128 tile, bit 1: 128x128 tile)
            constexpr int HOLD = ((TM * TN == 3 && (MRFP_EARLY_FULL & 1)) || (TM * TN == 4 && WM == 2 && WN == 2 && (MRFP_EARLY_FULL & 2))) ? 2 : 1;
            uint4 fa[HOLD][2 * TM], fb[HOLD][2 * TN];
            if (nkt > 0) load_tile(0);
            for (int kt = 0; kt < nkt; ++kt) {
                // EXPLICIT vmcnt(0): across the loop's back edge the compiler puts its own wait for the builtin's transfers AFTER
                // the barrier (`s_waitcnt vmcnt(5); s_barrier; s_waitcnt vmcnt(0); ds_read` in the ISA) -- a wave would pass the
                // barrier with its pieces still in flight and the others would read stale LDS
                dma_wait<0>();
                if constexpr (WREG) {
                    // (vmcnt(0) above covers the register loads too) next -> current: the loads of tile kt + 1 below get fresh registers
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < 2 * TN; ++j) { fbc[kk][j] = fbn[kk][j]; settle(fbc[kk][j]); }
                }
                __syncthreads();      // barrier: tile kt has landed everywhere
                if constexpr (HOLD == 1) compute(0, 1);
#pragma unroll
                for (int h = 0; h < HOLD; ++h) {
                    const int ch = (2 - HOLD + h) * 4 + lq;
#pragma unroll
                    for (int i = 0; i < 2 * TM; ++i) fa[h][i] = *reinterpret_cast<const uint4*>(sA0 + lds_off(wm * 32 * TM + i * 16 + l15, ch));
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j) {
                        if constexpr (WREG) fb[h][j] = fbc[2 - HOLD + h][j];
                        else fb[h][j] = *reinterpret_cast<const uint4*>(sB0 + lds_off(wn * 32 * TN + j * 16 + l15, ch));
                    }
                }
                __syncthreads();      // lgkmcnt(0) + barrier: every wave holds its last fragments, the buffer is free
                if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
                for (int h = 0; h < HOLD; ++h)
#pragma unroll
                    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2 * TN; ++j) Mma16<T>::run(acc16[i][j], fa[h][i], fb[h][j]);
            }
            __syncthreads();          // the epilogue reuses the buffer
        } else
        for (int kt = 0; kt < nkt; ++kt) {
            load_tile(kt);
            dma_wait<0>();            // (the compiler's own wait sits here as well; spelled out so that it cannot move behind the barrier)
            __syncthreads();          // vmcnt(0) + barrier: the tile has landed
            compute();
            __syncthreads();          // everybody is done reading before the next fill
        }
    }

    MRFP_STAMP_END(g_stamps_igemm);
    // epilogue.  MFMA 32x32 accumulator layout: D[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31], i.e. a
    // lane owns single elements of 16 rows -- storing that directly is 2-byte scattered traffic.  Instead every
    // wave transposes its tile through LDS (free after the K loop) 32 rows at a time and writes whole 16-byte
    // chunks of output rows, 8 (bf16) / 4 (fp32) rows per wave-instruction, fully coalesced.
    constexpr int EPC = 16 / (int)sizeof(T);              // elements per 16-byte chunk
    constexpr int ROWB = 32 * TN * (int)sizeof(T);        // bytes of one row of the wave's tile
    constexpr bool WIDE = kWideEp && WN > 1 && (64 % (WN * ROWB / 16) == 0) && (32 % WN == 0) &&
                          ((32 / WN) % (64 / (WN * ROWB / 16)) == 0);
    constexpr int EPITCH = (WIDE ? WN * ROWB : ROWB) + 16; // +16: the two lane halves (rows r, r+4) hit disjoint banks
    constexpr int CPRW = (WIDE ? WN * ROWB : ROWB) / 16;   // chunks per staged row
    constexpr int RPI = 64 / CPRW;                         // rows per wave-instruction
    // per-wave strips: 4.5 KB (bf16) / 8.5 KB (fp32) per wave; WIDE: one 32-row block of the workgroup tile per wave row
    char* const ep = WIDE ? smem + wm * (32 * EPITCH) + wn * ROWB : smem + wave * (32 * EPITCH);
    char* const epr = WIDE ? smem + wm * (32 * EPITCH) : ep;
    T* y = reinterpret_cast<T*>(p.y);
    const int nb = n0 + wn * 32 * TN;
    float bv[2 * TN], cs[2 * TN], cq[2 * TN];      // 32x32 blocks use the first TN entries, 16x16 blocks all of them
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
        const int n = M16 ? nb + j * 16 + l15 : nb + j * 32 + lr;
        bv[j] = (p.bias && n < p.N && (M16 || j < TN)) ? p.bias[n] : 0.f;
        cs[j] = 0.f;
        cq[j] = 0.f;
    }
    // `wst` (uniform): the statistics count row m `rowweight[m]` times (ConvP::rowweight).  That variant of the staging loop is a
    // separate copy: the launches without a table -- every convolution outside the HRFP branch -- run the loop they always ran
    // (folding both into one loop cost those launches 0.6 %: same-box A/B against the previous library, profiles/r04_experiments.md 13)
    const bool wst = p.colstats && p.rowweight;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (wst) {
            // weights of this lane's rows: four consecutive rows per 32-bit load (the table is padded to whole tiles and 4-byte aligned,
            // every group of four rows starts at a multiple of 4); rows past M weigh 0
            if constexpr (M16) {
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const int rbase = m0 + wm * 32 * TM + i * 32 + i2 * 16 + 4 * lq;
                    const unsigned pk = *reinterpret_cast<const unsigned*>(p.rowweight + rbase);
                    float w4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w4[e] = (rbase + e < p.M) ? (float)((pk >> (8 * e)) & 0xffu) : 0.f;
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = i2 * 16 + 4 * lq + e;
                            const T sv = from_f<T>(acc16[2 * i + i2][j][e] + bv[j]);
                            *reinterpret_cast<T*>(ep + row * EPITCH + (j * 16 + l15) * (int)sizeof(T)) = sv;
                            const float fv = to_f(sv), wf = fv * w4[e];
                            cs[j] += wf;
                            cq[j] += wf * fv;
                        }
                }
            } else {
                float w16[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rbase = m0 + wm * 32 * TM + i * 32 + 8 * g + 4 * lh;
                    const unsigned pk = *reinterpret_cast<const unsigned*>(p.rowweight + rbase);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w16[4 * g + e] = (rbase + e < p.M) ? (float)((pk >> (8 * e)) & 0xffu) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const T sv = from_f<T>(acc[i][j][e] + bv[j]);
                        *reinterpret_cast<T*>(ep + row * EPITCH + (j * 32 + lr) * (int)sizeof(T)) = sv;
                        const float fv = to_f(sv), wf = fv * w16[e];
                        cs[j] += wf;
                        cq[j] += wf * fv;
                    }
            }
        } else
        if constexpr (M16) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = i2 * 16 + 4 * lq + e;
                        const T sv = from_f<T>(acc16[2 * i + i2][j][e] + bv[j]);
                        *reinterpret_cast<T*>(ep + row * EPITCH + (j * 16 + l15) * (int)sizeof(T)) = sv;
                        if (p.colstats) {
                            const float fv = (m0 + wm * 32 * TM + i * 32 + row < p.M) ? to_f(sv) : 0.f;
                            cs[j] += fv;
                            cq[j] += fv * fv;
                        }
                    }
        } else
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const T sv = from_f<T>(acc[i][j][e] + bv[j]);
                *reinterpret_cast<T*>(ep + row * EPITCH + (j * 32 + lr) * (int)sizeof(T)) = sv;
                if (p.colstats) {      // BatchNorm statistics of the STORED (rounded) values, fused into the producer
                    const float fv = (m0 + wm * 32 * TM + i * 32 + row < p.M) ? to_f(sv) : 0.f;
                    cs[j] += fv;
                    cq[j] += fv * fv;
                }
            }
        // same-wave LDS round trip: no workgroup barrier needed, only the wave's own LDS ops must have landed
        // (WIDE: the block is shared by the WN waves of this wave row -> workgroup barriers around the read-out)
        if constexpr (WIDE) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // also a compiler barrier (T stores vs uint4 loads)
        const int mb = m0 + wm * 32 * TM + i * 32;
        constexpr int NK = WIDE ? (32 / WN) / RPI : 32 / RPI;      // read-out instructions of this wave
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int row = (WIDE ? wn * (32 / WN) : 0) + k * RPI + lane / CPRW, ch = lane % CPRW;
            const int mlog = mb + row, n = (WIDE ? n0 : nb) + ch * EPC;
            if (mlog < p.M && n < p.N) {
                int m = mlog;
                if (classed) {       // back from the class-major row index to the pixel's row of the output
                    int b, oh, ow;
                    pixel_of(mlog, b, oh, ow);
                    m = (b * p.Ho + oh) * p.Wo + ow;
                }
                uint4 v = *reinterpret_cast<const uint4*>(epr + row * EPITCH + ch * 16);
                const int mquad = m;
#pragma unroll 1
                for (int qd = 0; qd < (quad ? 4 : 1); ++qd) {
                if (qd > 0) {          // the three tap-less pixels of the 2 x 2 block: zeros (+ addend)
                    m = mquad + (qd >> 1) * p.Wo + (qd & 1);
                    v = make_uint4(0u, 0u, 0u, 0u);
                }
                T* dst = y + (size_t)m * p.ldy + n;
                const bool full = n + EPC <= p.N;
                // (everything below indexes the chunk with compile-time constants only: a run-time index would
                //  push `v` into scratch memory)
                if (p.addend) {      // y += addend (the skip-connection gradient): one 16-byte read instead of a separate add pass
                    const T* ad = reinterpret_cast<const T*>(p.addend) + (size_t)m * p.ldy + n;
                    uint4 av = make_uint4(0u, 0u, 0u, 0u);
                    if (full) {
                        av = *reinterpret_cast<const uint4*>(ad);
                    } else {
#pragma unroll
                        for (int u = 0; u < EPC; ++u)
                            if (n + u < p.N) chunk_set<T>(av, u, ad[u]);
                    }
                    if constexpr (sizeof(T) == 2) {
                        if (p.addend_mask) av = gate_chunk16(av, p.addend_mask[((size_t)m * p.ldy + n) >> 3]);
                    }
                    v = chunk_add<T>(v, av);
                }
                if (full) {
                    *reinterpret_cast<uint4*>(dst) = v;
                } else {
#pragma unroll
                    for (int u = 0; u < EPC; ++u)
                        if (n + u < p.N) dst[u] = chunk_get<T>(v, u);
                }
                }       // qd
            }
        }
        if constexpr (WIDE) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (p.colstats) {
        // a lane holds 16 of the 32 rows of each block column, its partner (lane ^ 32) the other 16
        float* out = p.colstats + (size_t)((tile / ntn) * WM + wm) * 2 * p.ldy;
        if constexpr (M16) {
            // a lane holds 8 of the 32 rows of each 16-column block; lanes ^16, ^32 hold the others
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) {
                float s2 = cs[j] + __shfl_xor(cs[j], 16, 64), q2 = cq[j] + __shfl_xor(cq[j], 16, 64);
                s2 += __shfl_xor(s2, 32, 64);
                q2 += __shfl_xor(q2, 32, 64);
                const int n = nb + j * 16 + l15;
                if (lq == 0 && n < p.N) {
                    out[n] = s2;
                    out[p.ldy + n] = q2;
                }
            }
        } else
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float s2 = cs[j] + __shfl_xor(cs[j], 32, 64), q2 = cq[j] + __shfl_xor(cq[j], 32, 64);
            const int n = nb + j * 32 + lr;
            if (lh == 0 && n < p.N) {
                out[n] = s2;
                out[p.ldy + n] = q2;
            }
        }
    }
}

// Folds the per-row-block statistics [nblk][2][C] of a forward launch into kStatGroups rows (appended after row
// nblk) so that the BatchNorm finalize kernel walks 64 partials per channel instead of thousands.
constexpr int kStatGroups = 64;
// up to this many row blocks the BatchNorm finalize kernel (8 channels x 128 partial lanes per workgroup) walks the
// partials itself; a separate compaction launch costs ~5 us whatever it does
constexpr int kCompactAbove = 2048;
__global__ __launch_bounds__(256) void compact_stats_kernel(float* __restrict__ st, int nblk, int C2) {   // C2 = 2*C floats per row
    // block = 64 columns x 4 row quarters (fixed split and fixed order: bitwise reproducible)
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, g = blockIdx.y;
    const int per = (nblk + kStatGroups - 1) / kStatGroups;
    const int r0 = g * per, r1 = min(nblk, r0 + per);
    const int q = (r1 - r0 + 3) / 4;
    const int a = min(r1, r0 + ty * q), b = min(r1, a + q);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C2) {
        int r = a;
        for (; r + 3 < b; r += 4) {
            a0 += st[(size_t)r * C2 + c];
            a1 += st[(size_t)(r + 1) * C2 + c];
            a2 += st[(size_t)(r + 2) * C2 + c];
            a3 += st[(size_t)(r + 3) * C2 + c];
        }
        for (; r < b; ++r) a0 += st[(size_t)r * C2 + c];
    }
    part[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && c < C2) st[(size_t)(nblk + g) * C2 + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

// K tiles are staged by LDS-DMA into ONE LDS buffer per workgroup (fill, barrier, multiply, barrier, with the next fill issued
// early: see the kernel); the 3-5 co-resident workgroups of a CU hide each other's fill latency.
template <typename T, int WM, int WN, bool ALIGNED, bool STRIDED, int TM, int TN>
static int launch_igemm(const ConvP& p, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int EP = WM * WN * 32 * (32 * TN * (int)sizeof(T) + 16);      // epilogue staging (+16 bytes of pitch per row)
    const int lds = (BM + BN) * 128 > EP ? (BM + BN) * 128 : EP;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, WM, WN, ALIGNED, STRIDED, TM, TN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, ALIGNED, STRIDED, TM, TN>), dim3((unsigned)tiles), dim3(64 * WM * WN), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

// row-reuse kernels (conv_igemm_kernel<..., RR = true>): 3x3, stride 1, pad = dil, 64-channel-aligned C, W % 16 == 0
template <typename T, int WM, int WN, int TM, int TN>
static int launch_igemm_rr(const ConvP& p, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int EP = WM * WN * 32 * (32 * TN * (int)sizeof(T) + 16);
    const int fill = (BM + 32 + 3 * BN) * 128;
    const int lds = fill > EP ? fill : EP;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, WM, WN, true, false, TM, TN, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, true, false, TM, TN, true>), dim3((unsigned)tiles),
                       dim3(64 * WM * WN), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T, int WM, int WN, int TM, int TN>
static int pick_igemm(const ConvP& p, hipStream_t st) {
    const bool aligned = (p.cpr & 7) == 0, strided = p.sstride > 1;
    if (aligned) return strided ? launch_igemm<T, WM, WN, true, true, TM, TN>(p, st) : launch_igemm<T, WM, WN, true, false, TM, TN>(p, st);
    return strided ? launch_igemm<T, WM, WN, false, true, TM, TN>(p, st) : launch_igemm<T, WM, WN, false, false, TM, TN>(p, st);
}

// 96x128 tile (4 waves x 96x32): same K loop, 3/4 of the rows.  Chosen when the 128-row tiling leaves the last
// round of workgroups (3 per CU x 256 CUs) mostly empty: M = 36 864 (16 x 48 x 48), N = 256 is 576 tiles = 0.75 rounds
// at 128 rows but exactly one full round (768) at 96 rows.
static int g_t96 = -1;
static bool use_tile96(const ConvP& p, int esz) {
    if (g_t96 < 0) {
        const char* e = getenv("MRFP_CONV_T96");
        g_t96 = e ? atoi(e) : 1;
    }
    if (!g_t96 || p.N <= 64) return false;
    if (g_t96 == 2) return true;      // A/B measurements: 96-row tile wherever it is legal
    // strided dgrads (a stride-2 forward's input gradient): the 128x128 tile won all three layers of the bench step by 8-13 %
    // (tools/tile_sweep.sh, round 3)
    if (p.sstride > 1) return false;
    // rounds of resident workgroups: 3 per CU for the 128x128 tile (144-148 registers), 4 per CU for the 96x128 tile
    // (<= 120).  Measured in the bench workload (bench.py --dump-convs, MRFP_CONV_T96=0/1/2): the 96-row tile wins
    // when everything fits one round (M = 36 864 layers: +7..35 %) and on short-K (memory-bound) layers; long-K
    // layers with many rounds keep the 128x128 tile (higher FLOP per LDS byte).
    const int64_t n128 = (p.N + 127) / 128;
    const int64_t t128 = ((p.M + 127) / 128) * n128, t96 = ((p.M + 95) / 96) * n128;
    const int nkt = (p.kchunks + 7) >> 3;
    if (t96 <= 1024) return true;
    if (nkt > 8) return false;
    const int64_t c128 = ((t128 + 767) / 768) * 128, c96 = ((t96 + 1023) / 1024) * 96;
    return c96 * 10 <= c128 * 9;       // at least 10 % fewer row-rounds
}
// 192x128 tile (4 waves x 96x64): 77 FLOP per LDS-fill byte instead of 64 and 0.83 KiB of fragment reads per MFMA instead
// of 1; for long-K layers with many rounds of tiles
static int g_t192 = -1;
static bool use_tile192(const ConvP& p, int esz) {
    if (g_t192 < 0) {
        const char* e = getenv("MRFP_CONV_T192");      // =0: A/B measurements (measured +4..5 % on the long-K layers)
        g_t192 = e ? atoi(e) : 1;
    }
    if (!g_t192 || esz != 2 || p.N <= 64) return false;
    if (g_t192 == 2) return true;        // A/B measurements: 192-row tile wherever it is legal
    if (p.sstride > 1) return false;     // strided dgrads: the 128x128 tile wins (-12 %: the strided 192x128 instance spills 48 bytes per lane)
    // Whole rounds: three workgroups of this tile fit a CU (768 slots), and a launch of exactly 1, 2 or 4 rounds keeps every CU
    // equally busy to the end -- measured (tools/tile_sweep.sh, round 3) on every such layer of the bench step against the tile
    // the rules below pick: 2048 -> 512 @48^2 1x1 -19 %, 128 -> 128 @96^2 3x3 -16 %, 512 -> 2048 -11 %, 512 -> 128 @96^2 -8 %,
    // 512 -> 512 @48^2 3x3 d2 -6 %, the stride-2 downsample 1x1s -12..-14 %.  (Not the strided dgrads: +47 %; not unaligned C.)
    {
        const int64_t t192 = ((int64_t)(p.M + 191) / 192) * ((p.N + 127) / 128);
        if (p.sstride == 1 && (p.cpr & 7) == 0 && p.M % 192 == 0 && t192 % 768 == 0 && t192 <= 3072) return true;
    }
    if (use_tile96(p, esz)) return false;
    const int nkt = (p.kchunks + 7) >> 3;
    return nkt >= 9 && ((p.M + 191) / 192) * ((p.N + 127) / 128) >= 2048;
}

// (measured and dropped in round 2: a 160x128 tile, 4 waves x 160x32, 71 instead of 55 FLOP per fill byte and 462 tiles
//  = one round at two workgroups per CU for the M = 36 864, N = 256 layers: 55.9 vs 49.8 us on the 3x3 layer, 31.5 vs
//  29.3 us on the 1024 -> 256 pointwise layer -- fewer co-resident workgroups cost more than the fill bytes save)

// Row-reuse kernels: the tile (192 rows; 0 = not applicable) a launch runs on.  MRFP_CONV_RR=0: off; 1: where the plain
// kernel would run the 192x128 tile; 2 (default): also instead of the 128x128 / 96x128 tiles; 3: wherever 192 rows fit the image
// geometry (tests).  (A 96-row and a 192x64 variant for the N <= 64 layers were measured slower in round 2 and removed.)
static int g_rr = -1;
static int rr_tile(const ConvP& p, int esz) {
    if (g_rr < 0) {
        const char* e = getenv("MRFP_CONV_RR");
        g_rr = e ? atoi(e) : 2;
    }
    if (!g_rr || esz != 2) return 0;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.sstride != 1 || p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.dil < 1 || p.dil > 2 || p.pad_h != p.dil || p.pad_w != p.dil) return 0;
    if ((p.cpr & 7) != 0 || (p.W & 15) != 0) return 0;
    auto fits = [&](int BM) {
        if ((p.H * p.W) % BM != 0 || (p.W % BM != 0 && BM % p.W != 0)) return false;
        const int PW = p.W < BM ? p.W : BM, RT = BM / PW;
        return RT * (PW + 2 * p.dil) <= BM + 32;
    };
    if (p.N <= 64) {
        // N <= 64 (round 5): the HRFP ends (reference deepv3.py:221-237: 128 -> 64, 64 -> 64 at 192^2 .. 384^2) ran on
        // the plain 256x64 tile at 52 FLOP per fill byte -- 3.3 GB through the fill path for the 64 -> 64 layer at 384^2 = 255 us at
        // the 13 TB/s that path gives, and 270 us is what the launch takes.  A 384 x 64 tile of four waves STACKED along M (each
        // 96 x 64: the per-wave tile, fragment reads and MFMA count of the 192x128 row-reuse kernel) fills a patch of 384 + 32 pixel
        // rows and three 64 x 64 weight tiles per filter row: 77 KB for 3 x 144 MFMAs per wave, 122 FLOP per fill byte.  (Round 2's
        // 192 x 64 variant -- 53 KB per fill, 82 FLOP per byte, three workgroups per CU -- had measured slower than the plain tile.)
        // MRFP_CONV_RR64=0 switches it off (A/B runs).
        static int rr64 = -1;
        if (rr64 < 0) { const char* e = getenv("MRFP_CONV_RR64"); rr64 = e ? atoi(e) : 1; }
        if (!rr64 || !fits(384)) return 0;
        const int64_t t384 = (int64_t)(p.M / 384);
        return (g_rr >= 3 || t384 >= 512) ? 384 : 0;      // at least one full round at two workgroups per CU
    }
    // Where it pays (bench.py --dump-convs with the switch off / on, several boxes): long K (C >= 256: at least 12 patch fills
    // per tile to amortise the 76 KB prologue), at least one full round of tiles at two workgroups per CU, and an N that does not
    // waste most of its last 128-column tile.  Lost: M = 36 864, 256 -> 256 (384 tiles: 825 vs 880 TFLOP/s against the 96x128
    // tile), C = 128 (862 vs 912), N = 304 (918 vs 977).  Mode 3 lifts these restrictions (A/B runs).
    const int64_t t192 = (int64_t)(p.M / 192) * ((p.N + 127) / 128);
    // (two of these workgroups fit a CU: a launch that is not whole rounds of 512 and short -- 512 -> 512 @48^2 d2, 768 tiles -- runs
    //  better on the plain 192x128 tile, three per CU: 887 vs 930 us for the six launches of the bench step)
    const bool pays = p.C >= 256 && t192 >= 512 && (t192 >= 2048 || t192 % 512 == 0) && ((p.N + 127) / 128) * 128 - p.N <= 64;
    if (fits(192) && (g_rr >= 3 || (pays && (g_rr >= 2 || use_tile192(p, esz))))) return 192;
    return 0;
}

// number of statistics row blocks (= m-tiles x wave rows) the epilogue of a forward launch writes
static int64_t stats_row_blocks(const ConvP& p, int esz) {
    if (pw_applicable(p, esz)) return pw_stats_blocks(p);            // pointwise kernels (conv_pw.hip): one per workgroup range
    if (pwk_applicable(p, esz)) return pwk_stats_blocks(p);          // long-K pointwise kernel (conv_pwk.hip): likewise
    if (c64_applicable(p, esz)) return c64_stats_blocks(p);          // weight-stationary 3x3 kernel (conv_c64.hip): one per workgroup and sub-strip
    if (rr_tile(p, esz) == 384) return (int64_t)(p.M / 384) * 4;      // row-reuse kernel for N <= 64: 384-row tiles, 4 wave rows
    if (rr_tile(p, esz)) return (int64_t)(p.M / 192) * 2;             // row-reuse kernels: 192-row tiles, 2 wave rows
    if (p.N > 64 && use_tile192(p, esz)) return (int64_t)((p.M + 191) / 192) * 2;      // <2,2,3,2>: 192-row tile, 2 wave rows
    if (p.N <= 64) return (int64_t)((p.M + 255) / 256) * 4;          // <4,1,2,2>: 256-row tile, 4 wave rows
    if (use_tile96(p, esz)) return (int64_t)((p.M + 95) / 96);          // <1,4,3,1>: 96-row tile, 1 wave row
    return (int64_t)((p.M + 127) / 128) * 2;                          // <2,2,2,2>: 128-row tile, 2 wave rows
}

// output rows one statistics row block covers (block r = rows [r * rb, (r + 1) * rb) of the [M][N] output)
static int64_t stats_block_rows(const ConvP& p, int esz) {
    if (pw_applicable(p, esz)) return pw_stats_block_rows(p);
    if (pwk_applicable(p, esz)) return pwk_stats_block_rows(p);
    if (c64_applicable(p, esz)) return c64_stats_block_rows(p);     // negative: -(statistics rows per image)
    if (rr_tile(p, esz)) return 96;
    if (p.N > 64 && use_tile192(p, esz)) return 96;
    if (p.N <= 64) return 64;
    if (use_tile96(p, esz)) return 96;
    return 64;
}

// tile height (output rows) run_igemm picks for the dgrad of a strided convolution (sstride > 1: neither the pointwise nor the
// row-reuse kernels take those).  The class-major row order (ConvP::classed) needs every parity class to be WHOLE tiles of it --
// with MRFP_CONV_T96 / T192 = 2 (A/B runs, tests) the 96- and 192-row tiles reach strided launches too.
static int strided_tile_rows(const ConvP& p, int esz) {
    if (p.N <= 64) return 256;
    if (esz == 2 && use_tile192(p, esz)) return 192;
    if (use_tile96(p, esz)) return 96;
    return 128;
}

template <typename T>
static int run_igemm(const ConvP& p, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (pw_applicable(p, 2)) return pw_run(p, std::is_same<T, f16>::value, st);
        if (pwk_applicable(p, 2)) return pwk_run(p, std::is_same<T, f16>::value, st);
        if (c64_applicable(p, 2)) return c64_run(p, std::is_same<T, f16>::value, st);
        const int rr = rr_tile(p, 2);
        if (rr == 384) return launch_igemm_rr<T, 4, 1, 3, 2>(p, st);
        if (rr) return launch_igemm_rr<T, 2, 2, 3, 2>(p, st);
    }
    if (p.N <= 64) return pick_igemm<T, 4, 1, 2, 2>(p, st);
    // (measured and dropped: the 256x256 8-wave tile, a two-wave 96x128 variant, a 256x128 tile, a 160x128 tile:
    //  profiles/r02_experiments.md)
    if constexpr (sizeof(T) == 2) {
        if (use_tile192(p, (int)sizeof(T))) return pick_igemm<T, 2, 2, 3, 2>(p, st);
    }
    if (use_tile96(p, (int)sizeof(T))) return pick_igemm<T, 1, 4, 3, 1>(p, st);
    return pick_igemm<T, 2, 2, 2, 2>(p, st);
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

static int conv_fwd_impl(const void* x, const void* wpack, const float* bias, void* y, int dtype, int64_t B, int64_t H,
                         int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                         int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride, const void* addend,
                         float* colstats, void* stream, const void* addend_mask = nullptr, const unsigned char* rowweight = nullptr) {
    MRFP_CHECK(!addend || aligned16(addend), "conv_fwd: addend must be 16-byte aligned");
    MRFP_CHECK(!rowweight || (colstats && ((uintptr_t)rowweight & 3) == 0), "conv_fwd_wstats: the row weights need statistics and 4-byte alignment");
    MRFP_CHECK(!addend_mask || (addend && dtype != MRFP_F32 && (N & 7) == 0 && ldy == N),
               "conv_fwd_gated: a gate mask needs an addend, 16-bit activations, N %% 8 == 0 and a dense output");
    MRFP_CHECK(x && wpack && y && B > 0 && H > 0 && W > 0 && C > 0 && N > 0 && R > 0 && S > 0 && Ho > 0 && Wo > 0,
               "conv_fwd: bad arguments");
    MRFP_CHECK(stride >= 1 && dil >= 1 && sstride >= 1 && ldy >= N, "conv_fwd: bad stride/dilation/pitch");
    const int esz = dtype == MRFP_F32 ? 4 : 2;
    MRFP_CHECK(dtype == MRFP_F32 || dtype == MRFP_BF16 || dtype == MRFP_F16, "conv_fwd: unknown dtype %d", dtype);
    MRFP_CHECK((C * esz) % 16 == 0, "conv_fwd: C=%lld must make 16-byte chunks (pad the channels)", (long long)C);
    MRFP_CHECK(aligned16(x) && aligned16(wpack), "conv_fwd: x / wpack must be 16-byte aligned");
    MRFP_CHECK(B * Ho * Wo < (1LL << 31), "conv_fwd: tensor too large for 32-bit tile indices");
    ConvP p;
    p.x = (const char*)x; p.w = (const char*)wpack; p.y = (char*)y; p.bias = bias; p.addend = (const char*)addend; p.colstats = colstats;
    p.addend_mask = (const unsigned char*)addend_mask;
    p.rowweight = rowweight;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.N = (int)N; p.ldy = (int)ldy;
    p.R = (int)R; p.S = (int)S; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.stride = (int)stride; p.pad_h = (int)pad_h; p.pad_w = (int)pad_w; p.dil = (int)dil; p.sstride = (int)sstride;
    p.M = (int)(B * Ho * Wo); p.cpr = (int)(C * esz / 16); p.kchunks = (int)(R * S * p.cpr);
    {   // dgrad of a stride-2 convolution: parity-class-major rows (see the kernel) where the geometry allows it -- even output size,
        // whole 64-channel K tiles, classes of whole (<= 256-row) tiles, one launch.  MRFP_DGRAD_CLASSED=0: the per-pixel form (A/B runs)
        static int on = -1;
        if (on < 0) { const char* e = getenv("MRFP_DGRAD_CLASSED"); on = e ? atoi(e) : 1; }
        p.classed = on && sstride == 2 && stride == 1 && dil == 1 && (Ho & 1) == 0 && (Wo & 1) == 0 && (p.cpr & 7) == 0 &&
                    ((B * (Ho / 2) * (Wo / 2)) % 256) == 0 && !colstats;
        // ... and whole tiles of the height this launch will actually run (a tile straddling two classes would index past dx)
        if (p.classed && ((B * (Ho / 2) * (Wo / 2)) % strided_tile_rows(p, esz)) != 0) p.classed = 0;
        // pointwise: rows = the even-even pixels only, the epilogue stores 2 x 2 blocks (measured: three classes of store-only tiles
        // made the class-major form 20 % SLOWER than the per-pixel one on the 1x1 stride-2 downsample dgrads)
        if (p.classed && R == 1 && S == 1 && pad_h == 0 && pad_w == 0) p.classed = 2;
    }
    const int64_t img = H * W * C * esz, wb = N * (int64_t)p.kchunks * 16;      // bytes of one input image, of the pack
    MRFP_CHECK(img < (int64_t)kOOB && wb < (int64_t)kOOB,
               "conv_fwd: one input image / the weight pack exceeds the 3.75 GB buffer-descriptor range");
    // The gather addresses of the K loop are 32-bit offsets into a buffer descriptor (hardware bounds check = zero fill
    // for padding and tails), so ONE launch can read at most kOOB bytes of input.  A larger activation (configs[4] at 16
    // images per GPU: 16 x 256 x 512 x 1024 bf16 = 4.3 GB) runs as several launches over batch ranges; images are
    // independent in a convolution, so nothing else changes.  (The fused per-row-block statistics are per launch: the
    // caller does not ask for them on such tensors, mrfp_conv_single_launch() tells it.)
    const int64_t bmax = (int64_t)(kOOB - 1) / img;          // images per launch
    const bool chunked = B > bmax;
    MRFP_CHECK(!chunked || !colstats, "conv_fwd: fused statistics are not available for inputs above 3.75 GB (see mrfp_conv_single_launch)");
    MRFP_CHECK(!rowweight || !pw_applicable(p, esz), "conv_fwd_wstats: not available on the pointwise kernels");
    if (chunked) p.classed = 0;          // (batch ranges: the class size would change per range)
    int dbg_drop = 0;
    {   // timing-only diagnostics: zero-record descriptors drop that operand's traffic, instruction stream unchanged
        static int dbg = -1;
        if (dbg < 0) { const char* e = getenv("MRFP_DEBUG_DROP"); dbg = e ? atoi(e) : 0; }
        dbg_drop = dbg;
    }
    int rc = 0;
    for (int64_t b0 = 0; b0 < B && !rc; b0 += bmax) {
        const int64_t bc = B - b0 < bmax ? B - b0 : bmax;
        const int64_t m0 = b0 * Ho * Wo;
        p.B = (int)bc;
        p.M = p.classed == 2 ? (int)(bc * (Ho / 2) * (Wo / 2)) : (int)(bc * Ho * Wo);
        p.x = (const char*)x + b0 * img;
        p.y = (char*)y + m0 * ldy * esz;
        p.addend = addend ? (const char*)addend + m0 * ldy * esz : nullptr;
        p.addend_mask = addend_mask ? (const unsigned char*)addend_mask + ((m0 * ldy) >> 3) : nullptr;
        p.xbytes = (dbg_drop & 1) ? 0u : (unsigned)(bc * img);
        p.wbytes = (dbg_drop & 2) ? 0u : (unsigned)wb;
        rc = dtype == MRFP_F32 ? run_igemm<float>(p, (hipStream_t)stream)
             : dtype == MRFP_F16 ? run_igemm<f16>(p, (hipStream_t)stream) : run_igemm<bf16>(p, (hipStream_t)stream);
    }
    if (rc || !colstats) return rc;
    const int64_t nblk = stats_row_blocks(p, esz);
    if (nblk > kCompactAbove) {
        const int C2 = 2 * (int)ldy;
        hipLaunchKernelGGL(compact_stats_kernel, dim3((unsigned)((C2 + 63) / 64), kStatGroups), dim3(256), 0,
                           (hipStream_t)stream, colstats, (int)nblk, C2);
        MRFP_LAUNCH_CHECK();
    }
    return 0;
}

/* diagnostic: the clock stamps of the last launch of a kernel family (0: conv_igemm, 1: pointwise, 2: wgrad; 3: the per-phase cycle sums
 * of the two-group long-K pointwise kernel, conv_pwk.hip / tools/phase_stamp.py) -- only in a library built with -DMRFP_CLOCK_STAMP=1
 * (tools/clock_stamp.py); the product build returns -1 */
int mrfp_debug_clock_stamps(int family, uint64_t* out, int64_t n) {
    MRFP_CHECK(MRFP_CLOCK_STAMP != 0, "clock stamps: not a diagnostic build (-DMRFP_CLOCK_STAMP=1)");
    MRFP_CHECK(out && n > 0 && n <= kStampSlots && family >= 0 && family <= 3, "clock stamps: bad arguments");
    (void)hipDeviceSynchronize();
    const int rc = family == 0 ? stamps_igemm((unsigned long long*)out, (int)n)
                   : family == 1 ? stamps_pw((unsigned long long*)out, (int)n)
                   : family == 2 ? stamps_wgrad((unsigned long long*)out, (int)n) : stamps_pwk((unsigned long long*)out, (int)n);
    MRFP_CHECK(rc == 0, "clock stamps: read-back failed");
    return 0;
}

/* 1 when a convolution over an input of B images of `image_bytes` bytes runs as ONE launch (fused statistics available) */
int mrfp_conv_single_launch(int64_t B, int64_t image_bytes) {
    return image_bytes > 0 && image_bytes < (int64_t)kOOB && B <= (int64_t)(kOOB - 1) / image_bytes;
}

int mrfp_conv_fwd(const void* x, const void* wpack, const float* bias, void* y, int dtype, int64_t B, int64_t H,
                  int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                  int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride, const void* addend,
                  float* colstats, void* stream) {
    return conv_fwd_impl(x, wpack, bias, y, dtype, B, H, W, C, N, ldy, R, S, Ho, Wo, stride, pad_h, pad_w, dil, sstride, addend,
                         colstats, stream);
}

int mrfp_conv_fwd_wstats(const void* x, const void* wpack, const float* bias, void* y, int dtype, int64_t B, int64_t H,
                         int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                         int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, const uint8_t* rowweight, float* colstats,
                         void* stream) {
    MRFP_CHECK(rowweight && colstats, "conv_fwd_wstats: row weights and a statistics buffer are required");
    return conv_fwd_impl(x, wpack, bias, y, dtype, B, H, W, C, N, ldy, R, S, Ho, Wo, stride, pad_h, pad_w, dil, 1, nullptr,
                         colstats, stream, nullptr, rowweight);
}

int mrfp_conv_fwd_gated(const void* x, const void* wpack, const float* bias, void* y, int dtype, int64_t B, int64_t H,
                        int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                        int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride, const void* addend,
                        const void* addend_mask, void* stream) {
    MRFP_CHECK(addend && addend_mask, "conv_fwd_gated: addend and its gate mask are required");
    return conv_fwd_impl(x, wpack, bias, y, dtype, B, H, W, C, N, ldy, R, S, Ho, Wo, stride, pad_h, pad_w, dil, sstride, addend,
                         nullptr, stream, addend_mask);
}

int64_t mrfp_conv_stats_blocks(int dtype, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t R, int64_t S, int64_t Ho,
                               int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride) {
    // the SAME geometry the launch will see (conv_fwd_impl): the kernel choice -- and with it the number of row blocks -- looks at
    // the filter, stride, padding, dilation and image size, not only at M, N, C
    ConvP p;
    const int esz = dtype == MRFP_F32 ? 4 : 2;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.N = (int)N; p.ldy = (int)N;
    p.R = (int)R; p.S = (int)S; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.stride = (int)stride; p.pad_h = (int)pad_h; p.pad_w = (int)pad_w; p.dil = (int)dil; p.sstride = (int)sstride;
    p.M = (int)(B * Ho * Wo); p.cpr = (int)(C * esz / 16); p.kchunks = (int)(R * S * p.cpr);
    p.classed = 0;
    p.bias = nullptr; p.colstats = nullptr; p.rowweight = nullptr; p.addend = nullptr; p.addend_mask = nullptr;
    return stats_row_blocks(p, esz);
}
int64_t mrfp_conv_stats_block_rows(int dtype, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t R, int64_t S, int64_t Ho,
                                   int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride) {
    ConvP p;
    const int esz = dtype == MRFP_F32 ? 4 : 2;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.N = (int)N; p.ldy = (int)N;
    p.R = (int)R; p.S = (int)S; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.stride = (int)stride; p.pad_h = (int)pad_h; p.pad_w = (int)pad_w; p.dil = (int)dil; p.sstride = (int)sstride;
    p.M = (int)(B * Ho * Wo); p.cpr = (int)(C * esz / 16); p.kchunks = (int)(R * S * p.cpr);
    p.classed = 0;
    p.bias = nullptr; p.colstats = nullptr; p.rowweight = nullptr; p.addend = nullptr; p.addend_mask = nullptr;
    return stats_block_rows(p, esz);
}
/* rows the caller must allocate for `colstats` (row blocks + the compacted groups) */
int64_t mrfp_conv_stats_rows(int64_t nblk) { return nblk > kCompactAbove ? nblk + kStatGroups : nblk; }
/* where the rows to hand to mrfp_bn_finalize start, and how many there are */
int64_t mrfp_conv_stats_final_first(int64_t nblk) { return nblk > kCompactAbove ? nblk : 0; }
int64_t mrfp_conv_stats_final_count(int64_t nblk) { return nblk > kCompactAbove ? kStatGroups : nblk; }


}  // extern "C"
