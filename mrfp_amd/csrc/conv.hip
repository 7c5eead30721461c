// conv.hip -- implicit-GEMM convolution on the gfx950 matrix cores (MFMA), NHWC activations.
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
// by default LDS-DMA (buffer_load ... lds) straight into ONE LDS buffer per workgroup -- no staging registers, no
// ds_write; the 3-5 co-resident workgroups of a CU hide each other's fill latency -- with the register-staged single
// buffer and the double-buffered LDS-DMA variants kept behind MRFP_CONV_DMA.  LDS rows are 128 B with the 16-byte slot
// XOR-swizzled by (row>>1)&7 (applied on the SOURCE side for LDS-DMA) so that the ds_read_b128 fragment reads are
// bank-conflict free.
//
//   bf16 / f16: v_mfma_f32_16x16x32_{bf16,f16} (fp32 accumulate; 32x32x16 in the weight-gradient kernel)  -- bench dtype
//   fp32:       v_mfma_f32_32x32x2_f32   (exact fp32 FMA chains)                                        -- parity dtype
// All share the byte geometry, so there is one kernel template.
//
// Replaces (reference): every nn.Conv2d on the hot path -- Resnet.py:156-161 (Bottleneck),
// deepv3.py:96-112 (ASPP), 200-219 (decoder), 221-237 (HRFP), and their autograd backward.
#include "common.hpp"
#include <stdlib.h>

namespace mrfp {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

struct ConvP {
    const char* x;      // source activation [B,H,W,C]
    const char* w;      // packed weight [N][kchunks] 16-byte chunks
    char* y;            // output [M][ldy]
    const float* bias;  // [N] or null
    const char* addend; // [M][ldy] (T) added to the result in the epilogue, or null (fused gradient accumulation)
    const unsigned char* addend_mask;   // or null: 1 bit per addend element (bit e & 7 of byte e >> 3, e = m*ldy + n): the addend is
                        // taken as 0 where the bit is clear -- the ReLU gate of a residual tail applied while its gradient is
                        // added (16-bit types, N % 8 == 0, dense ldy)
    float* colstats;    // [row blocks][2][ldy] per-channel sum / sum of squares of the stored output, or null
    // BNB kernels (dgrad feeding a BatchNorm backward): the stored output is dL/d(BN output); its BatchNorm-backward
    // statistics  sum g'  and  sum g' * (x - mean)  with  g' = g * [ReLU mask]  are produced here, per row block, instead
    // of by a separate pass over (g, x):  bnx = BN input [M][ldy], bny = BN output for the mask (residual blocks) or null,
    // bnA / bnS = forward apply coefficients for the recomputed mask (x*A+S > 0) or null (no ReLU), bnmean [ldy].
    int stagger8;       // 8-wave tile: waves 4-7 issue their transfers between the two halves of the multiply
    const char* bnx;
    const char* bny;
    const float* bnmean;
    const float* bnA;
    const float* bnS;
    int B, H, W, C;
    int N, ldy;
    int R, S, Ho, Wo;
    int stride, pad_h, pad_w, dil, sstride;
    int M, cpr, kchunks;
    unsigned xbytes, wbytes;   // sizes of x and of the weight pack (buffer descriptors)
};

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                      acc, 0, 0, 0);
    }
};
template <> struct Mma<f16> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // the 16-byte fragment holds 4 consecutive k of one row; MFMA j pairs k = 8q+j (lanes 0-31)
    // with k = 8q+4+j (lanes 32-63) -- the same pairing for A and B, so the sum over k is complete.
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// MRFP_M16 (build switch, default on, 16-bit types only): the forward / dgrad kernel multiplies with
// v_mfma_f32_16x16x32 instead of 32x32x16 -- the same LDS image, LDS bytes, ds_read_b128 count and MFMA cycles per K
// tile, but the chip holds a higher clock on the 16x16 shape (MI355X_MICROARCH.md, DVFS give-back item 7).  Measured
// on one box, same run (tools/ab_m16.sh): 16x256x192x192 3x3 978 -> 1025 TFLOP/s, 16x256x384x384 -> 128 3x3 989 -> 1048,
// short-K layers unchanged, the bench step 63.4 -> 62.5 ms.  `tools/build_variant.sh m32 -DMRFP_M16=0` builds the
// 32x32x16 library for A/B (MRFP_HIP_LIB).
#ifndef MRFP_M16
#define MRFP_M16 1
#endif
constexpr bool kM16 = MRFP_M16 != 0;
// MRFP_WIDE_EP (build switch, default on): tiles with several waves across N (WN > 1) stage a 32-row block of the WHOLE
// workgroup tile in LDS and write full output rows (WN x wider runs, e.g. 256 instead of 64 bytes for the 96x128 tile)
// instead of every wave writing its own 32*TN-column strip.  Small but consistent: bench step 61.8 -> 61.55 ms over three
// alternations on one box (tools/ab_wide.sh) -- L2 write combining already hid most of the narrow runs; the short-K 1x1
// layers stay bound by the fill -> multiply serialisation of their 4-step K loops, not by their stores.
#ifndef MRFP_WIDE_EP
#define MRFP_WIDE_EP 1
#endif
// MRFP_TR (build switch, default OFF, 16-bit types): transposed MFMA + direct 16-byte channel stores in the forward / dgrad
// epilogue instead of the LDS transposition (see TR in conv_igemm_kernel).  It is what makes the persistent B-stationary
// kernel fast (no other workgroup hides its epilogue), but in the generic kernel the co-resident workgroups already do, and
// the 64-byte store runs lose to the wide layout's 256-byte runs: 60.78 vs 60.42 ms per bench step over three alternations
// on one box (`tools/build_variant.sh tr conv -DMRFP_TR=1`, tools/ab_lib.sh).
#ifndef MRFP_TR
#define MRFP_TR 0
#endif
constexpr bool kWideEp = MRFP_WIDE_EP != 0;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <typename T> struct Mma16 {
    static __device__ __forceinline__ void run(f32x4&, const uint4&, const uint4&) {}
};
template <> struct Mma16<bf16> {
    static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma16<f16> {
    static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
};

// two floats -> one dword of two 16-bit values (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: one instruction) and back
typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef __bf16 __attribute__((ext_vector_type(2))) bf16x2_t;
typedef _Float16 __attribute__((ext_vector_type(2))) f16x2_t;
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<bf16>(float a, float b) {
    f32x2 v; v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <> __device__ __forceinline__ unsigned pack2<f16>(float a, float b) {
    f32x2 v; v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
template <typename T> __device__ __forceinline__ void unpack2(unsigned w, float& a, float& b);
template <> __device__ __forceinline__ void unpack2<bf16>(unsigned w, float& a, float& b) {
    a = __uint_as_float(w << 16);
    b = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16>(unsigned w, float& a, float& b) {
    const f32x2 v = __builtin_convertvector(__builtin_bit_cast(f16x2_t, w), f32x2);
    a = v.x;
    b = v.y;
}
// sum over the 16 lanes of a DPP row (lanes 16q .. 16q + 15), result in every lane of the row; fixed order, VALU only
// (__shfl_xor compiles to ds_bpermute_b32 + 4 address instructions per step)
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));    // row_mirror
    return v;
}
template <> __device__ __forceinline__ unsigned pack2<float>(float a, float) { return __float_as_uint(a); }   // (unused: 16-bit epilogue only)
template <> __device__ __forceinline__ void unpack2<float>(unsigned w, float& a, float& b) { a = __uint_as_float(w); b = 0.f; }

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): consecutive logical tiles
// land on the same XCD so the tiles sharing an activation panel hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
// Every tensor the kernels accept is smaller than this many bytes, so a buffer load at an offset >= kOOB is
// out of range and returns zeros: padding, M / N / K tails are all handled by the hardware bounds check of
// buffer_load (no branches, no 64-bit address arithmetic in the K loop).
constexpr unsigned kOOB = 0xF0000000u;

__device__ __forceinline__ uint4 bload(const __amdgpu_buffer_rsrc_t& r, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Asynchronous LDS-DMA for the multi-buffer pipelines.  The compiler's waitcnt pass treats a `buffer_load ... lds` issued
// through the builtin as an LDS store that ANY later ds_read may alias and puts `s_waitcnt vmcnt(0)` in front of the first
// fragment read after it -- which silently serialises "tile k+1 streams in while tile k is multiplied" (the round-1
// double-buffered variants all measured slower for exactly this reason: the ISA of their K loop reads issue, vmcnt(0),
// ds_read).  Issued from inline assembly the transfer is invisible to that pass; ordering is then ours: a counted
// `s_waitcnt vmcnt(N)` (dma_wait) before the workgroup barrier that publishes a stage, and vmcnt(0) before LDS is reused
// by the epilogue.  LDS destination of lane l = m0 + 16*l (m0 = wave-uniform LDS byte address of the 1 KiB piece).
typedef int __attribute__((ext_vector_type(4))) i32x4;
__device__ __forceinline__ i32x4 rsrc_words(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)((unsigned)(a >> 32) & 0xffffu);      // stride 0
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16_async(const i32x4& rsrc, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc));   // (m0 is a reserved register: it cannot be listed as a clobber; nothing
                                                             //  else in these kernels uses it -- checked in the ISA)
}
template <int N> __device__ __forceinline__ void dma_wait() {      // all but the N youngest vector-memory operations done
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__device__ __forceinline__ void settle(uint4& v) {      // forces the compiler to wait for a tracked load right here
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// ALIGNED: the channel count fills whole K tiles (C*sizeof(T) % 128 == 0), so a K tile never straddles a
//          filter tap and the tap (r,s) is tracked in scalar registers; otherwise every thread tracks the tap
//          of its own 16-byte chunk.
// STRIDED: dgrad of a strided convolution (taps exist only where the position divides the source stride).
// WM x WN waves per workgroup, every wave owns TM x TN accumulator blocks of 32x32 (2x2 = 64x64 per wave, 64
// accumulator registers, three waves per SIMD; 4x2 = 128x64 per wave, 128 accumulator registers, two waves per
// SIMD: half the LDS reads and half the L2->LDS bytes per FLOP, and twice the MFMA work per K step to hide the
// global-load latency behind -- used where the problem has enough 256-row tiles to fill the chip).
// DMA: tiles go global -> LDS directly (buffer_load ... lds, no staging registers, no ds_write traffic); the LDS
//      image of one wave-instruction is lane-linear (base + lane*16), so the XOR swizzle is applied to the SOURCE
//      chunk each lane fetches.  NBUF == 2: the DMA of tile k+1 lands while tile k is multiplied; NBUF == 1: fill, barrier,
//      multiply, barrier.
// element access into a 16-byte chunk with COMPILE-TIME indices (keeps the chunk in registers)
template <typename T> __device__ __forceinline__ T chunk_get(const uint4& v, int u);
template <> __device__ __forceinline__ float chunk_get<float>(const uint4& v, int u) {
    const unsigned w = u == 0 ? v.x : u == 1 ? v.y : u == 2 ? v.z : v.w;
    return __uint_as_float(w);
}
template <> __device__ __forceinline__ bf16 chunk_get<bf16>(const uint4& v, int u) {
    const unsigned w = (u >> 1) == 0 ? v.x : (u >> 1) == 1 ? v.y : (u >> 1) == 2 ? v.z : v.w;
    const unsigned short h = (unsigned short)((u & 1) ? (w >> 16) : (w & 0xffffu));
    bf16 r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
template <> __device__ __forceinline__ f16 chunk_get<f16>(const uint4& v, int u) {
    const unsigned w = (u >> 1) == 0 ? v.x : (u >> 1) == 1 ? v.y : (u >> 1) == 2 ? v.z : v.w;
    const unsigned short h = (unsigned short)((u & 1) ? (w >> 16) : (w & 0xffffu));
    f16 r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
template <typename T> __device__ __forceinline__ void chunk_set(uint4& v, int u, T x);
template <> __device__ __forceinline__ void chunk_set<float>(uint4& v, int u, float x) {
    const unsigned w = __float_as_uint(x);
    if (u == 0) v.x = w; else if (u == 1) v.y = w; else if (u == 2) v.z = w; else v.w = w;
}
template <> __device__ __forceinline__ void chunk_set<bf16>(uint4& v, int u, bf16 x) {
    unsigned short h;
    __builtin_memcpy(&h, &x, 2);
    const unsigned sh = (u & 1) ? 16u : 0u, mask = ~(0xffffu << sh), bits = (unsigned)h << sh;
    if ((u >> 1) == 0) v.x = (v.x & mask) | bits;
    else if ((u >> 1) == 1) v.y = (v.y & mask) | bits;
    else if ((u >> 1) == 2) v.z = (v.z & mask) | bits;
    else v.w = (v.w & mask) | bits;
}
template <> __device__ __forceinline__ void chunk_set<f16>(uint4& v, int u, f16 x) {
    unsigned short h;
    __builtin_memcpy(&h, &x, 2);
    const unsigned sh = (u & 1) ? 16u : 0u, mask = ~(0xffffu << sh), bits = (unsigned)h << sh;
    if ((u >> 1) == 0) v.x = (v.x & mask) | bits;
    else if ((u >> 1) == 1) v.y = (v.y & mask) | bits;
    else if ((u >> 1) == 2) v.z = (v.z & mask) | bits;
    else v.w = (v.w & mask) | bits;
}
template <typename T> __device__ __forceinline__ uint4 chunk_add(const uint4& a, const uint4& b) {
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int u = 0; u < 16 / (int)sizeof(T); ++u) chunk_set<T>(r, u, from_f<T>(to_f(chunk_get<T>(a, u)) + to_f(chunk_get<T>(b, u))));
    return r;
}

#ifndef MRFP_EARLY
#define MRFP_EARLY 1
#endif
#ifndef MRFP_RR_HOLD
#define MRFP_RR_HOLD 1         // k steps (of the 6 per filter row) of the row-reuse kernels multiplied after the next fill has been issued
#endif
// 16-bit chunk with the elements whose mask bit is clear set to +0 (bit u of `bits` = element u of the chunk)
__device__ __forceinline__ uint4 gate_chunk16(const uint4& v, unsigned bits) {
    auto w = [&](unsigned word, int u) {
        const unsigned keep = (((bits >> u) & 1u) ? 0x0000ffffu : 0u) | (((bits >> (u + 1)) & 1u) ? 0xffff0000u : 0u);
        return word & keep;
    };
    return make_uint4(w(v.x, 0), w(v.y, 2), w(v.z, 4), w(v.w, 6));
}

#ifndef MRFP_EARLY_FULL
#define MRFP_EARLY_FULL 1
#endif
template <typename T, int WM, int WN, bool ALIGNED, bool STRIDED, int NBUF, int TM, int TN, bool DMA, bool BNB = false, bool RR = false>
__global__ __launch_bounds__(64 * WM * WN, (TM * TN >= 8 || (RR && TM * TN >= 6) ? 2 : 3)) void conv_igemm_kernel(ConvP p) {   // 2nd = waves per SIMD
    const bool g_stagger8 = p.stagger8 != 0;
    constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int SA = BM * 8 / NT, SB = BN * 8 / NT, RSTEP = NT / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA0 = smem;
    char* const sB0 = smem + BM * 128;
    constexpr int BUF = (BM + BN) * 128;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int rbase = t >> 3;
    const int chunk = DMA ? ((t & 7) ^ ((rbase >> 1) & 7)) : (t & 7);   // SOURCE chunk of this thread's slots
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 8;          // first tile row of this wave's DMA pieces
    const int pixbytes = p.C * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
    constexpr bool ASYNC = DMA && NBUF >= 2;         // multi-stage LDS ring filled by asynchronous LDS-DMA (dma16_async)
    const i32x4 xw = rsrc_words(p.x, p.xbytes), ww = rsrc_words(p.w, p.wbytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the ring

    // fixed per-thread gather state for its SA rows of the A tile
    int a_ih0[SA], a_iw0[SA];
    unsigned a_base[SA];   // !STRIDED: byte offset of pixel (b, ih0, iw0) (mod 2^32); STRIDED: offset of image b
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int m = m0 + rbase + i * RSTEP;
        if (m < p.M) {
            const int b = m / (p.Ho * p.Wo), rem = m - b * (p.Ho * p.Wo);
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
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
    int tr = 0, ts = 0, tc = 0;
    if (!ALIGNED) {
        const int rs = chunk / p.cpr;
        tc = chunk - rs * p.cpr;
        tr = rs / p.S;
        ts = rs - tr * p.S;
    }

    typedef __attribute__((address_space(3))) void lds_void;
    auto load_tile = [&](int kt, uint4 (&ra)[SA], uint4 (&rb)[SB], int dbuf) {
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
            if (ASYNC)
                dma16_async(xw, lds0 + (unsigned)(dbuf * BUF + (wrow + i * RSTEP) * 128), voff);
            else if (DMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(sA0 + dbuf * BUF + (wrow + i * RSTEP) * 128), 16,
                                                         (int)voff, 0, 0, 0);
            else
                ra[i] = bload(xr, voff);
        }
        // weight-pack chunk of this K tile: taps are the INNER loop when ALIGNED (see the advance below)
        const unsigned qoff = !qok ? kOOB
                              : ALIGNED ? (unsigned)((tr * p.S + ts) * p.cpr + tc * 8 + chunk) * 16u
                                        : (unsigned)(kt * 8 + chunk) * 16u;
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const unsigned voff = (b_base[i] >= kOOB || !qok) ? kOOB : b_base[i] + qoff;
            if (ASYNC)
                dma16_async(ww, lds0 + (unsigned)(BM * 128 + dbuf * BUF + (wrow + i * RSTEP) * 128), voff);
            else if (DMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(sB0 + dbuf * BUF + (wrow + i * RSTEP) * 128), 16,
                                                         (int)voff, 0, 0, 0);
            else
                rb[i] = bload(wr, voff);
        }
        // advance to the next K tile.  ALIGNED: channel chunk OUTER, filter tap INNER -- the R*S taps of one 64-channel
        // slab re-read (shifted) the same input pixels back to back, so 8 of 9 reads of a 3x3 convolution are served
        // by the XCD's L2 instead of the fabric (the working set of a tap-outer order, 3 image rows x all channels x
        // 32 workgroups, does not fit the 4 MiB L2; measured: the 256x256 kernel was fill-bound at 6.3 TB/s).
        if (ALIGNED) {
            if (++ts == p.S) {
                ts = 0;
                if (++tr == p.R) { tr = 0; ++tc; }
            }
        } else {
            tc += 8;
            while (tc >= p.cpr) {
                tc -= p.cpr;
                if (++ts == p.S) { ts = 0; ++tr; }
            }
        }
    };
    auto store_tile = [&](int buf, const uint4 (&ra)[SA], const uint4 (&rb)[SB]) {
        char* a = sA0 + buf * BUF;
        char* b = sB0 + buf * BUF;
#pragma unroll
        for (int i = 0; i < SA; ++i) *reinterpret_cast<uint4*>(a + lds_off(rbase + i * RSTEP, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < SB; ++i) *reinterpret_cast<uint4*>(b + lds_off(rbase + i * RSTEP, chunk)) = rb[i];
    };

    constexpr bool M16 = kM16 && sizeof(T) == 2;
    // TR: the 16x16x32 MFMAs run TRANSPOSED (D = W_tile * X_tile^T: accumulator rows = output channels, columns = pixels) so
    // that a lane ends up with 8 consecutive CHANNELS of one pixel and stores them straight from its registers -- no
    // transposition of the result through LDS, no 2-byte LDS stores, no epilogue barrier (the epilogue's instruction count,
    // not its bytes, is what the short-K layers pay for: profiles/r02_experiments.md section 3).  Accumulator row r of
    // channel block j stands for channel 32*(j>>1) + 8*(r>>2) + 4*(j&1) + (r&3) of the wave's columns (a permutation of the
    // weight-tile rows, applied in the fragment read address).
    constexpr bool TR = M16 && !BNB && MRFP_TR != 0;
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

    const int nkt = (p.kchunks + 7) >> 3;
    uint4 ra[SA], rb[SB];
    const int lr = lane & 31, lh = lane >> 5;
    const int l15 = lane & 15, lq = lane >> 4;
    auto compute = [&](int buf, int kk0 = 0, int kk1 = 2) {
        const char* a = sA0 + buf * BUF;
        const char* b = sB0 + buf * BUF;
        if constexpr (M16) {
#pragma unroll
            for (int kk = kk0; kk < kk1; ++kk) {      // K step 32 = 4 chunks, one per lane quarter
                const int ch = kk * 4 + lq;
                uint4 fa[2 * TM], fb[2 * TN];
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(a + lds_off(wm * 32 * TM + i * 16 + l15, ch));
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) {
                    const int brow = TR ? 32 * (j >> 1) + 8 * (l15 >> 2) + 4 * (j & 1) + (l15 & 3) : j * 16 + l15;
                    fb[j] = *reinterpret_cast<const uint4*>(b + lds_off(wn * 32 * TN + brow, ch));
                }
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j) {
                        if constexpr (TR) Mma16<T>::run(acc16[i][j], fb[j], fa[i]);
                        else Mma16<T>::run(acc16[i][j], fa[i], fb[j]);
                    }
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
        static_assert(M16 && ALIGNED && !STRIDED && DMA && NBUF == 1 && !BNB, "row-reuse kernels");
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
        unsigned r_j = 0, r_okm = 0;   // per piece i: image row j of its patch row (4 bits each) and "inside the image row" bit
        unsigned r_base[SAR];   // byte offset of its pixel at filter row r = 1 (the centre row), chunk included
#pragma unroll
        for (int i = 0; i < SAR; ++i) {
            const int L = rbase + i * RSTEP;
            const int j = L / PWH, x = L - j * PWH - dil, col = ow0 + x;
            const bool ok = j < RT && col >= 0 && col < p.W;
            r_j |= (unsigned)(j & 15) << (4 * i);
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
    } else if constexpr (DMA && NBUF == 1) {
        // single LDS buffer filled by LDS-DMA: no register staging and no ds_write at all; the fill latency of a
        // workgroup is exposed and hidden only by the other workgroups of the CU (more of them fit: fewer registers)
        if constexpr (M16 && ALIGNED && MRFP_EARLY != 0) {
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
            load_tile(0, ra, rb, 0);
            for (int kt = 0; kt < nkt; ++kt) {
                // EXPLICIT vmcnt(0): across the loop's back edge the compiler puts its own wait for the builtin's transfers AFTER
                // the barrier (`s_waitcnt vmcnt(5); s_barrier; s_waitcnt vmcnt(0); ds_read` in the ISA) -- a wave would pass the
                // barrier with its pieces still in flight and the others would read stale LDS
                dma_wait<0>();
                __syncthreads();      // barrier: tile kt has landed everywhere
                if constexpr (HOLD == 1) compute(0, 0, 1);
#pragma unroll
                for (int h = 0; h < HOLD; ++h) {
                    const int ch = (2 - HOLD + h) * 4 + lq;
#pragma unroll
                    for (int i = 0; i < 2 * TM; ++i) fa[h][i] = *reinterpret_cast<const uint4*>(sA0 + lds_off(wm * 32 * TM + i * 16 + l15, ch));
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j) {
                        const int brow = TR ? 32 * (j >> 1) + 8 * (l15 >> 2) + 4 * (j & 1) + (l15 & 3) : j * 16 + l15;
                        fb[h][j] = *reinterpret_cast<const uint4*>(sB0 + lds_off(wn * 32 * TN + brow, ch));
                    }
                }
                __syncthreads();      // lgkmcnt(0) + barrier: every wave holds its last fragments, the buffer is free
                if (kt + 1 < nkt) load_tile(kt + 1, ra, rb, 0);
#pragma unroll
                for (int h = 0; h < HOLD; ++h)
#pragma unroll
                    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2 * TN; ++j) {
                            if constexpr (TR) Mma16<T>::run(acc16[i][j], fb[h][j], fa[h][i]);
                            else Mma16<T>::run(acc16[i][j], fa[h][i], fb[h][j]);
                        }
            }
            __syncthreads();          // the epilogue reuses the buffer
        } else
        for (int kt = 0; kt < nkt; ++kt) {
            load_tile(kt, ra, rb, 0);
            dma_wait<0>();            // (the compiler's own wait sits here as well; spelled out so that it cannot move behind the barrier)
            __syncthreads();          // vmcnt(0) + barrier: the tile has landed
            compute(0);
            __syncthreads();          // everybody is done reading before the next fill
        }
    } else if constexpr (DMA && NBUF == 2 && WM * WN == 8 && M16) {
        // ---- 8-wave tile, PING-PONG: the two waves of every SIMD (wave w and w + 4) alternate roles phase by phase -- one
        // multiplies from fragments already in its registers while the other reads its next fragments from LDS and issues
        // its LDS-DMA transfers -- so the matrix pipe of a SIMD always has exactly one wave feeding it, instead of both
        // waves multiplying together and then both stalling together (lock-step after the barrier: the round-1 / async-ring
        // variants of this tile, 984 TFLOP/s; a plain half-step stagger already gave +5 %).  One work item = half a K tile
        // (one k step of 32); group X (waves 0-3) reads item i in phase 2i and multiplies it in phase 2i+1, group Y (waves
        // 4-7) one phase later; one s_barrier per phase.  K tile kt sits in slot kt & 1 and is read in phases 4kt .. 4kt+3;
        // tile kt+1 is issued in phases 4kt (X) / 4kt+1 (Y) into the slot tile kt-1 left in phase 4kt-1, every wave waits
        // for its own pieces at the end of phase 4kt+3 (vmcnt(0): issued 2-3 phases earlier), and the barrier that opens
        // phase 4kt+4 publishes them.
        const bool grpY = wave >= 4;
        uint4 fa[2 * TM], fb[2 * TN];
        auto read_frags = [&](int buf, int kk) {
            const char* a = sA0 + buf * BUF;
            const char* b = sB0 + buf * BUF;
            const int ch = kk * 4 + lq;
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(a + lds_off(wm * 32 * TM + i * 16 + l15, ch));
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) {
                const int brow = TR ? 32 * (j >> 1) + 8 * (l15 >> 2) + 4 * (j & 1) + (l15 & 3) : j * 16 + l15;
                fb[j] = *reinterpret_cast<const uint4*>(b + lds_off(wn * 32 * TN + brow, ch));
            }
        };
        auto mma_frags = [&]() {
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) {
                    if constexpr (TR) Mma16<T>::run(acc16[i][j], fb[j], fa[i]);
                    else Mma16<T>::run(acc16[i][j], fa[i], fb[j]);
                }
        };
        load_tile(0, ra, rb, 0);
        dma_wait<0>();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            const bool more = kt + 1 < nkt && p.stagger8 != 2;      // (stagger8 == 2: timing experiment, no transfers issued)
            __builtin_amdgcn_s_barrier();                     // phase 4kt
            if (!grpY) {
                read_frags(buf, 0);
                if (more) load_tile(kt + 1, ra, rb, buf ^ 1);
            } else if (kt > 0) {
                mma_frags();                                  // item (kt-1, 1)
            }
            __builtin_amdgcn_s_barrier();                     // phase 4kt+1
            if (!grpY) {
                mma_frags();
            } else {
                read_frags(buf, 0);
                if (more) load_tile(kt + 1, ra, rb, buf ^ 1);
            }
            __builtin_amdgcn_s_barrier();                     // phase 4kt+2
            if (!grpY) read_frags(buf, 1);
            else mma_frags();
            __builtin_amdgcn_s_barrier();                     // phase 4kt+3
            if (!grpY) mma_frags();
            else read_frags(buf, 1);
            dma_wait<0>();                                    // this wave's pieces of tile kt+1 have landed
        }
        __builtin_amdgcn_s_barrier();                         // phase 4 nkt: Y multiplies its last item
        if (grpY) mma_frags();
        __builtin_amdgcn_s_barrier();                         // everybody is done reading the ring: the epilogue reuses it
    } else if constexpr (DMA) {
        // NBUF-stage LDS ring, NBUF-1 K tiles in flight per workgroup.  Iteration kt: wait until this wave's pieces of
        // tile kt have landed (all but the (NBUF-2)*NP youngest transfers), barrier (everybody's have, and everybody has
        // finished multiplying tile kt-1, whose slot is the one refilled next), issue tile kt+NBUF-1, multiply tile kt.
        // ONE barrier per K tile and no wave ever waits for a transfer it has just issued.
        constexpr int NP = SA + SB;                  // DMA pieces per wave per tile
        static_assert((NBUF - 2) * NP < 64, "vmcnt range");
#pragma unroll
        for (int s = 0; s < NBUF - 1; ++s)
            if (s < nkt) load_tile(s, ra, rb, s);
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + NBUF - 1 <= nkt) dma_wait<(NBUF - 2) * NP>();      // NBUF-2 younger tiles may still be in flight
            else dma_wait<0>();                                        // tail: fewer tiles behind this one
            __builtin_amdgcn_s_barrier();
            // 8-wave tile (two waves per SIMD, same program): waves 4-7 issue their transfers BETWEEN the two halves of the
            // multiply, so a SIMD's two waves are not both in their (MFMA-free) issue phase right after the barrier
            // (MI355X_MICROARCH.md, two waves per SIMD, item 9: stagger by wave number >= 4)
            if (WM * WN == 8 && M16 && g_stagger8 && wave >= 4) {
                compute(kt % NBUF, 0, 1);
                if (kt + NBUF - 1 < nkt) load_tile(kt + NBUF - 1, ra, rb, (kt + NBUF - 1) % NBUF);
                compute(kt % NBUF, 1, 2);
            } else {
                if (kt + NBUF - 1 < nkt) load_tile(kt + NBUF - 1, ra, rb, (kt + NBUF - 1) % NBUF);
                compute(kt % NBUF);
            }
        }
        dma_wait<0>();
        __builtin_amdgcn_s_barrier();             // everybody is done reading the ring: the epilogue reuses it
    } else {
        load_tile(0, ra, rb, 0);
        store_tile(0, ra, rb);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = NBUF == 2 ? (kt & 1) : 0;
            if (kt + 1 < nkt) load_tile(kt + 1, ra, rb, 0);
            compute(buf);
            if (NBUF == 2) {
                if (kt + 1 < nkt) store_tile(buf ^ 1, ra, rb);
                __syncthreads();
            } else {
                // single LDS buffer (half the LDS -> one more workgroup per CU): the prefetched tile waits in
                // registers until every wave has finished reading the current one
                __syncthreads();
                if (kt + 1 < nkt) store_tile(0, ra, rb);
                __syncthreads();
            }
        }
    }

    if constexpr (TR) {
        // ---- direct epilogue: lane (lq = lane>>4, l15 = lane&15) holds, for pixel block ib and channel pair P, the 8 channels
        // nb + 32 P + 8 lq .. + 7 of pixel m0 + wm*32*TM + 16 ib + l15: bias, statistics, addend and ONE 16-byte store.
        T* const yT = reinterpret_cast<T*>(p.y);
        const int nbw = n0 + wn * 32 * TN;
        float bvp[TN][8], cs8[TN][8], cq8[TN][8];
#pragma unroll
        for (int P = 0; P < TN; ++P)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = nbw + 32 * P + 8 * lq + u;
                bvp[P][u] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
                cs8[P][u] = 0.f;
                cq8[P][u] = 0.f;
            }
#pragma unroll
        for (int ib = 0; ib < 2 * TM; ++ib) {
            const int m = m0 + wm * 32 * TM + ib * 16 + l15;
#pragma unroll
            for (int P = 0; P < TN; ++P) {
                const int n = nbw + 32 * P + 8 * lq;
                uint4 v;
                v.x = pack2<T>(acc16[ib][2 * P][0] + bvp[P][0], acc16[ib][2 * P][1] + bvp[P][1]);
                v.y = pack2<T>(acc16[ib][2 * P][2] + bvp[P][2], acc16[ib][2 * P][3] + bvp[P][3]);
                v.z = pack2<T>(acc16[ib][2 * P + 1][0] + bvp[P][4], acc16[ib][2 * P + 1][1] + bvp[P][5]);
                v.w = pack2<T>(acc16[ib][2 * P + 1][2] + bvp[P][6], acc16[ib][2 * P + 1][3] + bvp[P][7]);
                if (p.colstats) {       // BatchNorm statistics of the STORED (rounded) values, fused into the producer
                    float f[8];
                    unpack2<T>(v.x, f[0], f[1]);
                    unpack2<T>(v.y, f[2], f[3]);
                    unpack2<T>(v.z, f[4], f[5]);
                    unpack2<T>(v.w, f[6], f[7]);
                    const float live = m < p.M ? 1.f : 0.f;       // (rows beyond M hold 0 + bias)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float fv = f[u] * live;
                        cs8[P][u] += fv;
                        cq8[P][u] += fv * fv;
                    }
                }
                if (m < p.M && n < p.N) {
                    T* dst = yT + (size_t)m * p.ldy + n;
                    const bool full = n + 8 <= p.N;
                    if (p.addend) {      // y += addend (the skip-connection gradient)
                        const T* ad = reinterpret_cast<const T*>(p.addend) + (size_t)m * p.ldy + n;
                        uint4 av = make_uint4(0u, 0u, 0u, 0u);
                        if (full) {
                            av = *reinterpret_cast<const uint4*>(ad);
                        } else {
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (n + u < p.N) chunk_set<T>(av, u, ad[u]);
                        }
                        if (p.addend_mask) av = gate_chunk16(av, p.addend_mask[((size_t)m * p.ldy + n) >> 3]);
                        v = chunk_add<T>(v, av);
                    }
                    if (full) {
                        *reinterpret_cast<uint4*>(dst) = v;
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (n + u < p.N) dst[u] = chunk_get<T>(v, u);
                    }
                }
            }
        }
        if (p.colstats) {
            // the 16 lanes of a quarter hold the same channels for 16 different pixels: fold them (DPP, fixed order)
            float* out = p.colstats + (size_t)((tile / ntn) * WM + wm) * 2 * p.ldy;
#pragma unroll
            for (int P = 0; P < TN; ++P)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float s2 = row16_sum(cs8[P][u]), q2 = row16_sum(cq8[P][u]);
                    const int n = nbw + 32 * P + 8 * lq + u;
                    if (l15 == 0 && n < p.N) {
                        out[n] = s2;
                        out[p.ldy + n] = q2;
                    }
                }
        }
        return;
    }
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
    // BNB: a lane's read-out chunk column is the same for every row it handles, so its EPC channels' coefficients live in
    // registers and the two sums are per-lane accumulators, folded over the rows at the end
    float b_mu[BNB ? EPC : 1], b_fa[BNB ? EPC : 1], b_fs[BNB ? EPC : 1], b_s0[BNB ? EPC : 1], b_s1[BNB ? EPC : 1];
    if constexpr (BNB) {
        const int n = (WIDE ? n0 : nb) + (lane % CPRW) * EPC;
#pragma unroll
        for (int u = 0; u < EPC; ++u) {
            const bool in = n + u < p.N;
            b_mu[u] = in ? p.bnmean[n + u] : 0.f;
            b_fa[u] = (in && p.bnA) ? p.bnA[n + u] : 0.f;
            b_fs[u] = (in && p.bnS) ? p.bnS[n + u] : 0.f;
            b_s0[u] = 0.f;
            b_s1[u] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
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
            const int m = mb + row, n = (WIDE ? n0 : nb) + ch * EPC;
            if (m < p.M && n < p.N) {
                uint4 v = *reinterpret_cast<const uint4*>(epr + row * EPITCH + ch * 16);
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
                if constexpr (BNB) {      // (host guarantees N % EPC == 0 for these launches: every chunk is full)
                    const uint4 xv = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.bnx) + (size_t)m * p.ldy + n);
                    uint4 yv = make_uint4(0u, 0u, 0u, 0u);
                    if (p.bny) yv = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.bny) + (size_t)m * p.ldy + n);
#pragma unroll
                    for (int u = 0; u < EPC; ++u) {
                        const float xf = to_f(chunk_get<T>(xv, u));
                        const float gate = p.bny ? to_f(chunk_get<T>(yv, u)) : (p.bnA ? xf * b_fa[u] + b_fs[u] : 1.f);
                        const float g = gate > 0.f ? to_f(chunk_get<T>(v, u)) : 0.f;
                        b_s0[u] += g;
                        b_s1[u] += g * (xf - b_mu[u]);
                    }
                }
                if (full) {
                    *reinterpret_cast<uint4*>(dst) = v;
                } else {
#pragma unroll
                    for (int u = 0; u < EPC; ++u)
                        if (n + u < p.N) dst[u] = chunk_get<T>(v, u);
                }
            }
        }
        if constexpr (WIDE) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (BNB) {
        // fold the RPI row-lanes of every chunk column (lanes ch, ch + CPRW, ch + 2 CPRW, ...) in a fixed order
#pragma unroll
        for (int off = CPRW; off < 64; off <<= 1)
#pragma unroll
            for (int u = 0; u < EPC; ++u) {
                b_s0[u] += __shfl_xor(b_s0[u], off, 64);
                b_s1[u] += __shfl_xor(b_s1[u], off, 64);
            }
        float* out = p.colstats + (size_t)((tile / ntn) * WM + wm) * 2 * p.ldy;
        const int ch = lane % CPRW;
        if constexpr (WIDE) {
            // the WN waves of a wave row read out different rows of the same columns: combine them through LDS
            // (fixed order: bitwise reproducible), behind the staging area
            float* red = reinterpret_cast<float*>(smem + WM * 32 * EPITCH) + wm * (WN * 2 * BN);
            if (lane < CPRW) {
#pragma unroll
                for (int u = 0; u < EPC; ++u) {
                    red[(wn * 2 + 0) * BN + ch * EPC + u] = b_s0[u];
                    red[(wn * 2 + 1) * BN + ch * EPC + u] = b_s1[u];
                }
            }
            __syncthreads();
            for (int o = wn * 64 + lane; o < 2 * BN; o += WN * 64) {
                const int st = o / BN, c = o - st * BN;
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) a += red[(w * 2 + st) * BN + c];
                if (n0 + c < p.N) out[st * p.ldy + n0 + c] = a;
            }
        } else if (lane < CPRW) {
#pragma unroll
            for (int u = 0; u < EPC; ++u) {
                const int n = nb + ch * EPC + u;
                if (n < p.N) {
                    out[n] = b_s0[u];
                    out[p.ldy + n] = b_s1[u];
                }
            }
        }
    } else if (p.colstats) {
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

template <typename T, int WM, int WN, bool ALIGNED, bool STRIDED, int NBUF, int TM, int TN, bool DMA, bool BNB = false>
static int launch_igemm_nb(const ConvP& p, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    // epilogue staging (+16 bytes of pitch per row, also in the wide layout) + the cross-wave fold of the BNB sums
    constexpr int EP = WM * WN * 32 * (32 * TN * (int)sizeof(T) + 16) + (BNB ? WM * WN * 2 * BN * 4 : 0);
    const int lds = NBUF * (BM + BN) * 128 > EP ? NBUF * (BM + BN) * 128 : EP;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, WM, WN, ALIGNED, STRIDED, NBUF, TM, TN, DMA, BNB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, ALIGNED, STRIDED, NBUF, TM, TN, DMA, BNB>), dim3((unsigned)tiles),
                       dim3(64 * WM * WN), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T, int WM, int WN, bool ALIGNED, bool STRIDED, int TM, int TN>
static int launch_igemm(const ConvP& p, hipStream_t st) {
    // Staging of the K tiles (MRFP_CONV_DMA, A/B measurements): 0 = register staging, one LDS buffer; 3 (default) = LDS-DMA
    // everywhere: the 4-wave tiles fill ONE buffer with the builtin (fill, barrier, multiply, barrier; the 3-5 co-resident
    // workgroups of a CU hide each other's fill latency), the 8-wave tile runs the asynchronous ring.
    // MRFP_CONV_NBUF=2: every tile runs a 2-stage ring filled by asynchronous LDS-DMA (conv_igemm_kernel).  Measured in
    // round 2 (tools/ab_nbuf.sh, profiles/r02_experiments.md): -5 % on every shape (3 stages: -25 %) -- a second stage
    // costs a co-resident workgroup for the same bytes in flight (every byte in flight needs LDS to land in: tools/fill_micro.hip).
    static int dma = -1, nbuf = -1;
    if (dma < 0) {
        const char* e = getenv("MRFP_CONV_DMA");
        dma = e ? atoi(e) : 3;
        e = getenv("MRFP_CONV_NBUF");
        nbuf = e ? atoi(e) : 0;
    }
    if constexpr (ALIGNED) {      // dgrad + BatchNorm-backward statistics (mrfp_conv_dgrad_bnstats): aligned channel counts only
        if (p.bnx) {
            if constexpr (sizeof(T) == 2 && TM * TN >= 8) return launch_igemm_nb<T, WM, WN, ALIGNED, STRIDED, 2, TM, TN, true, true>(p, st);
            else return launch_igemm_nb<T, WM, WN, ALIGNED, STRIDED, 1, TM, TN, true, true>(p, st);
        }
    }
    if (dma == 0) return launch_igemm_nb<T, WM, WN, ALIGNED, STRIDED, 1, TM, TN, false>(p, st);   // register staging
    if constexpr (sizeof(T) == 2) {
        if (nbuf >= 2 || TM * TN >= 8) return launch_igemm_nb<T, WM, WN, ALIGNED, STRIDED, 2, TM, TN, true>(p, st);
    }
    return launch_igemm_nb<T, WM, WN, ALIGNED, STRIDED, 1, TM, TN, true>(p, st);
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
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, WM, WN, true, false, 1, TM, TN, true, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, true, false, 1, TM, TN, true, false, true>), dim3((unsigned)tiles),
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

static int g_big = -1;
static bool use_big_tile(const ConvP& p, int esz) {
    if (g_big < 0) {
        // MRFP_CONV_BIGTILE=1 enables the 256x256 / 8-wave / double-buffered LDS-DMA tile.  It won on long-K layers while
        // the 4-wave tiles staged through registers (+8..15 %); since those use single-buffer LDS-DMA (4 workgroups per
        // CU) the 128x128 tile is as fast or faster on every layer of the bench workload (41.2 vs 41.7 ms of
        // convolutions per step), so it is off by default.
        const char* e = getenv("MRFP_CONV_BIGTILE");
        g_big = e ? atoi(e) : 0;
    }
    if (!g_big || esz != 2 || p.N <= 64) return false;
    if (g_big == 2) return true;         // A/B measurements: the 8-wave tile wherever it is legal
    const int64_t m256 = (p.M + 255) / 256, n256 = (p.N + 255) / 256;
    const int nkt = (p.kchunks + 7) >> 3;
    return p.N >= 192 && n256 * 256 - p.N <= 64 && nkt >= 18 && m256 * n256 >= 448;
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
    if (!g_t96 || p.N <= 64 || use_big_tile(p, esz)) return false;
    if (g_t96 == 2) return true;      // A/B measurements: 96-row tile wherever it is legal
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
    if (!g_t192 || esz != 2 || p.N <= 64 || use_big_tile(p, esz)) return false;
    if (g_t192 == 2) return true;        // A/B measurements: 192-row tile wherever it is legal
    if (use_tile96(p, esz)) return false;
    const int nkt = (p.kchunks + 7) >> 3;
    return nkt >= 9 && ((p.M + 191) / 192) * ((p.N + 127) / 128) >= 2048;
}

// (measured and dropped in round 2: a 160x128 tile, 4 waves x 160x32, 71 instead of 55 FLOP per fill byte and 462 tiles
//  = one round at two workgroups per CU for the M = 36 864, N = 256 layers: 55.9 vs 49.8 us on the 3x3 layer, 31.5 vs
//  29.3 us on the 1024 -> 256 pointwise layer -- fewer co-resident workgroups cost more than the fill bytes save)

// Row-reuse kernels: the tile (192 or 96 rows; 0 = not applicable) a launch runs on.  MRFP_CONV_RR=0: off; 1: where the plain
// kernel would run the 192x128 tile; 2 (default): also instead of the 128x128 / 96x128 tiles; 3: wherever 192 rows fit the image
// geometry; 4: also the 96-row variant.
static int g_rr = -1;
static int rr_tile(const ConvP& p, int esz) {
    if (g_rr < 0) {
        const char* e = getenv("MRFP_CONV_RR");
        g_rr = e ? atoi(e) : 2;
    }
    if (!g_rr || esz != 2 || p.bnx || (p.N > 64 && use_big_tile(p, esz))) return 0;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.sstride != 1 || p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.dil < 1 || p.dil > 2 || p.pad_h != p.dil || p.pad_w != p.dil) return 0;
    if ((p.cpr & 7) != 0 || (p.W & 15) != 0) return 0;
    auto fits = [&](int BM) {
        if ((p.H * p.W) % BM != 0 || (p.W % BM != 0 && BM % p.W != 0)) return false;
        const int PW = p.W < BM ? p.W : BM, RT = BM / PW;
        return RT * (PW + 2 * p.dil) <= BM + 32;
    };
    // Where it pays (bench.py --dump-convs with the switch off / on, several boxes): long K (C >= 256: at least 12 patch fills
    // per tile to amortise the 76 KB prologue), at least one full round of tiles at two workgroups per CU, and an N that does not
    // waste most of its last 128-column tile.  Lost: M = 36 864, 256 -> 256 (384 tiles: 825 vs 880 TFLOP/s against the 96x128
    // tile), C = 128 (862 vs 912), N = 304 (918 vs 977).  Mode 3 lifts these restrictions (A/B runs).
    if (p.N <= 64) {
        // 192 x 64 tile (4 waves x 96 x 32) for the N <= 64 layers (C = 64 / 128: 9-18 K tiles of the 256 x 64 tile become 3-6
        // patch fills).  Measured and OFF (MRFP_CONV_RR64=1 enables it): 544 vs 490 us at 16x128x384^2 -> 64, 77.7 vs 67.9 us at
        // 16x64x192^2 -> 64 -- the 96 x 32 wave tile reads a third more fragments per MFMA than the plain kernel's 64 x 64, and
        // two to six fills per tile do not amortise the patch prologue.
        static int rr64 = -1;
        if (rr64 < 0) { const char* e = getenv("MRFP_CONV_RR64"); rr64 = e ? atoi(e) : 0; }
        return (rr64 && g_rr >= 2 && p.N > 32 && fits(192) && p.M / 192 >= 768) ? 19264 : 0;
    }
    const int64_t t192 = (int64_t)(p.M / 192) * ((p.N + 127) / 128);
    const bool pays = p.C >= 256 && t192 >= 512 && ((p.N + 127) / 128) * 128 - p.N <= 64;
    if (fits(192) && (g_rr >= 3 || (pays && (g_rr >= 2 || use_tile192(p, esz))))) return 192;
    if (g_rr >= 4 && fits(96)) return 96;
    return 0;
}

// =============================================================================================
// B-stationary kernel for the short-K 1x1 convolutions (16-bit types, K = C <= 256, stride 1): Y[M, N] = X[M, K] W[N, K]^T.
//
// The generic kernel re-fetches a 128-column weight tile with every 96..192-row tile (55..77 FLOP per byte of L2 -> LDS
// fill, the path that bounds it, profiles/r02_experiments.md), and with 2-4 K tiles per workgroup its prologue / epilogue
// weigh as much as its K loop (M = 36 864, 256 -> 1024: 412 TFLOP/s).  Here a workgroup is PERSISTENT over a range of
// 64-row M tiles of one 128-column panel:
//   * the weights never touch LDS: each wave keeps its 32 columns x K of the panel as MFMA B fragments in REGISTERS
//     (K = 256: 64 VGPRs), loaded once per workgroup;
//   * only X moves: a 64 x K tile per step, asynchronous LDS-DMA into a ring (counted vmcnt, one barrier per tile, the
//     transfer of tile t + NST - 1 issued before tile t is multiplied), i.e. 128 instead of 55..77 FLOP per fill byte;
//   * the MFMA runs transposed (accumulator rows = channels), so every lane stores 8 consecutive channels of a pixel
//     straight from its accumulators: no transposition through LDS, no epilogue barrier; the optional fused per-channel
//     statistics and skip-gradient addend of the generic kernel are kept.
// Two workgroups per CU interleave one's epilogue with the other's multiplies.
// Layout of an X tile in LDS: KB blocks of [64 rows][128 bytes], each with the generic kernel's XOR swizzle, so the
// fragment reads are the generic kernel's (bank-conflict free).
// =============================================================================================

struct BsP {
    const char* x;       // [M][K] dense (K = C elements)
    const char* w;       // forward pack [N][K]
    char* y;             // [M][ldy]
    const char* addend;  // [M][ldy] or null
    const unsigned char* addend_mask;   // 1 bit per addend element or null (ConvP::addend_mask)
    float* colstats;     // [ceil(M/64)][2][ldy] or null
    int M, N, ldy;
    int tiles;           // ceil(M / 64)
    int panels;          // ceil(N / 128)
    int chunks;          // M-tile ranges per panel (grid = panels * chunks)
    unsigned xbytes, wbytes, ybytes;
};

struct HasPrev { static constexpr bool value = true; };
struct NoPrev { static constexpr bool value = false; };
template <typename T, int KB, int NST, bool STATS, bool ADD>
__global__ __launch_bounds__(256, 2) void conv1x1_bstat_kernel(BsP p) {
    constexpr int ROWB = KB * 128;              // bytes of one row of X (K elements)
    constexpr int STAGE = KB * 64 * 128;        // one 64-row tile
    constexpr int NP = KB * 2;                  // DMA pieces (8 rows x 128 B) per wave per tile: KB blocks x 8 pieces / 4 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);          // scalar: the DMA's LDS address (m0) must be uniform
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 xw = rsrc_words(p.x, p.xbytes);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend ? p.addend : p.y), 0, (int)p.ybytes, 0x00020000);
    // vmcnt bookkeeping: the output stores of a tile are issued AFTER the transfer of a later tile and are counted by the
    // same in-order counter, so "tile t has landed" = all but the younger transfers AND the younger tiles' stores are
    // done.  The stores are therefore UNCONDITIONAL buffer stores (rows / columns outside the tensor get an out-of-range
    // offset and are dropped by the bounds check): exactly ST of them per wave per tile, whatever the tile covers.
    constexpr int ST = 4;

    // work: block b -> (chunk, panel) with the panels of one chunk (same rows of X) on one XCD (blocks b, b + 8, ... share an
    // L2): b = xcd + 8 * (panel + panels * c2), chunk = xcd + 8 * c2
    const int b = blockIdx.x, xcd = b & 7, rest = b >> 3;
    const int panel = rest % p.panels, chunk = xcd + 8 * (rest / p.panels);
    if (chunk >= p.chunks) return;              // (uniform per workgroup; no barrier has been passed yet)
    const int per = (p.tiles + p.chunks - 1) / p.chunks;
    const int t0 = chunk * per, t1 = min(p.tiles, t0 + per);
    if (t0 >= t1) return;
    const int n0 = panel * 128 + wave * 32;     // this wave's 32 columns

    // ---- the weights: this wave's fragments for every K step, straight into registers ------------------------------------
    // The MFMA runs TRANSPOSED (D = W_tile * X_tile^T: accumulator rows = output channels, columns = pixels), so a lane
    // ends up with consecutive CHANNELS of one pixel and stores them directly -- no transposition of the result through
    // LDS, no 2-byte LDS stores, no epilogue barrier.  Accumulator row r = 4*(lane>>4) + e of channel block j is mapped to
    // channel 8*(r>>2) + 4*j + (r&3) of the wave's 32 columns (a permutation of the weight rows, free at load time): a
    // lane's 2 x 4 values are then channels 8*(lane>>4) .. +7 of its pixel = one 16-byte store.
    uint4 fw[KB * 2][2];                        // [k step of 32][channel block]
#pragma unroll
    for (int ks = 0; ks < KB * 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 8 * (l15 >> 2) + 4 * j + (l15 & 3);           // the channel accumulator row l15 of block j stands for
            fw[ks][j] = bload(wr, n < p.N ? (unsigned)n * (unsigned)ROWB + (unsigned)(ks * 64 + lq * 16) : kOOB);
        }
    // The weights must have ARRIVED before the tile loop: otherwise the compiler waits for them at their first use INSIDE the
    // loop body, with `s_waitcnt vmcnt(15) ... vmcnt(0)` spread over the multiplies -- on every iteration, where they drain
    // the transfers of the next tile and the stores of the previous one (seen in the ISA; it cost a third of the kernel).
#pragma unroll
    for (int ks = 0; ks < KB * 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) settle(fw[ks][j]);

    // ---- X tile DMA: piece q of this wave covers block kb = q / 2, rows (q & 1) * 32 + wave * 8 .. + 7 --------------------
    unsigned src[NP], dst[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int kb = q >> 1, row = (q & 1) * 32 + wave * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);                     // source-side swizzle (LDS image is lane-linear)
        src[q] = (unsigned)row * (unsigned)ROWB + (unsigned)(kb * 128 + ch * 16);
        dst[q] = (unsigned)(kb * 64 * 128 + ((q & 1) * 32 + wave * 8) * 128);
    }
    auto issue = [&](int tile, int slot) {
        const unsigned base = (unsigned)tile * 64u * (unsigned)ROWB;      // rows beyond M lie beyond xbytes: zero fill
#pragma unroll
        for (int q = 0; q < NP; ++q) dma16_async(xw, lds0 + (unsigned)(slot * STAGE) + dst[q], base + src[q]);
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (t0 + s < t1) issue(t0 + s, s);

    // (a half-tile start delay for the second resident workgroup of every CU was measured: no effect)
    const int nl = n0 + 8 * lq;                 // first of this lane's 8 output channels
    float cs[8], cq[8];                         // per-channel sum / sum of squares over every tile of this workgroup
#pragma unroll
    for (int u = 0; u < 8; ++u) { cs[u] = 0.f; cq[u] = 0.f; }

    // SOFTWARE PIPELINE over the tiles: the epilogue of tile t-1 (conversions, statistics, addend, stores: ~150 vector
    // instructions) is written BETWEEN the k steps of tile t, in one basic block with its multiplies (STATS / ADD are
    // template parameters, so no branch splits the block): an MFMA holds the SIMD's vector issue for 8 of its 16 cycles, the
    // other 8 take two ordinary instructions for free, so the epilogue rides in the multiply's issue shadow instead of
    // running after it with the matrix pipe idle.  Two accumulator sets (A / B) alternate.
    auto wait_tile = [&](int tile) {
        // younger than the transfer of `tile` (issued in iteration tile-NST+1, before that iteration's body): the NST-2
        // transfers of the tiles behind it and the stores issued in the bodies of iterations tile-NST+1 .. tile-1 -- which,
        // one tile late in this pipeline, are those of tiles tile-NST .. tile-2: NST-1 batches, all of them real only from
        // tile t0+NST on (the body of t0 has no epilogue in it; counting its absent stores let the second tile of a range be
        // read before its last pieces had landed)
        if (tile - t0 >= NST && tile + NST - 1 <= t1) dma_wait<(NST - 2) * NP + (NST - 1) * ST>();
        else dma_wait<0>();                                               // first / last tiles of the range: fewer behind it
        __builtin_amdgcn_s_barrier();                                     // tile landed everywhere; tile - 1 fully consumed
    };
    auto fetch_addend = [&](int tile, uint4 (&av)[4], unsigned (&am)[4]) {
        // skip-gradient addend: fetched before the multiplies of its tile, consumed one tile later (a load issued in the
        // epilogue would be waited for right there: 58 us against 32 us per dgrad launch).  Compiler-tracked on purpose: an
        // untracked inline-asm load is WRONG here (the compiler may copy the destination registers before the data has
        // arrived; the bitwise-reproducibility test caught it).
        if constexpr (ADD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = tile * 64 + i * 16 + l15;
                av[i] = bload(ar, (m < p.M && nl < p.N) ? ((unsigned)m * (unsigned)p.ldy + (unsigned)nl) * 2u : kOOB);
                // the gate bits of these 8 channels (all ones without a mask); a plain tracked load, issued with the addend
                am[i] = (p.addend_mask && m < p.M && nl < p.N) ? p.addend_mask[((size_t)m * p.ldy + nl) >> 3] : 0xffu;
            }
        }
    };
    // one quarter (16 pixels) of the epilogue of `tile` from accumulator set acc
    auto epilogue_part = [&](int tile, int i, const f32x4 (&acc)[4][2], const uint4 (&av)[4], const unsigned (&am)[4]) {
        const int m = tile * 64 + i * 16 + l15;
        const bool ok = m < p.M && nl < p.N;                              // (N % 8 == 0 for this kernel: chunks are whole)
        uint4 v;
        v.x = pack2<T>(acc[i][0][0], acc[i][0][1]);
        v.y = pack2<T>(acc[i][0][2], acc[i][0][3]);
        v.z = pack2<T>(acc[i][1][0], acc[i][1][1]);
        v.w = pack2<T>(acc[i][1][2], acc[i][1][3]);
        if constexpr (STATS) {
            // statistics of the STORED (rounded) values, before the addend.  Rows beyond M were zero-filled by the transfer's
            // bounds check, so they add exactly 0: no mask.
            float f[8];
            unpack2<T>(v.x, f[0], f[1]);
            unpack2<T>(v.y, f[2], f[3]);
            unpack2<T>(v.z, f[4], f[5]);
            unpack2<T>(v.w, f[6], f[7]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cs[u] += f[u];
                cq[u] += f[u] * f[u];
            }
        }
        if constexpr (ADD) v = chunk_add<T>(v, gate_chunk16(av[i], am[i]));
        const unsigned off = ok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)nl) * 2u : kOOB;
        u32x4 dv;
        dv.x = v.x; dv.y = v.y; dv.z = v.z; dv.w = v.w;
        __builtin_amdgcn_raw_buffer_store_b128(dv, yr, (int)off, 0, 0);
    };
    // multiplies of `tile` into acc; when prev >= 0 the epilogue of tile `prev` (accumulators pacc, addend pav) in between
    auto body = [&](int tile, f32x4 (&acc)[4][2], auto has_prev, const f32x4 (&pacc)[4][2], const uint4 (&pav)[4], const unsigned (&pam)[4]) {
        const int prev = tile - 1;
        const char* a = ring + ((tile - t0) % NST) * STAGE;
        constexpr int KS = KB * 2;
        // fragment reads run ONE K STEP AHEAD of the multiplies that use them (two register sets): left to itself the compiler
        // issues each pair of reads two MFMAs before their use and waits for them (`s_waitcnt lgkmcnt(1)` after every
        // second MFMA in the ISA), i.e. an LDS latency per 32 cycles of matrix work
        uint4 fx[2][4];
        auto read_x = [&](int ks, uint4 (&f)[4]) {
            const char* ab = a + (ks >> 1) * (64 * 128);
            const int ch = (ks & 1) * 4 + lq;
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const uint4*>(ab + lds_off(i * 16 + l15, ch));
        };
        read_x(0, fx[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) read_x(ks + 1, fx[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);    // (the scheduler otherwise sinks the reads back to just before their use)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (ks == 0) {                // first k step: accumulate onto a literal zero (no register clearing)
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc[i][j] = z;
                    }
                    Mma16<T>::run(acc[i][j], fw[ks][j], fx[ks & 1][i]);
                }
            if constexpr (decltype(has_prev)::value) {
                // the 4 epilogue quarters of the previous tile, spread over the k steps
                if constexpr (KS >= 4) { if (ks % (KS / 4) == 0) epilogue_part(prev, ks / (KS / 4), pacc, pav, pam); }
                else { epilogue_part(prev, 2 * ks, pacc, pav, pam); epilogue_part(prev, 2 * ks + 1, pacc, pav, pam); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x4 accA[4][2], accB[4][2];
    uint4 avA[4], avB[4];
    unsigned amA[4] = {0xffu, 0xffu, 0xffu, 0xffu}, amB[4] = {0xffu, 0xffu, 0xffu, 0xffu};
#pragma unroll
    for (int i = 0; i < 4; ++i) { avA[i] = make_uint4(0u, 0u, 0u, 0u); avB[i] = make_uint4(0u, 0u, 0u, 0u); }
    // first tile: multiplies only
    wait_tile(t0);
    fetch_addend(t0, avA, amA);
    if (t0 + NST - 1 < t1) issue(t0 + NST - 1, (NST - 1) % NST);
    body(t0, accA, NoPrev{}, accB, avB, amB);
    int tile = t0 + 1;
    for (; tile + 1 < t1; tile += 2) {
        wait_tile(tile);
        fetch_addend(tile, avB, amB);
        if (tile + NST - 1 < t1) issue(tile + NST - 1, (tile - t0 + NST - 1) % NST);
        body(tile, accB, HasPrev{}, accA, avA, amA);
        wait_tile(tile + 1);
        fetch_addend(tile + 1, avA, amA);
        if (tile + NST < t1) issue(tile + NST, (tile + 1 - t0 + NST - 1) % NST);
        body(tile + 1, accA, HasPrev{}, accB, avB, amB);
    }
    if (tile < t1) {                            // an even number of tiles: one more B step, then its own epilogue
        wait_tile(tile);
        fetch_addend(tile, avB, amB);
        if (tile + NST - 1 < t1) issue(tile + NST - 1, (tile - t0 + NST - 1) % NST);
        body(tile, accB, HasPrev{}, accA, avA, amA);
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue_part(tile, i, accB, avB, amB);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue_part(t1 - 1, i, accA, avA, amA);
    }
    if constexpr (STATS) {
        // ONE statistics row block per workgroup range (all its tiles): the 16 lanes of a quarter hold the same 8 channels
        // for 16 different pixels -- fold them (DPP, fixed order) and let lane 0 of the quarter write
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

static int g_bstat = -1;
// the layers the B-stationary kernel takes (MRFP_CONV_BSTAT=0: generic kernel everywhere, for A/B runs)
static bool use_bstat(const ConvP& p, int esz, bool has_bias) {
    if (g_bstat < 0) {
        const char* e = getenv("MRFP_CONV_BSTAT");
        g_bstat = e ? atoi(e) : 1;
    }
    if (!g_bstat || esz != 2 || has_bias || p.bnx || (p.colstats && p.addend)) return false;
    if (p.R != 1 || p.S != 1 || p.stride != 1 || p.sstride != 1 || p.pad_h != 0 || p.pad_w != 0) return false;
    if (p.Ho != p.H || p.Wo != p.W || p.N < 128 || (p.N & 7) != 0) return false;
    if ((int64_t)p.M * p.ldy * esz >= (int64_t)kOOB) return false;        // the output is addressed through a buffer descriptor
    const int rowb = p.C * esz;
    return rowb == 128 || rowb == 256 || rowb == 512;
}

// M-tile ranges per panel: two workgroups per CU, each at least 4 tiles long (the weights are loaded once per workgroup),
// a multiple of 8 (the XCD mapping), and no range empty.  Also the number of statistics row blocks of such a launch.
static int bstat_chunks(int M, int N) {
    const int tiles = (M + 63) / 64, panels = (N + 127) / 128;
    int chunks = 512 / panels;
    while (chunks > 8 && (tiles + chunks - 1) / chunks < 4) chunks -= 8;
    chunks = (chunks + 7) / 8 * 8;
    if (chunks < 8) chunks = 8;
    const int per = (tiles + chunks - 1) / chunks;
    return (tiles + per - 1) / per;              // ranges that actually hold tiles (the trailing ones would be empty)
}

template <typename T, int KB, bool STATS, bool ADD>
static int launch_bstat(const ConvP& c, hipStream_t st) {
    constexpr int NST = KB == 4 ? 2 : 3;                          // K = 256: 2 x 32 KB stages (two workgroups per CU)
    constexpr int STAGE = KB * 64 * 128;
    const int lds = NST * STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_bstat_kernel<T, KB, NST, STATS, ADD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    BsP p;
    p.x = c.x; p.w = c.w; p.y = c.y; p.addend = c.addend; p.addend_mask = c.addend_mask; p.colstats = c.colstats;
    p.M = c.M; p.N = c.N; p.ldy = c.ldy;
    p.tiles = (c.M + 63) / 64;
    p.panels = (c.N + 127) / 128;
    p.chunks = bstat_chunks(c.M, c.N);
    const int chunks = (p.chunks + 7) / 8 * 8;           // grid: whole groups of 8 (workgroups past p.chunks exit at once)
    p.xbytes = c.xbytes; p.wbytes = c.wbytes; p.ybytes = (unsigned)((int64_t)c.M * c.ldy * 2);
    {   // timing-only diagnostics (MRFP_DEBUG_DROP bit 2: drop the output stores)
        static int dbg = -1;
        if (dbg < 0) { const char* e = getenv("MRFP_DEBUG_DROP"); dbg = e ? atoi(e) : 0; }
        if (dbg & 4) p.ybytes = 0;
    }
    hipLaunchKernelGGL((conv1x1_bstat_kernel<T, KB, NST, STATS, ADD>), dim3((unsigned)(p.panels * chunks)), dim3(256), lds, st, p);
    MRFP_LAUNCH_CHECK();
    return 0;
}

template <typename T, bool STATS, bool ADD>
static int run_bstat_v(const ConvP& p, hipStream_t st) {
    const int kb = p.C * 2 / 128;
    return kb == 1 ? launch_bstat<T, 1, STATS, ADD>(p, st) : kb == 2 ? launch_bstat<T, 2, STATS, ADD>(p, st) : launch_bstat<T, 4, STATS, ADD>(p, st);
}
template <typename T>
static int run_bstat(const ConvP& p, hipStream_t st) {
    // forward launches carry the fused statistics, dgrad launches the skip-gradient addend; never both in this network
    if (p.colstats && p.addend) return -1;
    if (p.colstats) return run_bstat_v<T, true, false>(p, st);
    if (p.addend) return run_bstat_v<T, false, true>(p, st);
    return run_bstat_v<T, false, false>(p, st);
}

// number of statistics row blocks (= m-tiles x wave rows) the epilogue of a forward launch writes
static int64_t stats_row_blocks(const ConvP& p, int esz) {
    if (use_bstat(p, esz, false)) return (int64_t)bstat_chunks(p.M, p.N);               // conv1x1_bstat_kernel: one per workgroup range
    if (const int rr = rr_tile(p, esz)) return rr == 96 ? (int64_t)(p.M / 96) : (int64_t)(p.M / 192) * 2;   // row-reuse kernels: 96-row (1 wave row) or 192-row tiles
    if (p.N > 64 && use_tile192(p, esz)) return (int64_t)((p.M + 191) / 192) * 2;      // <2,2,3,2>: 192-row tile, 2 wave rows
    if (p.N <= 64) return (int64_t)((p.M + 255) / 256) * 4;          // <4,1,2,2>: 256-row tile, 4 wave rows
    if (use_big_tile(p, esz)) return (int64_t)((p.M + 255) / 256) * 2;  // <2,4,4,2>: 256-row tile, 2 wave rows
    if (use_tile96(p, esz)) return (int64_t)((p.M + 95) / 96);          // <1,4,3,1>: 96-row tile, 1 wave row
    return (int64_t)((p.M + 127) / 128) * 2;                          // <2,2,2,2>: 128-row tile, 2 wave rows
}

template <typename T>
static int run_igemm(const ConvP& p, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (use_bstat(p, 2, p.bias != nullptr)) return run_bstat<T>(p, st);
    }
    if constexpr (sizeof(T) == 2) {
        if (const int rr = rr_tile(p, 2))
            return rr == 192 ? launch_igemm_rr<T, 2, 2, 3, 2>(p, st) : rr == 96 ? launch_igemm_rr<T, 1, 4, 3, 1>(p, st) : launch_igemm_rr<T, 2, 2, 3, 1>(p, st);
    }
    if (p.N <= 64) return pick_igemm<T, 4, 1, 2, 2>(p, st);
    // 256x256 tile (8 waves x 128x64, LDS-DMA, one workgroup per CU).  Measured per shape on MI355X inside the
    // bench workload (bench.py --dump-convs): +8..10 % on long-K 3x3 layers with >= 2 full rounds of tiles
    // (998 vs 913 TF/s at 16x192x192x256->256), but -25 % with ~1 round (M = 36 864), -20 % on short-K 1x1
    // layers (fill / drain dominate) and on N that wastes most of the second 256-column tile; a 256x128
    // 4-wave variant (254 VGPRs) lost 25 % everywhere and was dropped.
    if constexpr (sizeof(T) == 2) {      // 16-bit types only (no fp32 instantiation of these two tiles)
        if (use_big_tile(p, (int)sizeof(T))) return pick_igemm<T, 2, 4, 4, 2>(p, st);
    }
    // (measured and dropped: a two-wave 96x128 variant, 2 x (96x64), 216 registers, fewer LDS reads per MFMA: 5-25 %
    //  slower; a 256x128 8-wave LDS-DMA tile for the N = 128 layers: 753 vs 803 TF/s at 16x384x384x256 -> 128)
    if constexpr (sizeof(T) == 2) {
        if (use_tile192(p, (int)sizeof(T))) return pick_igemm<T, 2, 2, 3, 2>(p, st);
    }
    if (use_tile96(p, (int)sizeof(T))) return pick_igemm<T, 1, 4, 3, 1>(p, st);
    return pick_igemm<T, 2, 2, 2, 2>(p, st);
}

// ---------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 master  ->  forward pack Wf[Npad][R][S][Cpad]  (T)
//                                   ->  dgrad   pack Wd[Cin][R][S][Npad] with taps flipped (T)
// (pad channels are zero).  One thread per destination element.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int N, int C,
                                   int R, int S, int Npad, int Cpad) {
    const int64_t nf = (int64_t)Npad * R * S * Cpad, nd = (int64_t)C * R * S * Npad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nf) {
            if (!wf) continue;
            const int c = (int)(i % Cpad);
            int64_t rest = i / Cpad;
            const int s = (int)(rest % S); rest /= S;
            const int r = (int)(rest % R);
            const int n = (int)(rest / R);
            const float v = (n < N && c < C) ? w[(((int64_t)n * C + c) * R + r) * S + s] : 0.f;
            wf[i] = from_f<T>(v);
        } else {
            if (!wd) continue;
            const int64_t k = i - nf;
            const int n = (int)(k % Npad);
            int64_t rest = k / Npad;
            const int s = (int)(rest % S); rest /= S;
            const int r = (int)(rest % R);
            const int c = (int)(rest / R);
            const float v = (n < N) ? w[(((int64_t)n * C + c) * R + (R - 1 - r)) * S + (S - 1 - s)] : 0.f;
            wd[k] = from_f<T>(v);
        }
    }
}

// All weight packs of a model in ONE launch (after the optimizer step): jobs[] and the exclusive prefix of their element
// counts live in device memory; every thread finds its job by binary search and packs one element as above.
struct PackJob {
    const float* w;
    void* wf;
    void* wd;
    int N, C, R, S, Npad, Cpad;
};
// One workgroup = one (job, 64 output channels n, 8 input channels c) brick: it reads the 64 runs w[n][c0..c0+7][:][:] of
// 8*R*S contiguous floats (coalesced), keeps the brick in LDS, and writes both packs from there with the channel index
// that is contiguous in the pack as the fastest thread index: wf[n][r][s][c0..c0+7] (16-byte runs) and
// wd[c][R-1-r][S-1-s][n0..n0+63] (128-byte runs).  prefix[] counts bricks (plus the pad bricks that zero the padding).
constexpr int kBrickN = 64, kBrickC = 8, kBrickRSMax = 9;      // filters up to 3x3 (18.7 KB of LDS); larger ones pack per layer
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackJob* __restrict__ jobs,
                                                                    const int64_t* __restrict__ prefix, int njobs, int64_t total) {
    __shared__ float brick[kBrickN][kBrickC * kBrickRSMax + 1];      // (>= 64 + 1 columns: the pointwise bricks fit)
    const int64_t wg = blockIdx.x;
    int lo = 0, hi = njobs;                 // largest j with prefix[j] <= wg  (uniform: every lane does the same search)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= wg) lo = mid; else hi = mid;
    }
    const PackJob jb = jobs[lo];
    const int N = jb.N, C = jb.C, R = jb.R, S = jb.S, Npad = jb.Npad, Cpad = jb.Cpad, RS = R * S;
    // input channels per brick: 8 for 3x3 filters, 64 for pointwise ones (the same 64 x 72-float brick either way; with 8
    // channels a pointwise brick was 2 KB of work behind a 7-step binary search: 240 us for the 124 packs of ResNet-101)
    const int bc = RS == 1 ? kBrickC * 8 : kBrickC;
    const int ncb = (Cpad + bc - 1) / bc;
    const int local = (int)(wg - prefix[lo]);
    const int n0 = (local / ncb) * kBrickN, c0 = (local % ncb) * bc;
    const int run = bc * RS;                // floats per n in this brick (contiguous in w when c0 + bc <= C)
    const int t = threadIdx.x;
    for (int e = t; e < kBrickN * run; e += 256) {
        const int nn = e / run, k = e - nn * run;          // k = cc*RS + rs
        const int n = n0 + nn, c = c0 + k / RS;
        brick[nn][k] = (n < N && c < C) ? jb.w[((size_t)n * C + c0) * RS + k] : 0.f;
    }
    __syncthreads();
    T* wf = reinterpret_cast<T*>(jb.wf);
    T* wd = reinterpret_cast<T*>(jb.wd);
    // forward pack: wf[n][rs][c]  (c fastest over bc consecutive threads)
    for (int e = t; e < kBrickN * run; e += 256) {
        const int cc = e % bc, rest = e / bc;
        const int rs = rest % RS, nn = rest / RS;
        const int n = n0 + nn, c = c0 + cc;
        if (n < Npad && c < Cpad) wf[((size_t)n * RS + rs) * Cpad + c] = from_f<T>(brick[nn][cc * RS + rs]);
    }
    // dgrad pack: wd[c][flipped rs][n]  (n fastest over 64 consecutive threads); only real input channels have rows
    for (int e = t; e < kBrickN * run; e += 256) {
        const int nn = e % kBrickN, rest = e / kBrickN;
        const int rs = rest % RS, cc = rest / RS;
        const int n = n0 + nn, c = c0 + cc;
        if (n < Npad && c < C) wd[((size_t)c * RS + (RS - 1 - rs)) * Npad + n] = from_f<T>(brick[nn][cc * RS + rs]);
    }
}

// network input: NCHW fp32 [B,C,H,W] -> NHWC T [B,H,W,Cpad] (pad channels zero)
template <typename T>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int H, int W, int Cpad) {
    const int64_t npix = (int64_t)B * H * W;
    constexpr int EPC = 16 / (int)sizeof(T);
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = pix / ((int64_t)H * W), hw = pix % ((int64_t)H * W);
        if (Cpad == EPC && C <= EPC) {        // the network input (3 -> one 16-byte chunk per pixel): ONE store instead of Cpad
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int c = 0; c < EPC; ++c)
                if (c < C) chunk_set<T>(v, c, from_f<T>(x[(b * C + c) * (int64_t)H * W + hw]));
            *reinterpret_cast<uint4*>(y + pix * Cpad) = v;
            continue;
        }
        for (int c = 0; c < Cpad; ++c) {
            const float v = c < C ? x[(b * C + c) * (int64_t)H * W + hw] : 0.f;
            y[pix * Cpad + c] = from_f<T>(v);
        }
    }
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

static int conv_fwd_impl(const void* x, const void* wpack, const float* bias, void* y, int dtype, int64_t B, int64_t H,
                         int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                         int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride, const void* addend,
                         float* colstats, const void* bnx, const void* bny, const float* bnmean, const float* bnA,
                         const float* bnS, void* stream, const void* addend_mask = nullptr) {
    MRFP_CHECK(!addend || aligned16(addend), "conv_fwd: addend must be 16-byte aligned");
    MRFP_CHECK(!addend_mask || (addend && dtype != MRFP_F32 && (N & 7) == 0 && ldy == N && !bnx),
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
    {
        static int stg = -1;
        if (stg < 0) { const char* e = getenv("MRFP_CONV_STAGGER8"); stg = e ? atoi(e) : 1; }
        p.stagger8 = stg;
    }
    p.x = (const char*)x; p.w = (const char*)wpack; p.y = (char*)y; p.bias = bias; p.addend = (const char*)addend; p.colstats = colstats;
    p.addend_mask = (const unsigned char*)addend_mask;
    p.bnx = (const char*)bnx; p.bny = (const char*)bny; p.bnmean = bnmean; p.bnA = bnA; p.bnS = bnS;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.N = (int)N; p.ldy = (int)ldy;
    p.R = (int)R; p.S = (int)S; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.stride = (int)stride; p.pad_h = (int)pad_h; p.pad_w = (int)pad_w; p.dil = (int)dil; p.sstride = (int)sstride;
    p.M = (int)(B * Ho * Wo); p.cpr = (int)(C * esz / 16); p.kchunks = (int)(R * S * p.cpr);
    if (bnx) {
        MRFP_CHECK(colstats && bnmean && aligned16(bnx) && (!bny || aligned16(bny)), "conv_dgrad_bnstats: bad arguments");
        MRFP_CHECK((p.cpr & 7) == 0 && (N * esz) % 16 == 0 && ldy == N,
                   "conv_dgrad_bnstats: needs C*esz %% 128 == 0 and a dense output of whole chunks (query mrfp_conv_dgrad_bnstats_ok)");
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
        p.M = (int)(bc * Ho * Wo);
        p.x = (const char*)x + b0 * img;
        p.y = (char*)y + m0 * ldy * esz;
        p.addend = addend ? (const char*)addend + m0 * ldy * esz : nullptr;
        p.addend_mask = addend_mask ? (const unsigned char*)addend_mask + ((m0 * ldy) >> 3) : nullptr;
        p.bnx = bnx ? (const char*)bnx + m0 * ldy * esz : nullptr;
        p.bny = bny ? (const char*)bny + m0 * ldy * esz : nullptr;
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

/* 1 when a convolution over an input of B images of `image_bytes` bytes runs as ONE launch (fused statistics available) */
int mrfp_conv_single_launch(int64_t B, int64_t image_bytes) {
    return image_bytes > 0 && image_bytes < (int64_t)kOOB && B <= (int64_t)(kOOB - 1) / image_bytes;
}

int mrfp_conv_fwd(const void* x, const void* wpack, const float* bias, void* y, int dtype, int64_t B, int64_t H,
                  int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                  int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride, const void* addend,
                  float* colstats, void* stream) {
    return conv_fwd_impl(x, wpack, bias, y, dtype, B, H, W, C, N, ldy, R, S, Ho, Wo, stride, pad_h, pad_w, dil, sstride, addend,
                         colstats, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

int mrfp_conv_fwd_gated(const void* x, const void* wpack, const float* bias, void* y, int dtype, int64_t B, int64_t H,
                        int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                        int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride, const void* addend,
                        const void* addend_mask, void* stream) {
    MRFP_CHECK(addend && addend_mask, "conv_fwd_gated: addend and its gate mask are required");
    return conv_fwd_impl(x, wpack, bias, y, dtype, B, H, W, C, N, ldy, R, S, Ho, Wo, stride, pad_h, pad_w, dil, sstride, addend,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream, addend_mask);
}

int mrfp_conv_dgrad_bnstats_ok(int dtype, int64_t C, int64_t N) {
    const int esz = dtype == MRFP_F32 ? 4 : 2;
    return (C * esz) % 128 == 0 && (N * esz) % 16 == 0;
}

int mrfp_conv_dgrad_bnstats(const void* dy, const void* wpack, void* dx, int dtype, int64_t B, int64_t H, int64_t W,
                            int64_t C, int64_t N, int64_t R, int64_t S, int64_t Ho, int64_t Wo, int64_t pad_h, int64_t pad_w,
                            int64_t dil, int64_t sstride, const void* addend, const void* bn_x, const void* bn_y,
                            const float* bn_mean, const float* bn_fA, const float* bn_fS, float* bnstats, void* stream) {
    MRFP_CHECK(bn_x && bn_mean && bnstats, "conv_dgrad_bnstats: bn_x / bn_mean / bnstats are required");
    return conv_fwd_impl(dy, wpack, nullptr, dx, dtype, B, H, W, C, N, N, R, S, Ho, Wo, 1, pad_h, pad_w, dil, sstride, addend,
                         bnstats, bn_x, bn_y, bn_mean, bn_fA, bn_fS, stream);
}

int64_t mrfp_conv_stats_blocks(int dtype, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t R, int64_t S, int64_t Ho,
                               int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride, int64_t bn_bwd) {
    // the SAME geometry the launch will see (conv_fwd_impl): the kernel choice -- and with it the number of row blocks -- looks at
    // the filter, stride, padding, dilation and image size, not only at M, N, C
    ConvP p;
    const int esz = dtype == MRFP_F32 ? 4 : 2;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.N = (int)N; p.ldy = (int)N;
    p.R = (int)R; p.S = (int)S; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.stride = (int)stride; p.pad_h = (int)pad_h; p.pad_w = (int)pad_w; p.dil = (int)dil; p.sstride = (int)sstride;
    p.M = (int)(B * Ho * Wo); p.cpr = (int)(C * esz / 16); p.kchunks = (int)(R * S * p.cpr);
    p.bnx = bn_bwd ? "" : nullptr;      // (only tested against null by the kernel choice)
    p.bias = nullptr; p.colstats = nullptr; p.addend = nullptr; p.addend_mask = nullptr;
    return stats_row_blocks(p, esz);
}
/* rows the caller must allocate for `colstats` (row blocks + the compacted groups) */
int64_t mrfp_conv_stats_rows(int64_t nblk) { return nblk > kCompactAbove ? nblk + kStatGroups : nblk; }
/* where the rows to hand to mrfp_bn_finalize start, and how many there are */
int64_t mrfp_conv_stats_final_first(int64_t nblk) { return nblk > kCompactAbove ? nblk : 0; }
int64_t mrfp_conv_stats_final_count(int64_t nblk) { return nblk > kCompactAbove ? kStatGroups : nblk; }

int mrfp_pack_weight(const float* w, void* wf, void* wd, int dtype, int64_t N, int64_t C, int64_t R, int64_t S,
                     int64_t Npad, int64_t Cpad, void* stream) {
    MRFP_CHECK(w && (wf || wd) && N > 0 && C > 0 && R > 0 && S > 0 && Npad >= N && Cpad >= C, "pack_weight: bad arguments");
    const int64_t total = Npad * R * S * Cpad + C * R * S * Npad;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((pack_weight_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (float*)wf, (float*)wd, (int)N, (int)C, (int)R, (int)S, (int)Npad, (int)Cpad);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((pack_weight_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (bf16*)wf, (bf16*)wd, (int)N, (int)C, (int)R, (int)S, (int)Npad, (int)Cpad);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((pack_weight_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                           (f16*)wf, (f16*)wd, (int)N, (int)C, (int)R, (int)S, (int)Npad, (int)Cpad);
    else
        MRFP_CHECK(false, "pack_weight: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_pack_weights_batched(const void* jobs, const int64_t* prefix, int64_t njobs, int64_t total, int dtype, void* stream) {
    MRFP_CHECK(jobs && prefix && njobs > 0 && total > 0, "pack_weights_batched: bad arguments");   /* jobs with R*S > 9 are the caller's error */
    const int64_t blocks = total;           // one workgroup per brick
    MRFP_CHECK(blocks < (1LL << 31), "pack_weights_batched: too many bricks");
    const mrfp::PackJob* jb = (const mrfp::PackJob*)jobs;
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((pack_weights_batched_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jb, prefix, (int)njobs, total);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((pack_weights_batched_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jb, prefix, (int)njobs, total);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((pack_weights_batched_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jb, prefix, (int)njobs, total);
    else
        MRFP_CHECK(false, "pack_weights_batched: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_nchw_to_nhwc_pad(const float* x, void* y, int dtype, int64_t B, int64_t C, int64_t H, int64_t W, int64_t Cpad,
                          void* stream) {
    MRFP_CHECK(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_to_nhwc_pad: bad arguments");
    int64_t blocks = (B * H * W + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == MRFP_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                           (float*)y, (int)B, (int)C, (int)H, (int)W, (int)Cpad);
    else if (dtype == MRFP_BF16)
        hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                           (bf16*)y, (int)B, (int)C, (int)H, (int)W, (int)Cpad);
    else if (dtype == MRFP_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                           (f16*)y, (int)B, (int)C, (int)H, (int)W, (int)Cpad);
    else
        MRFP_CHECK(false, "nchw_to_nhwc_pad: unknown dtype %d", dtype);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

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
    unsigned xbytes, dybytes;
    FastDiv div_hw, div_w;   // by Ho*Wo and by Wo
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
template <typename T> struct WgTile { static constexpr int BKP = 64; };   // pixels per K' tile
template <> struct WgTile<float> { static constexpr int BKP = 32; };

// DENSE: pointwise convolution (1x1, stride 1, no padding): X is a dense [pixel][channel] matrix like dY, so its slots advance
// by a constant and need no (ih, iw) bookkeeping -- 60 of the 85 VALU and 40 of the 64 SALU instructions of a K' tile in
// the general kernel, on layers (M = 36 864 bottleneck 1x1) that are bound by exactly that instruction stream
// (profiles/r02_experiments.md section 4: 31 us with or without any global traffic, MFMA time 11 us).
template <typename T, int WM, int WN, bool DMA, bool DENSE = false>
__global__ __launch_bounds__(256, (DMA ? 4 : 3)) void conv_wgrad_kernel(WgP p) {   // 2nd = waves per SIMD
    static_assert(WM * WN == 4, "4 waves");
    static_assert(!DMA || sizeof(T) == 2, "LDS-DMA layout is for the 16-bit types");
    constexpr int BKP = WgTile<T>::BKP;
    constexpr int EPC = 16 / (int)sizeof(T);                 // elements per 16-byte chunk
    constexpr int CY = 64 * WM / EPC, CX = 64 * WN / EPC;    // chunks per tile row
    constexpr int SY = BKP * CY / 256, SX = BKP * CX / 256;  // slots per thread
    constexpr int PY = 64 * WM * (int)sizeof(T) + (DMA ? 0 : 64), PX = 64 * WN * (int)sizeof(T) + (DMA ? 0 : 64);   // row pitches
    constexpr int NBY = 2 * WM, NBX = 2 * WN;                 // 64-byte blocks per row (DMA layout)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ty = smem;               // single LDS buffer: the next tile waits in registers
    char* const tx = smem + BKP * PY;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    typedef __attribute__((address_space(3))) void lds_void;
    const int wave1k = __builtin_amdgcn_readfirstlane(wave) * 1024;      // this wave's first DMA piece (1 KiB each)
    const int ntq = (p.Q + 64 * WN - 1) / (64 * WN);
    // 1-D grid over (split, tile), split-major, dealt to the XCDs in contiguous chunks: the tiles of one split read the
    // same pixel range of x and dy, so they share one L2 instead of pulling those rows into all eight
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int split = work / p.tiles, tile = work - split * p.tiles;
    const int n0 = (tile / ntq) * 64 * WM, q0 = (tile % ntq) * 64 * WN;
    const int kbeg = split * p.klen;
    const int kend = min(p.M, kbeg + p.klen);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dybytes, 0x00020000);

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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
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
        constexpr int HOLD = (!DENSE && WM == 1) ? 1 : MRFP_WGRAD_HOLD;     // (the 64x256 gather variant spills with two held k steps)
        if (nkt > 0) load_tile(kbeg, ry, rx);
        for (int kt = 0; kt < nkt; ++kt) {
            dma_wait<0>();            // explicit: across the back edge the compiler's own wait lands behind the barrier
            __syncthreads();          // the tile has landed everywhere
#pragma unroll
            for (int ks = 0; ks < KSN - HOLD; ++ks) {
                uint4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = wg_read_sw<NBY>(ty, wm * 64 + i * 32, ks * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = wg_read_sw<NBX>(tx, wn * 64 + j * 32, ks * 16, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
            }
            uint4 ha[HOLD > 0 ? HOLD : 1][2], hb[HOLD > 0 ? HOLD : 1][2];
#pragma unroll
            for (int h = 0; h < HOLD; ++h) {
#pragma unroll
                for (int i = 0; i < 2; ++i) ha[h][i] = wg_read_sw<NBY>(ty, wm * 64 + i * 32, (KSN - HOLD + h) * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) hb[h][j] = wg_read_sw<NBX>(tx, wn * 64 + j * 32, (KSN - HOLD + h) * 16, lane);
            }
            __syncthreads();          // lgkmcnt(0) + barrier: everybody is done reading, the buffer is free
            if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * BKP, ry, rx);
#pragma unroll
            for (int h = 0; h < HOLD; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i)
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

    float* out = p.slab + (size_t)split * p.N * p.Q;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qq = q0 + wn * 64 + j * 32 + lr;
        if (qq >= p.Q) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < p.N) out[(size_t)n * p.Q + qq] = acc[i][j][e];
            }
    }
}

// dW[n][c][r][s] (OIHW fp32, c < Ctrue) = sum_z slab[z][n][(r*S+s)*C + c]
// Threads walk the SLAB order (4 consecutive channels each, 16-byte loads: the slabs are ~20x the size of dW, so
// their reads are the ones that must coalesce); the OIHW stores are 4-byte scattered but few.
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int N, int Q, int C, int Ctrue, int RS,
                                    float* __restrict__ dw, int accumulate) {
    const int64_t total4 = (int64_t)N * Q / 4, NQ = (int64_t)N * Q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = i * 4;
        const int n = (int)(j / Q), q = (int)(j - (int64_t)n * Q);
        const int rs = q / C, c = q - rs * C;
        if (c >= Ctrue) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int z = 0; z < splits; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)z * NQ + j);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
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
}

template <typename T, int WM, int WN, bool DMA, bool DENSE = false>
static int launch_wgrad_v(const WgP& p, int splits, hipStream_t st) {
    constexpr int PY = 64 * WM * (int)sizeof(T) + (DMA ? 0 : 64), PX = 64 * WN * (int)sizeof(T) + (DMA ? 0 : 64);
    const int lds = WgTile<T>::BKP * (PY + PX);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, WM, WN, DMA, DENSE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    WgP q = p;
    q.tiles = ((p.N + 64 * WM - 1) / (64 * WM)) * ((p.Q + 64 * WN - 1) / (64 * WN));
    hipLaunchKernelGGL((conv_wgrad_kernel<T, WM, WN, DMA, DENSE>), dim3((unsigned)(q.tiles * splits)), dim3(256), lds, st, q);
    MRFP_LAUNCH_CHECK();
    return 0;
}

// MRFP_WGRAD_DMA=0 keeps register staging for the 16-bit types (A/B measurements); fp32 always stages in registers
template <typename T, int WM, int WN>
static int launch_wgrad(const WgP& p, int splits, hipStream_t st) {
    static int dma = -1;
    if (dma < 0) { const char* e = getenv("MRFP_WGRAD_DMA"); dma = e ? atoi(e) : 1; }
    if (sizeof(T) == 2 && dma) {
        static int dense = -1;
        if (dense < 0) { const char* e = getenv("MRFP_WGRAD_DENSE"); dense = e ? atoi(e) : 1; }
        const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 && p.H == p.Ho && p.W == p.Wo;
        if (dense && pointwise) return launch_wgrad_v<T, WM, WN, sizeof(T) == 2, true>(p, splits, st);
        return launch_wgrad_v<T, WM, WN, sizeof(T) == 2>(p, splits, st);
    }
    return launch_wgrad_v<T, WM, WN, false>(p, splits, st);
}

static void wgrad_plan(int64_t M, int64_t N, int64_t Q, int bkp, int& wm, int& splits, int& klen, int64_t cap = 0) {
    wm = N <= 64 ? 1 : 2;
    const int wn = 4 / wm;
    const int64_t tiles = ((N + 64 * wm - 1) / (64 * wm)) * ((Q + 64 * wn - 1) / (64 * wn));
    const int64_t nkt = (M + bkp - 1) / bkp;
    // Split count from a small cost model (times in us, constants fitted to the bench workload's per-launch timings):
    //   a CU that holds w = ceil(tiles*sp/256) workgroups needs w * (K' tiles per split) tile-steps of ~0.84 us, divided
    //   by a latency-hiding efficiency (1 workgroup per CU 0.6, 2 -> 0.85, >= 3 -> 1); every split adds an fp32 slab of dW
    //   that is written once and read once by the reduction (~3 TB/s).
    // MRFP_WGRAD_WGS=<n> replaces the model by "about n workgroups" (A/B measurements).
    static int target = -1;
    if (target < 0) {
        const char* e = getenv("MRFP_WGRAD_WGS");
        target = e ? atoi(e) : 0;
    }
    int64_t sp = 1;
    if (target > 0) {
        sp = target / tiles;
    } else {
        double best = 1e30;
        int64_t smax = 1024 / tiles > 96 ? 1024 / tiles : 96;      // few tiles: enough splits to fill the chip
        if (smax > nkt) smax = nkt;
        for (int64_t c = 1; c <= smax; ++c) {
            const int64_t w = (tiles * c + 255) / 256, iters = (nkt + c - 1) / c;
            const double eff = w >= 3 ? 1.0 : w == 2 ? 0.85 : 0.6;
            const double cost = (double)w * (double)iters * 0.84 / eff + (double)c * ((double)N * (double)Q * 8.0 / 3.0e6);
            if (cost < best * 0.999) { best = cost; sp = c; }
        }
    }
    if (sp < 1) sp = 1;
    if (sp > nkt) sp = nkt;
    if (cap > 0 && sp > cap) sp = cap;         // (a batch range of a larger call: the workspace was sized for the whole call)
    int64_t per = (nkt + sp - 1) / sp;        // K' tiles per split
    sp = (nkt + per - 1) / per;
    splits = (int)sp;
    klen = (int)(per * bkp);
}

}  // namespace mrfp

extern "C" {

int64_t mrfp_conv_wgrad_ws_bytes(int64_t M, int64_t N, int64_t Q) {
    int wm, s32, s64, klen;
    mrfp::wgrad_plan(M, N, Q, 32, wm, s32, klen);         // fp32 K' tile
    mrfp::wgrad_plan(M, N, Q, 64, wm, s64, klen);         // bf16 K' tile
    return (int64_t)(s32 > s64 ? s32 : s64) * N * Q * 4;
}

int mrfp_conv_wgrad(const void* x, const void* dy, float* dw, void* ws, int dtype, int64_t B, int64_t H, int64_t W,
                    int64_t C, int64_t Ctrue, int64_t N, int64_t ldn, int64_t R, int64_t S, int64_t Ho, int64_t Wo,
                    int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, void* stream) {
    MRFP_CHECK(x && dy && dw && ws && B > 0 && H > 0 && W > 0 && C > 0 && N > 0 && R > 0 && S > 0 && Ho > 0 && Wo > 0,
               "conv_wgrad: bad arguments");
    MRFP_CHECK(dtype == MRFP_F32 || dtype == MRFP_BF16 || dtype == MRFP_F16, "conv_wgrad: unknown dtype %d", dtype);
    const int esz = dtype == MRFP_F32 ? 4 : 2;
    MRFP_CHECK((C * esz) % 16 == 0 && (ldn * esz) % 16 == 0 && ldn >= N && Ctrue <= C,
               "conv_wgrad: channel counts must make 16-byte chunks (C=%lld ldn=%lld)", (long long)C, (long long)ldn);
    MRFP_CHECK(aligned16(x) && aligned16(dy), "conv_wgrad: x / dy must be 16-byte aligned");
    MRFP_CHECK(B * Ho * Wo < (1LL << 31), "conv_wgrad: tensor too large");
    WgP p;
    p.x = (const char*)x; p.dy = (const char*)dy; p.slab = (float*)ws;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.N = (int)N; p.ldn = (int)ldn;
    p.R = (int)R; p.S = (int)S; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.stride = (int)stride; p.pad_h = (int)pad_h; p.pad_w = (int)pad_w; p.dil = (int)dil;
    p.Q = (int)(R * S * C);
    // Both operands are read through 32-bit buffer-descriptor offsets: an activation above kOOB bytes is walked in batch
    // ranges, every range one wgrad + reduction pair on the stream, the later ones adding to dw (fixed order: reproducible)
    const int64_t ximg = H * W * C * esz, yimg = Ho * Wo * ldn * esz;
    MRFP_CHECK(ximg < (int64_t)kOOB && yimg < (int64_t)kOOB, "conv_wgrad: one image exceeds the 3.75 GB buffer-descriptor range");
    int64_t bmax = (int64_t)(kOOB - 1) / (ximg > yimg ? ximg : yimg);
    int dbg_drop = 0;
    {   // timing-only diagnostics (see mrfp_conv_fwd)
        static int dbg = -1;
        if (dbg < 0) { const char* e = getenv("MRFP_DEBUG_DROP"); dbg = e ? atoi(e) : 0; }
        dbg_drop = dbg;
    }
    p.div_hw = make_fastdiv((unsigned)(Ho * Wo)); p.div_w = make_fastdiv((unsigned)Wo);
    hipStream_t st = (hipStream_t)stream;
    int wm0, cap, klen0;
    wgrad_plan(B * Ho * Wo, N, p.Q, dtype == MRFP_F32 ? 32 : 64, wm0, cap, klen0);     // what `ws` was sized for
    for (int64_t b0 = 0; b0 < B; b0 += bmax) {
        const int64_t bc = B - b0 < bmax ? B - b0 : bmax;
        p.B = (int)bc;
        p.M = (int)(bc * Ho * Wo);
        p.x = (const char*)x + b0 * ximg;
        p.dy = (const char*)dy + b0 * yimg;
        p.xbytes = (dbg_drop & 1) ? 0u : (unsigned)(bc * ximg);
        p.dybytes = (dbg_drop & 2) ? 0u : (unsigned)(bc * yimg);
        int wm, splits;
        wgrad_plan(p.M, N, p.Q, dtype == MRFP_F32 ? 32 : 64, wm, splits, p.klen, cap);
        int rc;
        if (dtype == MRFP_F32) rc = wm == 1 ? launch_wgrad<float, 1, 4>(p, splits, st) : launch_wgrad<float, 2, 2>(p, splits, st);
        else if (dtype == MRFP_F16) rc = wm == 1 ? launch_wgrad<f16, 1, 4>(p, splits, st) : launch_wgrad<f16, 2, 2>(p, splits, st);
        else rc = wm == 1 ? launch_wgrad<bf16, 1, 4>(p, splits, st) : launch_wgrad<bf16, 2, 2>(p, splits, st);
        if (rc) return rc;
        const int64_t total4 = N * (int64_t)p.Q / 4;          // Q = R*S*C and C*esz % 16 == 0  =>  Q % 4 == 0
        int64_t blocks = (total4 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, splits, (int)N,
                           p.Q, (int)C, (int)Ctrue, (int)(R * S), dw, b0 > 0 ? 1 : 0);
        MRFP_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
