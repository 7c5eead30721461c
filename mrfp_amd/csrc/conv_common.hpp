// conv_common.hpp -- helpers shared by the convolution translation units (conv_igemm.hip, conv_pw.hip, conv_wgrad.hip,
// conv_pack.hip): MFMA wrappers, 16-byte chunk access, LDS swizzle, XCD remap, buffer loads / LDS-DMA, launch parameters.
#pragma once
#include "common.hpp"
#include <stdlib.h>

// MRFP_VMCNT0 (build switch, default off; `tools/build_variant.sh vm0 "" -DMRFP_VMCNT0=1`): every COUNTED s_waitcnt vmcnt(N) of
// the asynchronous LDS-DMA rings becomes vmcnt(0).  tests/test_conv_gpu.py runs the bench workload's launch shapes through both
// builds and asserts bitwise equality -- an under-wait then shows as a mismatch against the conservative build instead of only
// as run-to-run noise.
#ifndef MRFP_VMCNT0
#define MRFP_VMCNT0 0
#endif

// MRFP_CLOCK_STAMP (diagnostic build only, default off; tools/clock_stamp.py builds libmrfp_hip_clk.so with it): every convolution
// workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) once at its start and once before its epilogue and
// writes the two differences to a buffer of its own that nothing else reads -- in-kernel clock = d(memtime) / d(memrealtime) x
// 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  In the product build no stamp executes.
#ifndef MRFP_CLOCK_STAMP
#define MRFP_CLOCK_STAMP 0
#endif

namespace mrfp {

// Launch geometry of the PERSISTENT kernels (conv_c64.hip, conv_wg3.hip, conv_wg1.hip): their grids are "one round" of resident
// workgroups, and the statistics / slab buffers their callers allocate (c64_spi, wg3_splits_bound, wg1_splits_bound,
// mrfp_conv_wgrad*_ws_bytes) are sized FROM these grids -- every such rule derives from the two constants below, and each *_run checks
// its chosen split / span count against the bound function before it launches (an error, never a write past a buffer sized elsewhere).
//   (kCUs: common.hpp)
constexpr int kGrid1PerCU = kCUs;               // one resident workgroup per CU (512-register waves: conv_c64 at 128 channels, conv_wg1)
constexpr int kGrid2PerCU = 2 * kCUs;           // two per CU (conv_c64 at 64 channels, conv_wg3)

constexpr int kStampSlots = 4096;
#if MRFP_CLOCK_STAMP
struct ClockStamp {
    unsigned long long t0, r0;
    __device__ __forceinline__ void begin() {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) alone: the loop's own LDS waits stay counted
    }
    __device__ __forceinline__ void end(unsigned long long (*buf)[2]) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (threadIdx.x == 0) {
            buf[blockIdx.x % kStampSlots][0] = t1 - t0;
            buf[blockIdx.x % kStampSlots][1] = r1 - r0;
        }
    }
};
#define MRFP_STAMP_DECL(name) __device__ unsigned long long name[::mrfp::kStampSlots][2];
#define MRFP_STAMP_BEGIN() ::mrfp::ClockStamp stamp__; stamp__.begin()
#define MRFP_STAMP_END(name) stamp__.end(name)
#define MRFP_STAMP_READ(name, out, n) (hipMemcpyFromSymbol(out, HIP_SYMBOL(name), (size_t)(n) * 16) == hipSuccess ? 0 : -1)
#else
#define MRFP_STAMP_DECL(name)
#define MRFP_STAMP_BEGIN()
#define MRFP_STAMP_END(name)
#define MRFP_STAMP_READ(name, out, n) (-1)
#endif
int stamps_igemm(unsigned long long* out, int n);     // conv_igemm.hip
int stamps_pw(unsigned long long* out, int n);        // conv_pw.hip
int stamps_wgrad(unsigned long long* out, int n);     // conv_wgrad.hip
int stamps_pwk(unsigned long long* out, int n);       // conv_pwk.hip: per-phase cycle sums of the two-group long-K pointwise kernel

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
    const unsigned char* rowweight;     // or null: [M rounded up to the tile height] one byte per output row (pixel): the statistics count
                        // that row `rowweight[m]` times -- the multiplicity of the pixel in a nearest-neighbour resize of the output
                        // (the HRFP stages, reference deepv3.py:320-327: the BatchNorm behind the resize takes its statistics from here)
    int B, H, W, C;
    int N, ldy;
    int R, S, Ho, Wo;
    int stride, pad_h, pad_w, dil, sstride;
    int classed;        // STRIDED launches (sstride == 2): rows are enumerated parity class by parity class (conv_igemm.hip)
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

// 16-bit forward / dgrad kernels multiply with v_mfma_f32_16x16x32 instead of 32x32x16: the same LDS image, LDS bytes,
// ds_read_b128 count and MFMA cycles per K tile, but the chip holds a higher clock on the 16x16 shape (MI355X_MICROARCH.md,
// DVFS give-back item 7; measured in round 1: 978 -> 1025 TFLOP/s at 16x256x192x192 3x3, profiles/r02_experiments.md).
// Tiles with several waves across N stage a 32-row block of the WHOLE workgroup tile in LDS and write full output rows
// (kWideEp: 256- instead of 64-byte runs on the 96x128 tile).  The measured-slower alternatives of rounds 1-2 (32x32x16 in
// the forward kernels, per-wave strips, a transposed-MFMA direct-store epilogue in the generic kernel, register staging,
// multi-stage asynchronous rings, the 256x256 8-wave tile, BatchNorm-backward statistics in the dgrad epilogue) were
// removed from the library in round 3; their numbers are in profiles/r02_experiments.md.
constexpr bool kM16 = true;
constexpr bool kWideEp = true;
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
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MRFP_VMCNT0 ? 0 : N) : "memory");
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
#define MRFP_EARLY_FULL 1      // bit 0: the 96x128 tile holds BOTH k steps of a K tile across the next fill, bit 1: the 128x128 tile too
#endif

// ---- pointwise (1x1, stride 1) short-K kernels, conv_pw.hip ----------------------------------------------------------------
bool pw_applicable(const ConvP& p, int esz);          // does run_igemm hand this launch to conv_pw.hip?
int64_t pw_stats_blocks(const ConvP& p);              // statistics row blocks such a launch writes
int64_t pw_stats_block_rows(const ConvP& p);          // output rows one of them covers
int pw_run(const ConvP& p, bool is_f16, hipStream_t st);

// ---- weight-stationary kernel for the long-K pointwise layers (K = 512 / 1024 / 1280), conv_pwk.hip -------------------------------
bool pwk_applicable(const ConvP& p, int esz);
int64_t pwk_stats_blocks(const ConvP& p);
int64_t pwk_stats_block_rows(const ConvP& p);
int pwk_run(const ConvP& p, bool is_f16, hipStream_t st);

// ---- weight-stationary 3x3 kernel for the 64-input-channel layers (HRFP ends, stem / layer-1 3x3), conv_c64.hip ---------------
// accumulator-stationary weight gradient of the 3x3 / stride 1 / dilation <= 2 layers (conv_wg3.hip): slab slots per problem it may use
// (0: never applicable to such a problem), the rule, the launch (slab layout and reduction: conv_wgrad.hip)
int64_t wg3_splits_bound(int64_t N, int64_t Q, int64_t count);
bool wg3_applicable(int dtype_size, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t ldn, int64_t R, int64_t S, int64_t Ho,
                    int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t count);
int wg3_run(const void* const* xs, const void* const* dys, int64_t count, float* slab, bool is_f16, int64_t B, int64_t H, int64_t W, int64_t C,
            int64_t N, int64_t ldn, int64_t dil, unsigned xbytes, unsigned dybytes, int* splits, hipStream_t st);

// accumulator-stationary weight gradient of the pointwise layers with C % 256 == N % 256 == 0 (conv_wg1.hip)
int64_t wg1_splits_bound(int64_t N, int64_t Q, int64_t count);
bool wg1_applicable(int dtype_size, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t ldn, int64_t R, int64_t S, int64_t Ho,
                    int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t count);
int wg1_run(const void* const* xs, const void* const* dys, int64_t count, float* slab, bool is_f16, int64_t M, int64_t C, int64_t N, int64_t ldn,
            unsigned xbytes, unsigned dybytes, int* splits, hipStream_t st);

bool c64_applicable(const ConvP& p, int esz);
int64_t c64_stats_blocks(const ConvP& p);             // statistics rows such a launch writes: [image][sub-strip][slot]
int64_t c64_stats_block_rows(const ConvP& p);         // NEGATIVE: -(rows per image) -- the rows are per image, not per fixed row count
int c64_run(const ConvP& p, bool is_f16, hipStream_t st);

}  // namespace mrfp
