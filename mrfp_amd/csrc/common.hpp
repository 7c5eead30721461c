// common.hpp -- shared device helpers for libmrfp_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/mrfp_hip.h"

namespace mrfp {

void set_error(const char* fmt, ...);

#define MRFP_CHECK(cond, ...)                      \
    do {                                           \
        if (!(cond)) {                             \
            ::mrfp::set_error(__VA_ARGS__);        \
            return -1;                             \
        }                                          \
    } while (0)

#define MRFP_LAUNCH_CHECK()                                                   \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            ::mrfp::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,  \
                              hipGetErrorString(e__));                        \
            return -2;                                                        \
        }                                                                     \
    } while (0)

typedef __hip_bfloat16 bf16;
typedef _Float16 f16;        // IEEE half (MRFP_F16): same 16-byte chunks and MFMA rate as bf16, 11-bit mantissa

// ---- element <-> float ----------------------------------------------------------------------
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return __bfloat162float(v); }
__device__ __forceinline__ float to_f(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ f16 from_f<f16>(float v) { return (f16)v; }

// ---- 16-byte vectors of VEC elements ---------------------------------------------------------
// VecT<T,VEC>: VEC elements moved with ONE memory instruction when VEC*sizeof(T) == 16
// (float x4, bf16 x8); VEC == 1 is the scalar tail path for odd channel counts (C = 19).
template <typename T, int VEC> struct alignas(sizeof(T) * VEC) VecT { T v[VEC]; };

template <typename T> struct FullVec { static constexpr int value = 16 / sizeof(T); };

template <typename T, int VEC>
__device__ __forceinline__ void load_f(const T* p, float (&out)[VEC]) {
    VecT<T, VEC> t = *reinterpret_cast<const VecT<T, VEC>*>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = to_f(t.v[i]);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_f(T* p, const float (&in)[VEC]) {
    VecT<T, VEC> t;
#pragma unroll
    for (int i = 0; i < VEC; ++i) t.v[i] = from_f<T>(in[i]);
    *reinterpret_cast<VecT<T, VEC>*>(p) = t;
}
// raw (unconverted) vector loads: keep several 16-byte loads in flight per lane at 4 VGPRs each, convert on use
template <typename T, int VEC>
__device__ __forceinline__ VecT<T, VEC> load_raw(const T* p) { return *reinterpret_cast<const VecT<T, VEC>*>(p); }
// the same with the non-temporal cache policy: for the LAST read of a tensor in a pass sequence (nothing re-reads these
// bytes soon, so they should not displace lines that the next kernel will hit in L2 / the Infinity Cache).  Used by the
// normalisation apply kernels (affine.hip; the statistics pass in front of them keeps the default policy so that the
// apply pass finds the tensor in the caches): bench step 62.4 -> 61.4 ms (tools/ab_lib.sh, same box, 3 alternations;
// `tools/build_variant.sh nont affine -DMRFP_NT=0` builds the default-policy library).
#ifndef MRFP_NT
#define MRFP_NT 1
#endif
template <typename T, int VEC>
__device__ __forceinline__ VecT<T, VEC> load_raw_nt(const T* p) {
    if constexpr (MRFP_NT != 0 && sizeof(T) * VEC == 16) {
        typedef unsigned __attribute__((ext_vector_type(4))) nt_u32x4;
        const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p));
        return __builtin_bit_cast(VecT<T, VEC>, v);
    } else {
        return *reinterpret_cast<const VecT<T, VEC>*>(p);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void cvt_f(const VecT<T, VEC>& t, float (&out)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = to_f(t.v[i]);
}

template <int VEC>
__device__ __forceinline__ void load_coef(const float* p, float (&out)[VEC]) {
    VecT<float, VEC> t = *reinterpret_cast<const VecT<float, VEC>*>(p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = t.v[i];
}

// ---- reductions -------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Geometry of a "row kernel": one workgroup walks destination lines (b, oh) of a [B,Ho,Wo,C]
// tensor; 256 threads are laid out as (rowthreads x colthreads) over (ow, channel-vector).
struct RowGeom {
    int B, Ho, Wo, C, Hs, Ws;
    const int32_t* tabH;
    const int32_t* tabW;
};

constexpr int kThreads = 256;
constexpr int kCUs = 256;        // compute units of one MI355X (MI355X_MICROARCH.md, chip-level parameters): what "one round of resident
                                 // workgroups" means to the persistent kernels (conv_common.hpp: kGrid1PerCU / kGrid2PerCU; whiten.hip)

struct Lanes {
    int lpr;         // channel vectors per row = C / VEC
    int colthreads;  // threads across the channel dimension
    int rowthreads;  // threads across pixels of a line
};
__host__ __device__ inline Lanes make_lanes(int C, int VEC) {
    Lanes l;
    l.lpr = C / VEC;
    l.colthreads = l.lpr < kThreads ? l.lpr : kThreads;
    l.rowthreads = kThreads / l.colthreads;
    return l;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// choose the vector width for a channel count: full 16-byte vectors when C allows it
template <typename T> inline int pick_vec(int64_t C) {
    const int full = FullVec<T>::value;
    return (C % full == 0) ? full : 1;
}

inline int lines_per_image(int64_t B, int64_t Ho) {
    // ~1024 workgroups over the chip (4 per CU, 16 waves/CU, 4 independent 16-byte loads per lane in the
    // statistics kernels): enough bytes in flight to cover HBM latency, few enough partial sums that the
    // finalize kernels stay in the microseconds.
    static int total = 0;
    if (total == 0) {
        const char* e = getenv("MRFP_ROW_BLOCKS");
        total = e ? atoi(e) : 2048;
        if (total < 64) total = 2048;
    }
    int64_t cap = total / (B > 0 ? B : 1);
    if (cap < 1) cap = 1;
    if (Ho <= cap) return (int)Ho;
    // every workgroup the same number of lines (the last few one less): 192 lines over 128 workgroups gave half of them two lines and
    // half one -- the launch ran as long as the two-line workgroups (round 4: 96 x 2 lines, normalisation family -0.3 ms per step)
    const int64_t per = (Ho + cap - 1) / cap;
    return (int)((Ho + per - 1) / per);
}

}  // namespace mrfp
