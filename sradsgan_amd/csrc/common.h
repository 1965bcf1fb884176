// Shared host/device helpers for libsradsgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/sradsgan_hip.h"

namespace srhip {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SRHIP_ERR_LAUNCH;
  }
  return SRHIP_OK;
}

#define SRHIP_REQUIRE(cond, ...)     \
  do {                               \
    if (!(cond)) {                   \
      srhip::set_error(__VA_ARGS__); \
      return SRHIP_ERR_ARG;          \
    }                                \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave-wide reductions (64 lanes)
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// value of the neighbouring lane (lane ^ 1) through the DPP network: quad_perm [1, 0, 3, 2]; both lanes of a pair must be active
__device__ __forceinline__ unsigned pair_swap(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// max-pool comparisons propagate NaN like ATen's (adaptive_max_pool2d / max(dim)): a NaN beats every number, the first
// NaN wins among NaNs -- a NaN activation must surface in the loss, not be masked by an attention gate (cbam.hip: the
// discriminator's pair; attn_tail.hip: the generator's 48 CLAM / SLAM pools)
__device__ __forceinline__ bool pool_takes(float v, float mx) { return v > mx || (v != v && mx == mx); }
__device__ __forceinline__ bool pool_merge_takes(float om, int oi, float mx, int idx) {
  const bool on = om != om, mn = mx != mx;
  if (on || mn) return on && (!mn || oi < idx);
  return om > mx || (om == mx && oi < idx);
}

constexpr int POOL_MAXSEG = 64;   // most CLAM pooling partial segments per image (attn_tail.hip consumers, conv epilogue producer)

}  // namespace srhip
