// Shared helpers for the libadm_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/adm_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// thread-local error text, set by ADM_FAIL / adm_check_launch
void adm_set_error(const char* fmt, ...);

#define ADM_FAIL(code, ...)        \
  do {                             \
    adm_set_error(__VA_ARGS__);    \
    return (code);                 \
  } while (0)

#define ADM_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) ADM_FAIL(code, __VA_ARGS__); \
  } while (0)

static inline int adm_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    adm_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static inline bool adm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

__device__ __forceinline__ float adm_bf16_to_f32(uint16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ uint16_t adm_f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float adm_silu(float v) {
  // v * sigmoid(v) = v * rcp(1 + 2^(-v*log2 e)): v_exp_f32 + v_rcp_f32 (1 ulp each), 5 VALU ops -- a full
  // IEEE division here would triple the VALU cost of the conv prologue, which shares issue slots with MFMA
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
// two at a time: the non-transcendental half of the work as packed fp32 (v_pk_mul_f32 / v_pk_add_f32)
typedef float adm_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ adm_f32x2_t adm_silu2(adm_f32x2_t v) {
  const adm_f32x2_t t = v * adm_f32x2_t{-1.4426950408889634f, -1.4426950408889634f};
  const adm_f32x2_t d = adm_f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + adm_f32x2_t{1.0f, 1.0f};
  return v * adm_f32x2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
