// Shared helpers for the libadm_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/adm_hip.h"

// The 16-bit element type of activations and weights.  Default build: bfloat16 (libadm_hip.so; BASELINE config 2 names bf16).
// -DADM_ACT_F16 builds the SAME kernels for IEEE half (libadm_hip_f16.so): the reference's own torso type
// (use_fp16=True, unet.py:618-624) -- 11 mantissa bits instead of 8, the matrix cores run both at one rate
// (v_mfma_f32_16x16x32_f16 / _bf16), accumulation / GroupNorm / softmax / sampler stay fp32 in either.
#ifdef ADM_ACT_F16
typedef _Float16 adm_elem_t;
#define ADM_ONE16 0x3C00
#else
typedef __bf16 adm_elem_t;
#define ADM_ONE16 0x3F80
#endif
typedef __attribute__((ext_vector_type(8))) adm_elem_t adm_h8;
typedef __attribute__((ext_vector_type(4))) adm_elem_t adm_h4;
typedef __attribute__((ext_vector_type(2))) adm_elem_t adm_h2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float adm_f32x16;
typedef __attribute__((ext_vector_type(4))) short adm_s16x4_t;

// thread-local error text, set by ADM_FAIL / adm_check_launch
void adm_set_error(const char* fmt, ...);
// CUs a persistent kernel launched on `stream` may occupy: the stream's registered CU budget (adm_stream_create_cumask /
// adm_stream_set_cus, adm_api.hip), else `dflt` (the device's CU count)
int adm_stream_cus(void* stream, int dflt);

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

// element <-> fp32 (the raw 16-bit pattern travels as uint16_t / packed pairs in uint32_t)
#ifdef ADM_ACT_F16
__device__ __forceinline__ float adm_h_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t adm_f32_to_h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }  // round-to-nearest-even
__device__ __forceinline__ float adm_lo_f32(uint32_t u) { return (float)__builtin_bit_cast(adm_h2, u)[0]; }
__device__ __forceinline__ float adm_hi_f32(uint32_t u) { return (float)__builtin_bit_cast(adm_h2, u)[1]; }  // v_cvt_f32_f16 ... WORD_1
#else
__device__ __forceinline__ float adm_h_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t adm_f32_to_h(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float adm_lo_f32(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float adm_hi_f32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
#endif
// two fp32 -> one packed pair (round-to-nearest-even): ONE instruction in either build (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t adm_pack2(float lo, float hi) {
  const adm_h2 v = {(adm_elem_t)lo, (adm_elem_t)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// the matrix-core products of the element type (A, B fragments as adm_h8 / 4 x 16 bits; fp32 accumulators)
__device__ __forceinline__ f32x4 adm_mfma_16x16x32(adm_h8 a, adm_h8 b, f32x4 c, int, int, int) {
#ifdef ADM_ACT_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ adm_f32x16 adm_mfma_32x32x16(adm_h8 a, adm_h8 b, adm_f32x16 c, int, int, int) {
#ifdef ADM_ACT_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 adm_mfma_16x16x16(adm_s16x4_t a, adm_s16x4_t b, f32x4 c, int, int, int) {
#ifdef ADM_ACT_F16
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(adm_h4, a), __builtin_bit_cast(adm_h4, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#endif
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
