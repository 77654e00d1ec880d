// Shared pieces of the attention forward / backward kernels (gfx950).
#pragma once
#include "adm_common.h"

typedef __attribute__((ext_vector_type(4))) short adm_s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int adm_u32x4;
typedef __attribute__((ext_vector_type(2))) float adm_f32x2;

// XCD-aware block order for grids of (row blocks, image x head): the dispatcher deals consecutive linear block ids
// round-robin to the 8 XCDs (each with its own L2), so with 8 row blocks per (image, head) every one of them would land
// on a different XCD and each L2 would fetch that head's K / V separately.  Block L instead takes logical block
// (L % 8) * ceil-share + L / 8: one XCD owns a contiguous run, i.e. all row blocks of a head share an L2.
__device__ __forceinline__ void adm_xcd_block(int& bx, int& by) {
  const unsigned nx = gridDim.x, total = gridDim.x * gridDim.y;
  const unsigned lin = blockIdx.x + nx * blockIdx.y;
  const unsigned share = total >> 3, rem = total & 7, xcd = lin & 7;
  const unsigned logical = xcd * share + min(xcd, rem) + (lin >> 3);
  bx = (int)(logical % nx);
  by = (int)(logical / nx);
}

// MFMA A-operand fragment (16 rows x 32 k, bf16) of the TRANSPOSE of a row-major LDS tile, read with the
// hardware transposing load ds_read_b64_tr_b16: lane (lc = l & 15, lq = l >> 4) receives
//   element e (0..7) = tile[row0 + 16*(e>>2) + 4*lq + (e&3)][col0 + lc]
// i.e. "column col0+lc" for a k-order that matches a B operand taken straight from 16x16x32
// accumulators (4 consecutive rows per lane quarter, two 16-row tiles per 32-deep k-step).
// Each 16-lane group reads a 4-row x 16-column block; lane 4q+p supplies the address of row q, columns 4p..4p+3.
__device__ __forceinline__ adm_h8 adm_tr_frag(const uint16_t* tile, int row_stride, int row0, int col0, int lc, int lq) {
  const uint16_t* p0 = tile + (row0 + 4 * lq + (lc >> 2)) * row_stride + col0 + 4 * (lc & 3);
  const adm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) adm_s16x4*)p0);
  const adm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) adm_s16x4*)(p0 + 16 * row_stride));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(adm_h8, v);
}

// The same fragment of a [rows][64] bf16 tile with UNPADDED 128-byte rows whose 16-byte segments are XOR-swizzled
// (segment s of row r lives in slot s ^ (r & 7): the layout an LDS-DMA fill leaves, adm_attention.hip); row0 must be a
// multiple of 8.  The 8-byte pieces of a lane group stay inside one segment, so the transposing read is unchanged.
__device__ __forceinline__ adm_h8 adm_tr_frag_swz(const uint16_t* tile, int row0, int col0, int lc, int lq) {
  const int r = 4 * lq + (lc >> 2);                         // row within the 16-row block (row0 = 0 mod 8: r & 7 decides)
  const int seg = ((col0 >> 3) + ((lc & 3) >> 1)) ^ (r & 7);
  const uint16_t* p0 = tile + (row0 + r) * 64 + seg * 8 + 4 * (lc & 1);
  const adm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) adm_s16x4*)p0);
  const adm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) adm_s16x4*)(p0 + 16 * 64));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(adm_h8, v);
}

// [ROWS x D] bf16 tile of a token-major tensor, register-staged: issue the loads early, write late.
template <int ROWS, int D, int NT>
struct AdmTileRegs {
  static constexpr int UNITS = ROWS * D / 8;
  static constexpr int PER = (UNITS + NT - 1) / NT;
  uint4 v[PER];
  __device__ __forceinline__ void load(const uint16_t* base, long long row_stride, int col0, int r0, int rmax, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * NT;
      const int r = u / (D / 8), sg = u % (D / 8);
      uint4 x = make_uint4(0, 0, 0, 0);
      if (u < UNITS && r0 + r < rmax) x = *reinterpret_cast<const uint4*>(base + (long long)(r0 + r) * row_stride + col0 + sg * 8);
      v[i] = x;
    }
  }
  // the same through a buffer descriptor over the rows that exist (num_records = rmax * row_stride * 2 bytes from
  // `base`): rows beyond it read as hardware zeros, so the loop carries no bounds branches
  __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rs, int row_stride, int col0, int r0, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * NT;
      const int r = u / (D / 8), sg = u % (D / 8);
      const unsigned off = u < UNITS ? (unsigned)((r0 + r) * row_stride + col0 + sg * 8) * 2u : 0x80000000u;
      const adm_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
      v[i] = make_uint4(x[0], x[1], x[2], x[3]);
    }
  }
  __device__ __forceinline__ void store(uint16_t* tile, int krow, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = tid + i * NT;
      if (u < UNITS) *reinterpret_cast<uint4*>(&tile[(u / (D / 8)) * krow + (u % (D / 8)) * 8]) = v[i];
    }
  }
};

// Reductions over the 4 lane quarters that share one MFMA column (lanes l, l^16, l^32, l^48) with the VALU
// row/half swaps of gfx950 instead of ds_bpermute shuffles: v_permlane16_swap exchanges odd 16-lane rows of
// the first operand with even rows of the second, v_permlane32_swap the upper half with the lower half.
typedef __attribute__((ext_vector_type(2))) unsigned int adm_u32x2;
__device__ __forceinline__ float adm_quarter_max(float x) {
  adm_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float adm_quarter_sum(float x) {
  adm_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
