// K11: FID activation statistics on the GPU.
//
// Replaces np.mean / np.cov over the [N, 2048] Inception activations
// (reference evaluations/evaluator_v1.py:218-221) by streaming accumulation of
//   s1[j] += sum_n a[n][j]          s2[i][j] += sum_n a[n][i] * a[n][j]
// in float64 (an fp32 x fp32 product is exact in fp64, so only the summation order differs from
// numpy's float64 covariance).  Batches are accumulated as they are produced; ranks pool (n, s1, s2)
// with one RCCL all-gather per candidate, and the host finishes
//   sigma = (s2 - n mu mu^T) / (n - 1)  and the 2048^2 sqrtm in float64.
#include "adm_common.h"

namespace {

constexpr int FT = 64, FK = 16;

// s2 tile [64 x 64] per block, 4x4 per thread; a is [n][d] fp32 row-major
__global__ void __launch_bounds__(256)
gram_kernel(const float* __restrict__ a, double* __restrict__ s2, double* __restrict__ s1, int n, int d) {
  __shared__ float As[FK][FT + 4];
  __shared__ float Bs[FK][FT + 4];
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int i0 = blockIdx.y * FT, j0 = blockIdx.x * FT;
  double acc[4][4] = {};
  double colsum[4] = {};
  for (int k0 = 0; k0 < n; k0 += FK) {
    const int r = threadIdx.x / 16, c4 = (threadIdx.x % 16) * 4;  // row k0+r, columns c4..c4+3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float va = 0.f, vb = 0.f;
      if (k0 + r < n) {
        if (i0 + c4 + j < d) va = a[(long long)(k0 + r) * d + i0 + c4 + j];
        if (j0 + c4 + j < d) vb = a[(long long)(k0 + r) * d + j0 + c4 + j];
      }
      As[r][c4 + j] = va;
      Bs[r][c4 + j] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FK; ++q) {
      float av[4], bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { av[j] = As[q][ty * 4 + j]; bv[j] = Bs[q][tx * 4 + j]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += (double)av[i] * (double)bv[j];
      if (blockIdx.y == 0 && ty == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) colsum[j] += (double)bv[j];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = i0 + ty * 4 + i, c = j0 + tx * 4 + j;
      if (r < d && c < d) s2[(long long)r * d + c] += acc[i][j];
    }
  if (blockIdx.y == 0 && ty == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j0 + tx * 4 + j < d) s1[j0 + tx * 4 + j] += colsum[j];
  }
}

}  // namespace

extern "C" int adm_fid_accumulate(const float* acts, double* s1, double* s2, int n, int d, void* stream) {
  ADM_REQUIRE(acts && s1 && s2, ADM_E_ARG, "adm_fid_accumulate: null pointer");
  ADM_REQUIRE(n > 0 && d > 0, ADM_E_ARG, "adm_fid_accumulate: bad shape n=%d d=%d", n, d);
  dim3 grid((d + FT - 1) / FT, (d + FT - 1) / FT);
  hipLaunchKernelGGL(gram_kernel, grid, dim3(256), 0, (hipStream_t)stream, acts, s2, s1, n, d);
  return adm_check_launch("adm_fid_accumulate");
}
