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

// Gram update on the f64 matrix cores: S2 += A^T A is a GEMM with M = i, N = j, K = sample, so per 4 samples one
// v_mfma_f64_16x16x4_f64 takes, in lane l, A-operand a[n0 + (l >> 4)][i0 + (l & 15)] and B-operand
// a[n0 + (l >> 4)][j0 + (l & 15)] (fp32 activations widened to f64: every product is exact, only the summation order
// differs from numpy's float64 covariance) and accumulates a 16 x 16 tile; a lane's 4 results are rows
// (l >> 4) + 4 r, column l & 15 (the f64 C/D map, not the f32 one).  Only the tiles on and above the diagonal are
// computed (S2 is symmetric): a block owns the 64 x 64 tile (bi <= bj) AND its mirror image, so no element has two writers
// and the += needs no atomics.  Diagonal blocks also add the column sums s1.
// Algorithmic work: 2 * n * d * (d + 64) / 2 FLOP per call (n = 5000, d = 2048: 21.6 GFLOP), plus one read-modify-write
// of S2 (67 MB at d = 2048), which bounds the per-batch calls of the search (n <= 256: HBM-bound).
constexpr int GT = 64;    // output tile edge per block (4 waves, 32 x 32 per wave = 2 x 2 MFMA tiles)
constexpr int GK = 16;    // samples staged per step
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
gram_mfma_kernel(const float* __restrict__ a, double* __restrict__ s2, double* __restrict__ s1, int n, int d) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) return;                    // the mirror image is written by block (bj, bi)'s twin
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int kq = lane >> 4, c = lane & 15;
  const int i0 = bi * GT, j0 = bj * GT;
  f64x4 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = f64x4{0.0, 0.0, 0.0, 0.0};
  double colsum = 0.0;                    // diagonal blocks: thread t < 64 sums column j0 + t
  const int r = tid / 16, c4 = (tid % 16) * 4;   // staging: row k0 + r, columns c4 .. c4 + 3 of both tiles
  for (int k0 = 0; k0 < n; k0 += GK) {
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
    if (k0 + r < n) {
      const float* row = a + (long long)(k0 + r) * d;
      if (i0 + c4 + 3 < d) va = *reinterpret_cast<const float4*>(row + i0 + c4);
      else { float t[4] = {0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 4; ++e) if (i0 + c4 + e < d) t[e] = row[i0 + c4 + e]; va = make_float4(t[0], t[1], t[2], t[3]); }
      if (j0 + c4 + 3 < d) vb = *reinterpret_cast<const float4*>(row + j0 + c4);
      else { float t[4] = {0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 4; ++e) if (j0 + c4 + e < d) t[e] = row[j0 + c4 + e]; vb = make_float4(t[0], t[1], t[2], t[3]); }
    }
    *reinterpret_cast<float4*>(&As[r][c4]) = va;
    *reinterpret_cast<float4*>(&Bs[r][c4]) = vb;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      double af[2], bf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = (double)As[kk + kq][(2 * wi + t) * 16 + c];
        bf[t] = (double)Bs[kk + kq][(2 * wj + t) * 16 + c];
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
    }
    if (bi == bj && tid < GT) {
#pragma unroll
      for (int q = 0; q < GK; ++q) colsum += (double)Bs[q][tid];
    }
    __syncthreads();
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + (2 * wi + ti) * 16 + kq + 4 * q, j = j0 + (2 * wj + tj) * 16 + c;
        if (i < d && j < d) {
          s2[(long long)i * d + j] += acc[ti][tj][q];
          if (bi != bj) s2[(long long)j * d + i] += acc[ti][tj][q];
        }
      }
  if (bi == bj && tid < GT && j0 + tid < d) s1[j0 + tid] += colsum;
}

}  // namespace

extern "C" int adm_fid_accumulate(const float* acts, double* s1, double* s2, int n, int d, void* stream) {
  ADM_REQUIRE(acts && s1 && s2, ADM_E_ARG, "adm_fid_accumulate: null pointer");
  ADM_REQUIRE(n > 0 && d > 0, ADM_E_ARG, "adm_fid_accumulate: bad shape n=%d d=%d", n, d);
  ADM_REQUIRE(d % 4 == 0 && adm_aligned16(acts), ADM_E_ALIGN, "adm_fid_accumulate: d %% 4 == 0 and 16-byte aligned activations required");
  const int t = (d + GT - 1) / GT;
  hipLaunchKernelGGL(gram_mfma_kernel, dim3(t, t), dim3(256), 0, (hipStream_t)stream, acts, s2, s1, n, d);
  return adm_check_launch("adm_fid_accumulate");
}
