// K1: timestep embedding and the fp32 embedding MLPs.
//
// timestep_embedding (reference guided_diffusion/nn.py:103-121) and the Linear layers fed by
// it: time_embed (unet.py:470-475), label_emb gather-add (unet.py:652-654) and every ResBlock's
// emb_layers = SiLU -> Linear (unet.py:199-205), batched across blocks by concatenating the
// weights along the output dimension.  0.03 % of the UNet's FLOPs; kept in fp32 like the
// reference (these layers are not converted by convert_to_fp16).
#include "adm_common.h"

namespace {

__global__ void __launch_bounds__(256)
timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int n, int dim, float neg_log_period) {
  const int half = dim / 2;
  const int total = n * dim;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / dim, col = i % dim;
    float v = 0.0f;
    if (col < 2 * half) {
      const int k = col < half ? col : col - half;
      const float freq = expf(neg_log_period * (float)k / (float)half);
      const float a = t[row] * freq;
      v = col < half ? cosf(a) : sinf(a);
    }
    out[i] = v;
  }
}

// out[n, o] = sum_k act(in[n, k]) * w[o, k] + bias[o] (+ table[idx[n], o])
// 64x64 output tile per 256-thread block, 4x4 outputs per thread, K staged 16 at a time.
constexpr int LT = 64, LK = 16;

template <bool SILU>
__global__ void __launch_bounds__(256)
linear_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
              const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out,
              int n, int k, int o) {
  __shared__ float As[LK][LT + 4];
  __shared__ float Ws[LK][LT + 4];
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int n0 = blockIdx.y * LT, o0 = blockIdx.x * LT;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < k; k0 += LK) {
    // each thread stages 4 elements of each operand: row r = tid/4 (+0), k-offset (tid%4)*4
    const int r = threadIdx.x / 4, kk = (threadIdx.x % 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kc = k0 + kk + j;
      float a = 0.0f, b = 0.0f;
      if (kc < k) {
        if (n0 + r < n) {
          a = in[(long long)(n0 + r) * k + kc];
          if (SILU) a = adm_silu(a);
        }
        if (o0 + r < o) b = w[(long long)(o0 + r) * k + kc];
      }
      As[kk + j][r] = a;
      Ws[kk + j][r] = b;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LK; ++q) {
      float av[4], bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        av[j] = As[q][ty * 4 + j];
        bv[j] = Ws[q][tx * 4 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = n0 + ty * 4 + i;
    if (row >= n) continue;
    const float* trow = table ? table + (long long)idx[row] * o : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = o0 + tx * 4 + j;
      if (col >= o) continue;
      float v = acc[i][j] + (bias ? bias[col] : 0.0f);
      if (trow) v += trow[col];
      out[(long long)row * o + col] = v;
    }
  }
}


// Many rows, K a multiple of 16 (every ResBlock's emb_layers at batch 256: n = 256, k = 768, o = 33 792): the same fp32
// product on the matrix pipe, v_mfma_f32_16x16x4_f32 -- exact fp32 multiplies with fp32 accumulation, so the numerics stay
// those of the tile kernel above up to the summation order.  A block is 4 waves x 64 rows by 32 output columns; both
// operands are K-contiguous, so a lane fetches 16 bytes of its row (4 consecutive k) straight from global memory and feeds
// element j to the j-th of 4 MFMAs (the k order inside a 16-deep step is permuted the same way for both operands); no LDS,
// the four waves share the weight fragments through the L1, the activations (n x k) stay in L2.  The tile kernel reached
// 50 TFLOP/s on that shape (264 us average, 1.2 % of the guided batch, profiles/r02/bench_guided_b256_b_kernel_stats.csv).
typedef __attribute__((ext_vector_type(4))) float lin_f32x4;

// COLSPLIT (n <= 64 rows: one 64-row wave tile holds them all): the four waves of a block take four 32-column groups instead
// of four 64-row groups, so that no wave idles.  An output element sees the same MFMA sequence either way: a row's bits do not
// depend on the batch it rides in (the kernel is chosen by (k, alignment) only -- adm_linear_f32).
template <bool SILU, int KU, bool COLSPLIT>   // KU float4 per operand row and step: 16 * KU k-values per step (k % (16 * KU) == 0)
__global__ void __launch_bounds__(256)
linear_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                   const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out,
                   int n, int k, int o) {
  constexpr int TM = 4, TN = 2, KSTEP = 16 * KU;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  const int row0 = COLSPLIT ? 0 : (blockIdx.y * 4 + wave) * (TM * 16);
  const int col0 = (COLSPLIT ? blockIdx.x * 4 + wave : blockIdx.x) * (TN * 16);
  if (row0 >= n || col0 >= o) return;   // wave-uniform; no barriers in this kernel
  const float* ap[TM];
  const float* bp[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) ap[i] = in + (long long)min(row0 + i * 16 + lc, n - 1) * k + lq * 4;   // clamped rows are not stored
#pragma unroll
  for (int j = 0; j < TN; ++j) bp[j] = w + (long long)min(col0 + j * 16 + lc, o - 1) * k + lq * 4;
  lin_f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = lin_f32x4{0.f, 0.f, 0.f, 0.f};
  float4 a[KU][TM], b[KU][TN], an[KU][TM], bn[KU][TN];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a[u][i] = *reinterpret_cast<const float4*>(ap[i] + u * 16);
#pragma unroll
    for (int j = 0; j < TN; ++j) b[u][j] = *reinterpret_cast<const float4*>(bp[j] + u * 16);
  }
  for (int k0 = 0; k0 < k; k0 += KSTEP) {
    const int kn = k0 + KSTEP < k ? k0 + KSTEP : k0;   // the last step re-reads its own operands (unused)
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int i = 0; i < TM; ++i) an[u][i] = *reinterpret_cast<const float4*>(ap[i] + kn + u * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) bn[u][j] = *reinterpret_cast<const float4*>(bp[j] + kn + u * 16);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (SILU) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          a[u][i].x = adm_silu(a[u][i].x); a[u][i].y = adm_silu(a[u][i].y);
          a[u][i].z = adm_silu(a[u][i].z); a[u][i].w = adm_silu(a[u][i].w);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float av = e == 0 ? a[u][i].x : e == 1 ? a[u][i].y : e == 2 ? a[u][i].z : a[u][i].w;
            const float bv = e == 0 ? b[u][j].x : e == 1 ? b[u][j].y : e == 2 ? b[u][j].z : b[u][j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
          }
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[u][i] = an[u][i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[u][j] = bn[u][j];
    }
  }
  // accumulator layout: lane holds column lc, rows 4 * lq + e
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 16 + lc;
    if (col >= o) continue;
    const float bj = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + i * 16 + lq * 4 + e;
        if (row >= n) continue;
        float v = acc[i][j][e] + bj;
        if (table) v += table[(long long)idx[row] * o + col];
        out[(long long)row * o + col] = v;
      }
  }
}

// K not a multiple of 16 (k % 4 == 0; designed for few rows): a GEMV-shaped pass.  The 64 x 64 tile kernel above
// leaves most of its rows empty and crawls through K behind two barriers per 16-deep step; here a block keeps 16 input rows
// (activation applied) in LDS, each wave streams whole weight rows with 16-byte lanes (every weight is read once per 16
// rows) and reduces 16 dot products per output column across its lanes.
constexpr int LS_ROWS = 16, LS_CPW = 8;  // rows per block, output columns per wave

template <bool SILU>
__global__ void __launch_bounds__(256)
linear_small_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                    const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out,
                    int n, int k, int o) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [LS_ROWS][k]
  const int r0 = blockIdx.y * LS_ROWS;
  const int nr = min(LS_ROWS, n - r0);
  for (int i = threadIdx.x; i < LS_ROWS * k; i += 256) {
    const int r = i / k, kk = i - r * k;
    float v = 0.0f;
    if (r < nr) {
      v = in[(long long)(r0 + r) * k + kk];
      if (SILU) v = adm_silu(v);
    }
    xs[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col0 = (blockIdx.x * 4 + wave) * LS_CPW;
  for (int c = 0; c < LS_CPW; ++c) {
    const int col = col0 + c;
    if (col >= o) break;  // wave-uniform
    float acc[LS_ROWS];
#pragma unroll
    for (int r = 0; r < LS_ROWS; ++r) acc[r] = 0.0f;
    const float* wrow = w + (long long)col * k;
    for (int kk = lane * 4; kk < k; kk += 256) {
      const float4 wv = *reinterpret_cast<const float4*>(wrow + kk);
#pragma unroll
      for (int r = 0; r < LS_ROWS; ++r) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + r * k + kk);
        // explicit fma chain: left to contraction, the unrolled rows were not all fused the same way and a row's
        // result depended on its position in the block (batch-slice invariance is tested bitwise)
        acc[r] = __builtin_fmaf(wv.x, xv.x, acc[r]);
        acc[r] = __builtin_fmaf(wv.y, xv.y, acc[r]);
        acc[r] = __builtin_fmaf(wv.z, xv.z, acc[r]);
        acc[r] = __builtin_fmaf(wv.w, xv.w, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < LS_ROWS; ++r) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc[r] += __shfl_xor(acc[r], off);
    }
    float mine = 0.0f;  // lane r keeps row r's sum
#pragma unroll
    for (int r = 0; r < LS_ROWS; ++r)
      if (lane == r) mine = acc[r];
    if (lane < nr) {
      const int row = r0 + lane;
      float v = mine + (bias ? bias[col] : 0.0f);
      if (table) v += table[(long long)idx[row] * o + col];
      out[(long long)row * o + col] = v;
    }
  }
}

}  // namespace

extern "C" int adm_timestep_embedding(const float* t, float* out, int n, int dim, float max_period, void* stream) {
  ADM_REQUIRE(t && out, ADM_E_ARG, "adm_timestep_embedding: null pointer");
  ADM_REQUIRE(n > 0 && dim > 1 && max_period > 0, ADM_E_ARG, "adm_timestep_embedding: bad args n=%d dim=%d", n, dim);
  const int total = n * dim;
  int blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, out, n, dim,
                     -logf(max_period));
  return adm_check_launch("adm_timestep_embedding");
}

extern "C" int adm_linear_f32(const float* in, const float* w, const float* bias, const float* table,
                              const int64_t* idx, float* out, int n, int k, int o, int silu_in, void* stream) {
  ADM_REQUIRE(in && w && out, ADM_E_ARG, "adm_linear_f32: null pointer");
  ADM_REQUIRE(n > 0 && k > 0 && o > 0, ADM_E_ARG, "adm_linear_f32: bad shape n=%d k=%d o=%d", n, k, o);
  ADM_REQUIRE((table == nullptr) == (idx == nullptr), ADM_E_ARG, "adm_linear_f32: table and idx go together");
  hipStream_t s = (hipStream_t)stream;
  const size_t small_lds = (size_t)LS_ROWS * k * sizeof(float);
  const bool mfma_ok = k % 16 == 0 && adm_aligned16(w) && adm_aligned16(in);
  // The kernel is chosen by (k, alignment) alone, never by the number of rows: the kernels sum k in different orders, and a
  // row's embedding bits must not depend on the batch it rides in (an image's result is independent of how a candidate's
  // images are sharded and batched: tests/test_hip_bigbatch.py holds the whole UNet to that, bitwise, at batch 256 vs 2).
  if (mfma_ok) {
    // 16 k-values per step: 32- and 64-deep steps (KU = 2, 4) measured 15 % slower (205 vs 179 us on 256 x 768 -> 33 792)
    if (n <= 64) {   // one wave tile of rows: the block's four waves split the columns
      dim3 g((o + 127) / 128, 1);
      if (silu_in) hipLaunchKernelGGL((linear_mfma_kernel<true, 1, true>), g, dim3(256), 0, s, in, w, bias, table, idx, out, n, k, o);
      else hipLaunchKernelGGL((linear_mfma_kernel<false, 1, true>), g, dim3(256), 0, s, in, w, bias, table, idx, out, n, k, o);
    } else {
      dim3 g((o + 31) / 32, (n + 255) / 256);
      if (silu_in) hipLaunchKernelGGL((linear_mfma_kernel<true, 1, false>), g, dim3(256), 0, s, in, w, bias, table, idx, out, n, k, o);
      else hipLaunchKernelGGL((linear_mfma_kernel<false, 1, false>), g, dim3(256), 0, s, in, w, bias, table, idx, out, n, k, o);
    }
    return adm_check_launch("adm_linear_f32");
  }
  // k not a multiple of 16 (the attention pool's 1000-wide backward projection): the GEMV-shaped kernel at ANY row count -- a block
  // owns 16 rows, so more rows are more blocks re-streaming the weights through L2; choosing the tile kernel from 65 rows on made
  // the guidance gradient's bits depend on the batch size (found by tests/test_hip_bigbatch.py at batch 256 vs 2)
  if (k % 4 == 0 && small_lds <= 128 * 1024 && adm_aligned16(w) && adm_aligned16(in)) {
    static bool attr_set[64][2] = {};  // per device: opt in to the dynamic LDS size once per instantiation
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& done = attr_set[dev & 63][silu_in ? 1 : 0];
    if (!done) {
      const void* fn = silu_in ? reinterpret_cast<const void*>(&linear_small_kernel<true>)
                               : reinterpret_cast<const void*>(&linear_small_kernel<false>);
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e != hipSuccess) ADM_FAIL((int)e, "adm_linear_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
      done = true;
    }
    dim3 g((o + 4 * LS_CPW - 1) / (4 * LS_CPW), (n + LS_ROWS - 1) / LS_ROWS);
    if (silu_in) hipLaunchKernelGGL((linear_small_kernel<true>), g, dim3(256), small_lds, s, in, w, bias, table, idx, out, n, k, o);
    else hipLaunchKernelGGL((linear_small_kernel<false>), g, dim3(256), small_lds, s, in, w, bias, table, idx, out, n, k, o);
    return adm_check_launch("adm_linear_f32");
  }
  dim3 grid((o + LT - 1) / LT, (n + LT - 1) / LT);
  if (silu_in) hipLaunchKernelGGL((linear_kernel<true>), grid, dim3(256), 0, s, in, w, bias, table, idx, out, n, k, o);
  else hipLaunchKernelGGL((linear_kernel<false>), grid, dim3(256), 0, s, in, w, bias, table, idx, out, n, k, o);
  return adm_check_launch("adm_linear_f32");
}
