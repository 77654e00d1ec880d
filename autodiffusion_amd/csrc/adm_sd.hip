// Token-wise kernels of the Stable-Diffusion SpatialTransformer (gfx950): LayerNorm and GEGLU over bf16
// [rows][C] token tensors (rows = images x pixels; in NHWC a token IS a pixel, so the Linear layers around them
// are adm_conv 1x1 launches).  Both are single-pass HBM streams: 16-byte lanes, fp32 math.
// Reference: "Stable Diffusion"/ldm/modules/attention.py:196-215 (BasicTransformerBlock), 36-44 (GEGLU).
#include "adm_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
  const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = adm_lo_f32(u[j]);
    f[2 * j + 1] = adm_hi_f32(u[j]);
  }
}

__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 pk;
  pk.x = adm_pack2(f[0], f[1]);
  pk.y = adm_pack2(f[2], f[3]);
  pk.z = adm_pack2(f[4], f[5]);
  pk.w = adm_pack2(f[6], f[7]);
  return pk;
}

// One wave per token row; the row stays in registers between the mean, the variance (two-pass, as
// torch.nn.LayerNorm computes it) and the normalised store.  SEGS = 16-byte segments per lane (C <= 512 * SEGS).
template <int SEGS>
__global__ void __launch_bounds__(256)
layernorm_kernel(const uint16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                 uint16_t* __restrict__ out, long long rows, int c, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nseg = c / 8;
  const float inv_c = 1.0f / (float)c;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    float v[SEGS][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SEGS; ++i) {
      const int sg = lane + i * 64;
      if (sg < nseg) {
        unpack8(*reinterpret_cast<const uint4*>(x + row * c + sg * 8), v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
      }
    }
    const float mean = wave_sum(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < SEGS; ++i)
      if (lane + i * 64 < nseg) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[i][j] - mean;
          ss += d * d;
        }
      }
    const float rstd = rsqrtf(wave_sum(ss) * inv_c + eps);
#pragma unroll
    for (int i = 0; i < SEGS; ++i) {
      const int sg = lane + i * 64;
      if (sg < nseg) {
        float g8[8], b8[8], y[8];
        *reinterpret_cast<float4*>(g8) = *reinterpret_cast<const float4*>(gamma + sg * 8);
        *reinterpret_cast<float4*>(g8 + 4) = *reinterpret_cast<const float4*>(gamma + sg * 8 + 4);
        *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(beta + sg * 8);
        *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(beta + sg * 8 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (v[i][j] - mean) * rstd * g8[j] + b8[j];
        *reinterpret_cast<uint4*>(out + row * c + sg * 8) = pack8(y);
      }
    }
  }
}

// out[r][i] = u[r][i] * gelu(u[r][inner + i]), exact (erf) GELU as F.gelu's default
__global__ void __launch_bounds__(256)
geglu_kernel(const uint16_t* __restrict__ u, uint16_t* __restrict__ out, long long rows, int inner) {
  const int sg = inner / 8;
  const long long items = rows * sg;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
    const long long r = it / sg;
    const int i = (int)(it % sg);
    float a[8], g[8], y[8];
    unpack8(*reinterpret_cast<const uint4*>(u + r * 2 * inner + i * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(u + r * 2 * inner + inner + i * 8), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = a[j] * (0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752f)));
    *reinterpret_cast<uint4*>(out + r * inner + i * 8) = pack8(y);
  }
}

}  // namespace

extern "C" int adm_layernorm(const adm_bf16* x, const float* gamma, const float* beta, adm_bf16* out, int64_t rows,
                             int c, float eps, void* stream) {
  ADM_REQUIRE(x && gamma && beta && out, ADM_E_ARG, "adm_layernorm: null pointer");
  ADM_REQUIRE(rows > 0 && c > 0 && c % 8 == 0 && c <= 2048, ADM_E_SHAPE, "adm_layernorm: rows=%lld c=%d unsupported (c %% 8 == 0, c <= 2048)",
              (long long)rows, c);
  ADM_REQUIRE(adm_aligned16(x) && adm_aligned16(out) && adm_aligned16(gamma) && adm_aligned16(beta), ADM_E_ALIGN,
              "adm_layernorm: unaligned pointer");
  long long blocks = (rows + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  const int segs = (c / 8 + 63) / 64;
  if (segs <= 1) hipLaunchKernelGGL((layernorm_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, out, (long long)rows, c, eps);
  else if (segs == 2) hipLaunchKernelGGL((layernorm_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, out, (long long)rows, c, eps);
  else hipLaunchKernelGGL((layernorm_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, out, (long long)rows, c, eps);
  return adm_check_launch("adm_layernorm");
}

extern "C" int adm_geglu(const adm_bf16* u, adm_bf16* out, int64_t rows, int inner, void* stream) {
  ADM_REQUIRE(u && out, ADM_E_ARG, "adm_geglu: null pointer");
  ADM_REQUIRE(rows > 0 && inner > 0 && inner % 8 == 0, ADM_E_SHAPE, "adm_geglu: rows=%lld inner=%d unsupported", (long long)rows, inner);
  ADM_REQUIRE(adm_aligned16(u) && adm_aligned16(out), ADM_E_ALIGN, "adm_geglu: unaligned pointer");
  const long long items = (long long)rows * (inner / 8);
  long long blocks = (items + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u, out, (long long)rows, inner);
  return adm_check_launch("adm_geglu");
}

// ------------------------------------------------------------------------------------------------
// One latent sampler update ("Stable Diffusion"/ldm/models/diffusion/ddim.py:165-203 p_sample_ddim,
// plms.py:195-258 p_sample_plms): classifier-free guidance combine, multistep eps blend, pred_x0 and x_prev in one
// pass over the latents (fp32, 5-7 streams instead of ~12 separate elementwise launches).
namespace {
struct SdStep {
  const float* x; const float* eu; const float* ec; const float* h1; const float* h2; const float* h3; const float* noise;
  float* x_prev; float* pred_x0; float* e_out;
  long long numel;
  float cfg, w0, w1, w2, w3, sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma;
};

__global__ void __launch_bounds__(256)
sd_step_kernel(const SdStep p) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.numel; i += (long long)gridDim.x * blockDim.x) {
    float e = p.ec[i];
    if (p.eu) {
      const float u = p.eu[i];
      e = u + p.cfg * (e - u);
    }
    if (p.e_out) p.e_out[i] = e;
    float ep = p.w0 * e;
    if (p.h1) ep += p.w1 * p.h1[i];
    if (p.h2) ep += p.w2 * p.h2[i];
    if (p.h3) ep += p.w3 * p.h3[i];
    const float x0 = (p.x[i] - p.sqrt_one_minus_at * ep) / p.sqrt_at;
    float xp = p.sqrt_a_prev * x0 + p.dir_coef * ep;
    if (p.noise) xp += p.sigma * p.noise[i];
    p.x_prev[i] = xp;
    if (p.pred_x0) p.pred_x0[i] = x0;
  }
}
}  // namespace

extern "C" int adm_sd_step(const float* x, const float* eps_uncond, const float* eps_cond, const float* h1, const float* h2,
                           const float* h3, const float* noise, float* x_prev, float* pred_x0, float* e_out, int64_t numel,
                           const adm_sd_step_coefs* c, void* stream) {
  ADM_REQUIRE(x && eps_cond && x_prev && c, ADM_E_ARG, "adm_sd_step: null pointer");
  ADM_REQUIRE(numel > 0, ADM_E_ARG, "adm_sd_step: empty tensor");
  ADM_REQUIRE(c->sqrt_at > 0.f, ADM_E_ARG, "adm_sd_step: sqrt(alpha_t) must be positive");
  SdStep p{x, eps_uncond, eps_cond, h1, h2, h3, noise, x_prev, pred_x0, e_out, (long long)numel,
           c->cfg_scale, c->w[0], c->w[1], c->w[2], c->w[3], c->sqrt_one_minus_at, c->sqrt_at, c->sqrt_a_prev, c->dir_coef, c->sigma};
  long long blocks = (numel + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sd_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return adm_check_launch("adm_sd_step");
}

// ------------------------------------------------------------------------------------------------
// One multistep DPM-Solver++ update in data-prediction form (dpm_solver.py:755-810 second order, 700-735 first order;
// model_wrapper's classifier-free guidance + data_prediction_fn :289-330, 380-400):
//   e = eu ? eu + cfg*(ec - eu) : ec;   m = (x - sigma_s*e) / alpha_s          (kept: the next step's model_prev)
//   x_next = a*x + b0*m + b1*m_prev
namespace {
struct DpmStep {
  const float* x; const float* eu; const float* ec; const float* m_prev;
  float* x_next; float* m_out;
  long long numel;
  float cfg, sigma_s, alpha_s, a, b0, b1;
};
__global__ void __launch_bounds__(256)
dpm_step_kernel(const DpmStep p) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.numel; i += (long long)gridDim.x * blockDim.x) {
    float e = p.ec[i];
    if (p.eu) {
      const float u = p.eu[i];
      e = u + p.cfg * (e - u);
    }
    const float x = p.x[i];
    const float m = (x - p.sigma_s * e) / p.alpha_s;
    float xn = p.a * x + p.b0 * m;
    if (p.m_prev) xn += p.b1 * p.m_prev[i];
    p.x_next[i] = xn;
    if (p.m_out) p.m_out[i] = m;
  }
}
}  // namespace

extern "C" int adm_dpm_step(const float* x, const float* eps_uncond, const float* eps_cond, const float* m_prev, float* x_next,
                            float* m_out, int64_t numel, float cfg_scale, float sigma_s, float alpha_s, float a, float b0,
                            float b1, void* stream) {
  ADM_REQUIRE(x && eps_cond && x_next, ADM_E_ARG, "adm_dpm_step: null pointer");
  ADM_REQUIRE(numel > 0 && alpha_s > 0.f, ADM_E_ARG, "adm_dpm_step: empty tensor or alpha_s <= 0");
  DpmStep p{x, eps_uncond, eps_cond, m_prev, x_next, m_out, (long long)numel, cfg_scale, sigma_s, alpha_s, a, b0, b1};
  long long blocks = (numel + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dpm_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return adm_check_launch("adm_dpm_step");
}
