// K9: fused per-pixel sampler update (DDIM eq. 12 / DDPM ancestral) + uint8 NHWC pack.
//
// Replaces ~25 elementwise torch kernels and >= 8 H2D coefficient copies per step of
// GaussianDiffusion.p_mean_variance / condition_score / condition_mean / ddim_sample /
// p_sample (reference guided_diffusion/gaussian_diffusion.py:232-439, 536-584).
// HBM-bound: reads x, model_out (C or 2C channels), [grad], [noise]; writes x_prev
// (+ pred_xstart, + uint8 image).  One thread owns VEC consecutive pixels of one image and
// walks the C channels, so fp32 NCHW reads are coalesced 16 B/lane and the uint8 NHWC write
// is C*VEC contiguous bytes per lane.
#include "adm_common.h"

namespace {

struct StepK {
  float A, Bm, sqrt_one_minus_ac, sqrt_ac_prev, dir_coef, sigma_nz;  // ddim
  float c1, c2, lo, hi, fixed_var, noise_nz;                           // ddpm
  int learned_range, predict_xstart, clip;
};

template <bool DDIM, int VEC>
__global__ void __launch_bounds__(256)
step_kernel(const float* __restrict__ x, const float* __restrict__ mo, const float* __restrict__ grad,
            const float* __restrict__ noise, float* __restrict__ x_prev, float* __restrict__ x0_out,
            uint8_t* __restrict__ u8, int n, int c, int hw, StepK k) {
#pragma clang fp contract(off)
  const long long groups = (long long)n * (hw / VEC);
  const int mo_c = k.learned_range ? 2 * c : c;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups;
       g += (long long)gridDim.x * blockDim.x) {
    const int img = (int)(g / (hw / VEC));
    const int p = (int)(g % (hw / VEC)) * VEC;
    for (int ch = 0; ch < c; ++ch) {
      const long long xi = ((long long)img * c + ch) * hw + p;
      const long long mi = ((long long)img * mo_c + ch) * hw + p;
      float xv[VEC], ev[VEC], vv[VEC], gv[VEC], nv[VEC], outv[VEC], x0v[VEC];
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + xi);
        *reinterpret_cast<float4*>(ev) = *reinterpret_cast<const float4*>(mo + mi);
        if (!DDIM && k.learned_range)
          *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(mo + mi + (long long)c * hw);
        if (grad) *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(grad + xi);
        if (noise) *reinterpret_cast<float4*>(nv) = *reinterpret_cast<const float4*>(noise + xi);
      } else {
        xv[0] = x[xi];
        ev[0] = mo[mi];
        if (!DDIM && k.learned_range) vv[0] = mo[mi + (long long)c * hw];
        if (grad) gv[0] = grad[xi];
        if (noise) nv[0] = noise[xi];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float x0 = k.predict_xstart ? ev[j] : (k.A * xv[j] - k.Bm * ev[j]);
        if (k.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        if constexpr (DDIM) {
          if (grad) {  // condition_score: x0 is not re-clamped (gaussian_diffusion.py:381-393)
            float e = (k.A * xv[j] - x0) / k.Bm;
            e = e - k.sqrt_one_minus_ac * gv[j];
            x0 = k.A * xv[j] - k.Bm * e;
          }
          const float e = (k.A * xv[j] - x0) / k.Bm;
          float s = x0 * k.sqrt_ac_prev + k.dir_coef * e;
          if (noise) s = s + k.sigma_nz * nv[j];
          outv[j] = s;
        } else {
          float var, logvar;
          if (k.learned_range) {
            const float frac = (vv[j] + 1.0f) / 2.0f;
            logvar = frac * k.hi + (1.0f - frac) * k.lo;
            var = expf(logvar);
          } else {
            logvar = k.lo;
            var = k.fixed_var;
          }
          float mean = k.c1 * x0 + k.c2 * xv[j];
          if (grad) mean = mean + var * gv[j];  // condition_mean
          float s = mean;
          if (noise) s = s + k.noise_nz * expf(0.5f * logvar) * nv[j];
          outv[j] = s;
        }
        x0v[j] = x0;
      }
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(x_prev + xi) = *reinterpret_cast<float4*>(outv);
        if (x0_out) *reinterpret_cast<float4*>(x0_out + xi) = *reinterpret_cast<float4*>(x0v);
      } else {
        x_prev[xi] = outv[0];
        if (x0_out) x0_out[xi] = x0v[0];
      }
      if (u8) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float q = (outv[j] + 1.0f) * 127.5f;
          q = fminf(fmaxf(q, 0.0f), 255.0f);
          u8[((long long)img * hw + p + j) * c + ch] = (uint8_t)q;  // truncation, as .to(uint8)
        }
      }
    }
  }
}

template <int VEC>
__global__ void __launch_bounds__(256)
pack_kernel(const float* __restrict__ x, uint8_t* __restrict__ u8, int n, int c, int hw) {
#pragma clang fp contract(off)
  const long long groups = (long long)n * (hw / VEC);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups;
       g += (long long)gridDim.x * blockDim.x) {
    const int img = (int)(g / (hw / VEC));
    const int p = (int)(g % (hw / VEC)) * VEC;
    for (int ch = 0; ch < c; ++ch) {
      const long long xi = ((long long)img * c + ch) * hw + p;
      float v[VEC];
      if constexpr (VEC == 4) *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(x + xi);
      else v[0] = x[xi];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float q = (v[j] + 1.0f) * 127.5f;
        q = fminf(fmaxf(q, 0.0f), 255.0f);
        u8[((long long)img * hw + p + j) * c + ch] = (uint8_t)q;
      }
    }
  }
}

int launch_step(bool ddim, const float* x, const float* mo, const float* grad, const float* noise,
                float* x_prev, float* x0, uint8_t* u8, int n, int c, int h, int w,
                const adm_step_coefs* cf, void* stream) {
  ADM_REQUIRE(x && mo && x_prev && cf, ADM_E_ARG, "adm_%s_step: null pointer", ddim ? "ddim" : "ddpm");
  ADM_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, ADM_E_ARG, "adm_step: bad shape %d %d %d %d", n, c, h, w);
  StepK k{};
  // float32 scalar arithmetic, same operation order as the reference's tensor expressions
  k.A = cf->sqrt_recip_ac;
  k.Bm = cf->sqrt_recipm1_ac;
  k.sqrt_one_minus_ac = sqrtf(1.0f - cf->ac);
  k.sqrt_ac_prev = sqrtf(cf->ac_prev);
  const float sigma = cf->eta * sqrtf((1.0f - cf->ac_prev) / (1.0f - cf->ac)) * sqrtf(1.0f - cf->ac / cf->ac_prev);
  k.dir_coef = sqrtf(1.0f - cf->ac_prev - sigma * sigma);
  k.sigma_nz = cf->nonzero ? sigma : 0.0f;
  k.c1 = cf->coef1;
  k.c2 = cf->coef2;
  k.lo = cf->log_var_lo;
  k.hi = cf->log_var_hi;
  k.fixed_var = cf->fixed_var;
  k.noise_nz = cf->nonzero ? 1.0f : 0.0f;
  k.learned_range = cf->learned_range;
  k.predict_xstart = cf->predict_xstart;
  k.clip = cf->clip_denoised;
  ADM_REQUIRE(noise || (ddim ? k.sigma_nz == 0.0f : !cf->nonzero), ADM_E_ARG,
              "adm_step: this step adds noise but noise == NULL");
  // noise multiplied by exactly 0 contributes nothing: skip the read
  const float* nz_ptr = noise;
  if (ddim && k.sigma_nz == 0.0f) nz_ptr = nullptr;
  if (!ddim && !cf->nonzero) nz_ptr = nullptr;
  const int hw = h * w;
  const bool vec4 = (hw % 4 == 0) && adm_aligned16(x) && adm_aligned16(mo) && adm_aligned16(x_prev) &&
                    (!grad || adm_aligned16(grad)) && (!nz_ptr || adm_aligned16(nz_ptr)) &&
                    (!x0 || adm_aligned16(x0));
  const long long groups = (long long)n * (vec4 ? hw / 4 : hw);
  int blocks = (int)((groups + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(D, V) \
  hipLaunchKernelGGL((step_kernel<D, V>), dim3(blocks), dim3(256), 0, s, x, mo, grad, nz_ptr, x_prev, x0, u8, n, c, hw, k)
  if (ddim) { if (vec4) LAUNCH(true, 4); else LAUNCH(true, 1); }
  else      { if (vec4) LAUNCH(false, 4); else LAUNCH(false, 1); }
#undef LAUNCH
  return adm_check_launch(ddim ? "adm_ddim_step" : "adm_ddpm_step");
}

}  // namespace

extern "C" int adm_ddim_step(const float* x, const float* model_out, const float* grad, const float* noise,
                             float* x_prev, float* pred_xstart, uint8_t* u8_nhwc, int n, int c, int h, int w,
                             const adm_step_coefs* coefs_host, void* stream) {
  return launch_step(true, x, model_out, grad, noise, x_prev, pred_xstart, u8_nhwc, n, c, h, w, coefs_host, stream);
}

extern "C" int adm_ddpm_step(const float* x, const float* model_out, const float* grad, const float* noise,
                             float* x_prev, float* pred_xstart, uint8_t* u8_nhwc, int n, int c, int h, int w,
                             const adm_step_coefs* coefs_host, void* stream) {
  return launch_step(false, x, model_out, grad, noise, x_prev, pred_xstart, u8_nhwc, n, c, h, w, coefs_host, stream);
}

extern "C" int adm_pack_u8_nhwc(const float* x, uint8_t* out, int n, int c, int h, int w, void* stream) {
  ADM_REQUIRE(x && out, ADM_E_ARG, "adm_pack_u8_nhwc: null pointer");
  ADM_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, ADM_E_ARG, "adm_pack_u8_nhwc: bad shape");
  const int hw = h * w;
  const bool vec4 = (hw % 4 == 0) && adm_aligned16(x);
  const long long groups = (long long)n * (vec4 ? hw / 4 : hw);
  int blocks = (int)((groups + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (vec4) hipLaunchKernelGGL((pack_kernel<4>), dim3(blocks), dim3(256), 0, s, x, out, n, c, hw);
  else hipLaunchKernelGGL((pack_kernel<1>), dim3(blocks), dim3(256), 0, s, x, out, n, c, hw);
  return adm_check_launch("adm_pack_u8_nhwc");
}
