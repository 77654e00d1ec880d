// The classifier's other heads (EncoderUNetModel pool = "adaptive" | "spatial" | "spatial_v2", reference
// guided_diffusion/unet.py:826-856, 880-896) and their backward-data pieces.  Everything here is small ([N, C] vectors, 8x8 .. 64x64
// maps of <= 1024 channels once per evaluation) and HBM / latency bound; the Linear layers between these kernels are adm_linear_f32.
//
//   adaptive:    out = Flatten(conv1x1(AvgPool(SiLU(GN(h)))))                         adm_channel_mean (with the GN affine) -> Linear
//   spatial:     f = cat_blocks(mean_pixels(h_block));  out = Linear(ReLU(Linear(f)))  adm_channel_mean (raw) -> Linear -> adm_vec_act
//   spatial_v2:  out = Linear(SiLU(GroupNorm32(32, 2048)(Linear(f))))                  ... -> adm_vec_gn -> Linear(silu_in)
//   backward:    d mean / d h = 1 / HW broadcast over the pixels                        adm_bcast_add (into the running gradient)
//                ReLU' / SiLU' on vectors                                               adm_vec_act_bwd
//                GroupNorm backward on vectors                                          adm_vec_gn_bwd
#include "adm_common.h"

namespace {

__device__ __forceinline__ float silu_grad_v(float z) {
  const float s = 1.0f / (1.0f + expf(-z));
  return s * (1.0f + z * (1.0f - s));
}

// out[n, col0 + ch] = mean over pixels of act(h[n, p, ch]); act = SiLU(a*h + b) with the affine, identity without.
// grid (ceil(c / 64), n), block 256 = 64 channels x 4 pixel lanes (a wave reads 128 contiguous bytes of a pixel row).
__global__ void __launch_bounds__(256)
channel_mean_kernel(const uint16_t* __restrict__ h, const float* __restrict__ aa, const float* __restrict__ ab,
                    float* __restrict__ out, int out_stride, int hw, int c) {
  __shared__ float red[4][64];
  const int img = blockIdx.y, ch = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  float s = 0.f;
  if (ch < c) {
    const float a = aa ? aa[(long long)img * c + ch] : 1.f, b = aa ? ab[(long long)img * c + ch] : 0.f;
    for (int p = pl; p < hw; p += 4) {
      float v = adm_h_to_f32(h[((long long)img * hw + p) * c + ch]);
      if (aa) v = adm_silu(a * v + b);
      s += v;
    }
  }
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && ch < c)
    out[(long long)img * out_stride + ch] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)hw;
}

// out[n, p, ch] = (add ? add[n, p, ch] : 0) + v[n, ch] * scale     (16-bit NHWC; v fp32 with row stride v_stride)
__global__ void __launch_bounds__(256)
bcast_add_kernel(const float* __restrict__ v, int v_stride, float scale, const uint16_t* __restrict__ add,
                 uint16_t* __restrict__ out, int n, int hw, int c) {
  const int cg = c / 8;
  const long long items = (long long)n * hw * cg;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % cg);
    const long long pix = it / cg;
    const int img = (int)(pix / hw);
    const float* vp = v + (long long)img * v_stride + g * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = vp[j] * scale;
    if (add) {
      const uint4 av = *reinterpret_cast<const uint4*>(add + pix * c + g * 8);
      const uint32_t au[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[2 * j] += adm_lo_f32(au[j]); f[2 * j + 1] += adm_hi_f32(au[j]); }
    }
    uint4 o;
    o.x = adm_pack2(f[0], f[1]); o.y = adm_pack2(f[2], f[3]); o.z = adm_pack2(f[4], f[5]); o.w = adm_pack2(f[6], f[7]);
    *reinterpret_cast<uint4*>(out + pix * c + g * 8) = o;
  }
}

// mode 1: SiLU, 2: ReLU.  out = act(x)   |   out = dy * act'(z)
__global__ void __launch_bounds__(256)
vec_act_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, long long items, int mode) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long long)gridDim.x * blockDim.x) {
    const float z = x[i];
    if (dy) out[i] = dy[i] * (mode == 1 ? silu_grad_v(z) : (z > 0.f ? 1.f : 0.f));
    else out[i] = mode == 1 ? z / (1.0f + expf(-z)) : fmaxf(z, 0.f);
  }
}

// GroupNorm32(32, C) of a [N, C] vector (spatial_v2's normalization(2048) sees no spatial axis): one block per row, one wave-half
// per group pair; y = gamma * (x - mean_g) * rstd_g + beta, stats[n][g] = (mean, rstd).  Two-pass variance in fp32 (<= 64 values).
__global__ void __launch_bounds__(256)
vec_gn_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
              float* __restrict__ y, float* __restrict__ stats, int c, float eps) {
  const int img = blockIdx.x, cpg = c / 32;
  const int grp = threadIdx.x / 8, sub = threadIdx.x % 8;   // 32 groups x 8 lanes
  const float* xr = x + (long long)img * c + grp * cpg;
  float s = 0.f;
  for (int j = sub; j < cpg; j += 8) s += xr[j];
  for (int off = 4; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)cpg;
  float ss = 0.f;
  for (int j = sub; j < cpg; j += 8) { const float d = xr[j] - mean; ss += d * d; }
  for (int off = 4; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
  const float rstd = 1.0f / sqrtf(ss / (float)cpg + eps);
  if (sub == 0) {
    stats[((long long)img * 32 + grp) * 2 + 0] = mean;
    stats[((long long)img * 32 + grp) * 2 + 1] = rstd;
  }
  for (int j = sub; j < cpg; j += 8) {
    const int ch = grp * cpg + j;
    y[(long long)img * c + ch] = gamma[ch] * (xr[j] - mean) * rstd + beta[ch];
  }
}

// dx = rstd * (gamma dz - mean_g(gamma dz) - xhat * mean_g(gamma dz xhat)),  xhat = (x - mean) * rstd
__global__ void __launch_bounds__(256)
vec_gn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ stats,
                  const float* __restrict__ dz, float* __restrict__ dx, int c) {
  const int img = blockIdx.x, cpg = c / 32;
  const int grp = threadIdx.x / 8, sub = threadIdx.x % 8;
  const float mean = stats[((long long)img * 32 + grp) * 2 + 0], rstd = stats[((long long)img * 32 + grp) * 2 + 1];
  const long long base = (long long)img * c + grp * cpg;
  float s1 = 0.f, s2 = 0.f;
  for (int j = sub; j < cpg; j += 8) {
    const float g = gamma[grp * cpg + j] * dz[base + j], xh = (x[base + j] - mean) * rstd;
    s1 += g;
    s2 += g * xh;
  }
  for (int off = 4; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  s1 /= (float)cpg;
  s2 /= (float)cpg;
  for (int j = sub; j < cpg; j += 8) {
    const float g = gamma[grp * cpg + j] * dz[base + j], xh = (x[base + j] - mean) * rstd;
    dx[base + j] = rstd * (g - s1 - xh * s2);
  }
}

int blocks_for(long long items) {
  long long b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int adm_channel_mean(const adm_bf16* h, const float* aff_a, const float* aff_b, float* out, int out_stride, int n,
                                int hw, int c, void* stream) {
  ADM_REQUIRE(h && out, ADM_E_ARG, "adm_channel_mean: null pointer");
  ADM_REQUIRE((aff_a != nullptr) == (aff_b != nullptr), ADM_E_ARG, "adm_channel_mean: aff_a / aff_b go together");
  ADM_REQUIRE(n > 0 && n <= 65535 && hw > 0 && c > 0 && out_stride >= c, ADM_E_SHAPE, "adm_channel_mean: bad shape");
  hipLaunchKernelGGL(channel_mean_kernel, dim3((c + 63) / 64, n), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint16_t*>(h), aff_a, aff_b, out, out_stride, hw, c);
  return adm_check_launch("adm_channel_mean");
}

extern "C" int adm_bcast_add(const float* v, int v_stride, float scale, const adm_bf16* add, adm_bf16* out, int n, int hw, int c,
                             void* stream) {
  ADM_REQUIRE(v && out, ADM_E_ARG, "adm_bcast_add: null pointer");
  ADM_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 8 == 0 && v_stride >= c, ADM_E_SHAPE, "adm_bcast_add: bad shape (c %% 8 == 0)");
  ADM_REQUIRE(adm_aligned16(add) && adm_aligned16(out), ADM_E_ALIGN, "adm_bcast_add: unaligned pointer");
  hipLaunchKernelGGL(bcast_add_kernel, dim3(blocks_for((long long)n * hw * (c / 8))), dim3(256), 0, (hipStream_t)stream, v, v_stride,
                     scale, reinterpret_cast<const uint16_t*>(add), reinterpret_cast<uint16_t*>(out), n, hw, c);
  return adm_check_launch("adm_bcast_add");
}

extern "C" int adm_vec_act(const float* x, const float* dy, float* out, int64_t items, int mode, void* stream) {
  ADM_REQUIRE(x && out, ADM_E_ARG, "adm_vec_act: null pointer");
  ADM_REQUIRE(items > 0 && (mode == 1 || mode == 2), ADM_E_ARG, "adm_vec_act: items > 0, mode 1 (SiLU) or 2 (ReLU)");
  hipLaunchKernelGGL(vec_act_kernel, dim3(blocks_for(items)), dim3(256), 0, (hipStream_t)stream, x, dy, out, (long long)items, mode);
  return adm_check_launch("adm_vec_act");
}

extern "C" int adm_vec_gn(const float* x, const float* gamma, const float* beta, float* y, float* stats, int n, int c, float eps,
                          void* stream) {
  ADM_REQUIRE(x && gamma && beta && y && stats, ADM_E_ARG, "adm_vec_gn: null pointer");
  ADM_REQUIRE(n > 0 && c > 0 && c % 32 == 0, ADM_E_SHAPE, "adm_vec_gn: c must be a multiple of the 32 groups");
  hipLaunchKernelGGL(vec_gn_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats, c, eps);
  return adm_check_launch("adm_vec_gn");
}

extern "C" int adm_vec_gn_bwd(const float* x, const float* gamma, const float* stats, const float* dz, float* dx, int n, int c,
                              void* stream) {
  ADM_REQUIRE(x && gamma && stats && dz && dx, ADM_E_ARG, "adm_vec_gn_bwd: null pointer");
  ADM_REQUIRE(n > 0 && c > 0 && c % 32 == 0, ADM_E_SHAPE, "adm_vec_gn_bwd: c must be a multiple of the 32 groups");
  hipLaunchKernelGGL(vec_gn_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, x, gamma, stats, dz, dx, c);
  return adm_check_launch("adm_vec_gn_bwd");
}
