// K10: backward-data helpers for the classifier-guidance gradient d log p(y|x,t) / dx.
//
// The reference builds this gradient with torch.autograd through EncoderUNetModel
// (search_imagenet64_classifier_guidance.py:319-326; guided_diffusion/unet.py:685-896).  On the HIP
// path the backward network is explicit: conv backward-data is the SAME MFMA conv kernel run with
// transposed + tap-flipped weights (adm_pack_conv_weight_bwd), attention backward lives in
// adm_attention_bwd.hip, and this file holds the HBM-bound pieces:
//   * GroupNorm(+FiLM)+SiLU backward: y = SiLU(z), z = a*x + b with a,b from adm_gn_finalize
//       dz = dy * SiLU'(z);   dx = a*dz + k1*x + k0
//       k1 = -rstd^2 * S2/m,  k0 = -rstd*S1/m - mean*k1,
//       S1 = sum_g (a/rstd)*dz,  S2 = sum_g a*dz*(x - mean)        (m = elements per group)
//     as partial (per slab, per channel) -> finalize (per image) -> apply (elementwise), mirroring
//     the forward statistics kernels; the AvgPool2d backward of down-sampling ResBlocks is folded
//     into the index math (dy / add read at half resolution, x 0.25)
//   * d logits = scale * (onehot(y) - softmax(logits))
//   * AttentionPool2d (unet.py:22-51) forward/backward around its single used query (token 0)
#include "adm_common.h"

namespace {

__device__ __forceinline__ float silu_grad(float z) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
  return s * (1.0f + z * (1.0f - s));
}

// dy lookup: mode 0 same resolution, mode 1 half resolution (x 0.25, AvgPool2d backward)
__device__ __forceinline__ long long src_pixel(int mode, int img, int y, int x, int h, int w) {
  if (mode == 0) return ((long long)img * h + y) * w + x;
  return ((long long)img * (h / 2) + y / 2) * (w / 2) + x / 2;
}

// grid (slabs, n): per-channel T1 = sum dz, T2 = sum dz * x over the slab's pixels
__global__ void __launch_bounds__(256)
gn_bwd_partial_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, const float* __restrict__ aa,
                      const float* __restrict__ ab, float* __restrict__ partial, int h, int w, int c, int slabs,
                      int silu, int dy_mode) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int hw = h * w;
  const int groups8 = c / 8;
  const int lanes = blockDim.x / groups8;
  const int lane = threadIdx.x / groups8, g8 = threadIdx.x % groups8;
  const int slab = blockIdx.x, img = blockIdx.y;
  const int per = (hw + slabs - 1) / slabs;
  const int p_begin = slab * per, p_end = min(hw, p_begin + per);
  const float dys = dy_mode ? 0.25f : 1.0f;
  float t1[8] = {}, t2[8] = {};
  if (lane < lanes) {
    const int ch = g8 * 8;
    float a8[8], b8[8];
    *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(aa + (long long)img * c + ch);
    *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(aa + (long long)img * c + ch + 4);
    *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(ab + (long long)img * c + ch);
    *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(ab + (long long)img * c + ch + 4);
#pragma unroll 2
    for (int p = p_begin + lane; p < p_end; p += lanes) {
      const uint4 xv = *reinterpret_cast<const uint4*>(x + ((long long)img * hw + p) * c + ch);
      const uint4 gv = *reinterpret_cast<const uint4*>(dy + src_pixel(dy_mode, img, p / w, p % w, h, w) * c + ch);
      const uint32_t xu[4] = {xv.x, xv.y, xv.z, xv.w}, gu[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = adm_lo_f32(xu[j]), x1 = adm_hi_f32(xu[j]);
        float g0 = adm_lo_f32(gu[j]) * dys, g1 = adm_hi_f32(gu[j]) * dys;
        if (silu) {
          g0 *= silu_grad(a8[2 * j] * x0 + b8[2 * j]);
          g1 *= silu_grad(a8[2 * j + 1] * x1 + b8[2 * j + 1]);
        }
        t1[2 * j] += g0; t2[2 * j] += g0 * x0;
        t1[2 * j + 1] += g1; t2[2 * j + 1] += g1 * x1;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[((long long)lane * c + ch + j) * 2 + 0] = t1[j];
      red[((long long)lane * c + ch + j) * 2 + 1] = t2[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c * 2; i += blockDim.x) {
    float t = 0.0f;
    for (int l = 0; l < lanes; ++l) t += red[(long long)l * c * 2 + i];
    partial[(((long long)img * slabs + slab) * c) * 2 + i] = t;
  }
}

// grid (n, 4): k1, k0 per (image, channel); a block takes 8 of the 32 groups, half a wave per group.
// The launch is dependent-latency bound (a few hundred 8-byte loads per group), so the work of a group is spread over
// 32 lanes with four loads in flight each: one thread-quarter per group and one load in flight took 52 us per call,
// 2.9 % of the guided step (profiles/r02/bench_guided_b256_b_kernel_stats.csv).
__global__ void __launch_bounds__(256)
gn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ aa, const float* __restrict__ stats,
                       const float* __restrict__ add, int add_stride, float* __restrict__ k1, float* __restrict__ k0, int c, int hw,
                       int slabs) {
  const int img = blockIdx.x;
  const int cpg = c / 32;
  const int grp = blockIdx.y * 8 + threadIdx.x / 32, sub = threadIdx.x % 32;
  const float mean = stats[((long long)img * 32 + grp) * 2 + 0], rstd = stats[((long long)img * 32 + grp) * 2 + 1];
  const float* pb = partial + ((long long)img * slabs * c + (long long)grp * cpg) * 2;
  const float* ab = aa + (long long)img * c + grp * cpg;
  const int items = slabs * cpg;     // item i = (slab i / cpg, channel-in-group i % cpg)
  double s1 = 0.0, s2 = 0.0;
  // add: the layer normalised x + e[img, ch] (use_scale_shift_norm = False) while x is what was stored and summed:
  // sum dz (x + e) = sum dz x + e sum dz, and dx = a dz + k1 (x + e) + k0 keeps its form with k0 += k1 e
  const float* eb = add ? add + (long long)img * add_stride + grp * cpg : nullptr;
  auto at = [&](int i, float2& v, float& a) {
    const int sl = i / cpg, j = i - sl * cpg;
    v = *reinterpret_cast<const float2*>(pb + ((long long)sl * c + j) * 2);
    if (eb) v.y += eb[j] * v.x;
    a = ab[j];
  };
  auto acc = [&](const float2& v, float a) {
    s1 += (double)a * (double)v.x;
    s2 += (double)a * ((double)v.y - (double)mean * (double)v.x);
  };
  int i = sub;
  for (; i + 96 < items; i += 128) {
    float2 v0, v1, v2, v3;
    float a0, a1, a2, a3;
    at(i, v0, a0); at(i + 32, v1, a1); at(i + 64, v2, a2); at(i + 96, v3, a3);
    acc(v0, a0); acc(v1, a1); acc(v2, a2); acc(v3, a3);
  }
  for (; i < items; i += 32) {
    float2 v;
    float a;
    at(i, v, a);
    acc(v, a);
  }
  s1 /= (double)rstd;
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
    s1 += __shfl_xor(s1, off);
    s2 += __shfl_xor(s2, off);
  }
  const double m = (double)cpg * (double)hw;
  const double kk1 = -(double)rstd * (double)rstd * s2 / m;
  const float f1 = (float)kk1, f0 = (float)(-(double)rstd * s1 / m - (double)mean * kk1);
  for (int j = sub; j < cpg; j += 32) {
    k1[(long long)img * c + grp * cpg + j] = f1;
    k0[(long long)img * c + grp * cpg + j] = eb ? f0 + f1 * eb[j] : f0;
  }
}

// dx = a*dz + k1*x + k0 (+ add).  grid (slabs, n), the thread layout of the partial pass: a thread keeps ONE 8-channel
// group of its image and walks the slab's pixels, so the four per-(image, channel) coefficient vectors are loaded once
// per thread instead of once per 16 bytes of x (a flat grid-stride version issued 128 B of coefficient loads per 48 B of
// tensor loads and streamed at 4.7 TB/s), and several pixels' loads are in flight per thread.
__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, const float* __restrict__ aa,
                    const float* __restrict__ ab, const float* __restrict__ k1, const float* __restrict__ k0,
                    const uint16_t* __restrict__ add, uint16_t* __restrict__ out, int h, int w, int c, int slabs,
                    int silu, int dy_mode, int add_mode) {
  const int hw = h * w;
  const int groups8 = c / 8;
  const int lanes = blockDim.x / groups8;
  const int lane = threadIdx.x / groups8, g8 = threadIdx.x % groups8;
  if (lane >= lanes) return;
  const int slab = blockIdx.x, img = blockIdx.y;
  const int per = (hw + slabs - 1) / slabs;
  const int p_begin = slab * per, p_end = min(hw, p_begin + per);
  const float dys = dy_mode ? 0.25f : 1.0f, adds = add_mode ? 0.25f : 1.0f;
  const int ch = g8 * 8;
  const long long cb = (long long)img * c + ch;
  float a8[8], b8[8], k18[8], k08[8];
  *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(aa + cb);
  *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(aa + cb + 4);
  *reinterpret_cast<float4*>(k18) = *reinterpret_cast<const float4*>(k1 + cb);
  *reinterpret_cast<float4*>(k18 + 4) = *reinterpret_cast<const float4*>(k1 + cb + 4);
  *reinterpret_cast<float4*>(k08) = *reinterpret_cast<const float4*>(k0 + cb);
  *reinterpret_cast<float4*>(k08 + 4) = *reinterpret_cast<const float4*>(k0 + cb + 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) b8[j] = 0.0f;
  if (silu) {
    *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(ab + cb);
    *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(ab + cb + 4);
  }
#pragma unroll 2
  for (int p = p_begin + lane; p < p_end; p += lanes) {
    const long long pix = (long long)img * hw + p;
    const int py = p / w, px = p - py * w;
    const uint4 xv = *reinterpret_cast<const uint4*>(x + pix * c + ch);
    const uint4 gv = *reinterpret_cast<const uint4*>(dy + src_pixel(dy_mode, img, py, px, h, w) * c + ch);
    uint4 av = make_uint4(0, 0, 0, 0);
    if (add) av = *reinterpret_cast<const uint4*>(add + src_pixel(add_mode, img, py, px, h, w) * c + ch);
    const uint32_t xu[4] = {xv.x, xv.y, xv.z, xv.w}, gu[4] = {gv.x, gv.y, gv.z, gv.w}, au[4] = {av.x, av.y, av.z, av.w};
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xx = (j & 1) ? adm_hi_f32(xu[j >> 1]) : adm_lo_f32(xu[j >> 1]);
      float gg = ((j & 1) ? adm_hi_f32(gu[j >> 1]) : adm_lo_f32(gu[j >> 1])) * dys;
      const float ad = ((j & 1) ? adm_hi_f32(au[j >> 1]) : adm_lo_f32(au[j >> 1])) * adds;
      if (silu) gg *= silu_grad(a8[j] * xx + b8[j]);
      r[j] = a8[j] * gg + k18[j] * xx + k08[j] + ad;
    }
    uint4 pk;
    pk.x = adm_pack2(r[0], r[1]);
    pk.y = adm_pack2(r[2], r[3]);
    pk.z = adm_pack2(r[4], r[5]);
    pk.w = adm_pack2(r[6], r[7]);
    *reinterpret_cast<uint4*>(out + pix * c + ch) = pk;
  }
}

// out = a (+ b at same or half resolution x0.25): plain bf16 gradient accumulation
__global__ void __launch_bounds__(256)
add_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ out, int n, int h,
           int w, int c, int b_mode) {
  const int cg = c / 8;
  const long long items = (long long)n * h * w * cg;
  const float bs = b_mode ? 0.25f : 1.0f;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items;
       it += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(it % cg);
    const long long pix = it / cg;
    const int px = (int)(pix % w), py = (int)((pix / w) % h), img = (int)(pix / ((long long)w * h));
    const uint4 av = *reinterpret_cast<const uint4*>(a + pix * c + g * 8);
    const uint4 bv = *reinterpret_cast<const uint4*>(b + src_pixel(b_mode, img, py, px, h, w) * c + g * 8);
    const uint32_t au[4] = {av.x, av.y, av.z, av.w}, bu[4] = {bv.x, bv.y, bv.z, bv.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = adm_lo_f32(au[j]) + bs * adm_lo_f32(bu[j]);
      const float hi = adm_hi_f32(au[j]) + bs * adm_hi_f32(bu[j]);
      o[j] = adm_pack2(lo, hi);
    }
    *reinterpret_cast<uint4*>(out + pix * c + g * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// one block per row: dl = scale * (onehot - softmax)
__global__ void __launch_bounds__(256)
logsoftmax_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ y, float* __restrict__ dl,
                       float* __restrict__ logp_sel, int k, float scale) {
  __shared__ float red[256];
  const int row = blockIdx.x;
  const float* l = logits + (long long)row * k;
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < k; i += blockDim.x) mx = fmaxf(mx, l[i]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < k; i += blockDim.x) sum += expf(l[i] - mx);
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  sum = red[0];
  const int yy = (int)y[row];
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const float sm = expf(l[i] - mx) / sum;
    dl[(long long)row * k + i] = scale * ((i == yy ? 1.0f : 0.0f) - sm);
  }
  if (logp_sel && threadIdx.x == 0) logp_sel[row] = l[yy] - mx - logf(sum);
}

// ---- AttentionPool2d -------------------------------------------------------------------------
// tok[n, 0, :] = mean_p act + pos[:, 0];  tok[n, 1+p, :] = act[n, p, :] + pos[:, 1+p];  rows >= T zero.
// act = SiLU(a*h + b) (the classifier's `out` GroupNorm + SiLU).  grid (n), block = C/8 lanes x pixels.
__global__ void __launch_bounds__(256)
pool_prep_kernel(const uint16_t* __restrict__ hsrc, const float* __restrict__ aa, const float* __restrict__ ab,
                 const float* __restrict__ pos, uint16_t* __restrict__ tok, int hw, int c, int tpad) {
  const int img = blockIdx.x;
  const int T = hw + 1;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    const float a = aa[(long long)img * c + ch], b = ab[(long long)img * c + ch];
    float mean = 0.f;
    for (int p = 0; p < hw; ++p) {
      const float v = adm_silu(a * adm_h_to_f32(hsrc[((long long)img * hw + p) * c + ch]) + b);
      mean += v;
      tok[((long long)img * tpad + 1 + p) * c + ch] = adm_f32_to_h(v + pos[(long long)ch * T + 1 + p]);
    }
    tok[((long long)img * tpad) * c + ch] = adm_f32_to_h(mean / (float)hw + pos[(long long)ch * T]);
    for (int p = T; p < tpad; ++p) tok[((long long)img * tpad + p) * c + ch] = 0;
  }
}

// grid (heads, n), 64 threads: attention of query token 0 over T keys. a0 fp32 [N, C]; wts fp32 [N, H, tpad]
__global__ void __launch_bounds__(64)
pool_attn_fwd_kernel(const uint16_t* __restrict__ qkv, float* __restrict__ a0, float* __restrict__ wts, int T,
                     int tpad, int heads, int d) {
  extern __shared__ float sh[];  // q0[d] | w[tpad]
  float* q0 = sh;
  float* w = sh + d;
  const int hd = blockIdx.x, img = blockIdx.y, c = heads * d, c3 = 3 * c;
  const uint16_t* base = qkv + (long long)img * tpad * c3;
  for (int j = threadIdx.x; j < d; j += 64) q0[j] = adm_h_to_f32(base[hd * d + j]);
  __syncthreads();
  const float scale = rsqrtf((float)d);
  float mx = -3.0e38f;
  for (int s = threadIdx.x; s < T; s += 64) {
    const uint16_t* kr = base + (long long)s * c3 + c + hd * d;
    float acc = 0.f;
    for (int j = 0; j < d; ++j) acc += q0[j] * adm_h_to_f32(kr[j]);
    acc *= scale;
    w[s] = acc;
    mx = fmaxf(mx, acc);
  }
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int s = threadIdx.x; s < T; s += 64) { const float e = expf(w[s] - mx); w[s] = e; sum += e; }
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  __syncthreads();
  for (int s = threadIdx.x; s < tpad; s += 64) {
    const float v = s < T ? w[s] / sum : 0.f;
    w[s] = v;
    wts[((long long)img * heads + hd) * tpad + s] = v;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < d; j += 64) {
    float acc = 0.f;
    for (int s = 0; s < T; ++s) acc += w[s] * adm_h_to_f32(base[(long long)s * c3 + 2 * c + hd * d + j]);
    a0[(long long)img * c + hd * d + j] = acc;
  }
}

// grid (heads, n), 64 threads: backward of the above. dqkv bf16 [N, tpad, 3C]: token 0 gets dq; tokens < T get dk, dv.
__global__ void __launch_bounds__(64)
pool_attn_bwd_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ wts, const float* __restrict__ da0,
                     uint16_t* __restrict__ dqkv, int T, int tpad, int heads, int d) {
  extern __shared__ float sh[];  // q0[d] | da[d] | dlogit[tpad] | dq[d]
  float* q0 = sh;
  float* da = sh + d;
  float* dlg = sh + 2 * d;
  const int hd = blockIdx.x, img = blockIdx.y, c = heads * d, c3 = 3 * c;
  const uint16_t* base = qkv + (long long)img * tpad * c3;
  uint16_t* obase = dqkv + (long long)img * tpad * c3;
  const float* w = wts + ((long long)img * heads + hd) * tpad;
  for (int j = threadIdx.x; j < d; j += 64) { q0[j] = adm_h_to_f32(base[hd * d + j]); da[j] = da0[(long long)img * c + hd * d + j]; }
  __syncthreads();
  const float scale = rsqrtf((float)d);
  const int d8 = d / 8;  // 16-byte segments per head row (d % 8 == 0)
  float delta = 0.f;
  for (int s = threadIdx.x; s < T; s += 64) {
    const uint16_t* vr = base + (long long)s * c3 + 2 * c + hd * d;
    float dw = 0.f;
    for (int j8 = 0; j8 < d8; ++j8) {
      const uint4 v = *reinterpret_cast<const uint4*>(vr + j8 * 8);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        dw += da[j8 * 8 + 2 * e] * adm_lo_f32(u[e]) + da[j8 * 8 + 2 * e + 1] * adm_hi_f32(u[e]);
    }
    dlg[s] = dw;
    delta += w[s] * dw;
  }
  for (int off = 32; off >= 1; off >>= 1) delta += __shfl_xor(delta, off);
  __syncthreads();
  for (int s = threadIdx.x; s < T; s += 64) dlg[s] = w[s] * (dlg[s] - delta) * scale;  // d logit_s / sqrt(d)
  __syncthreads();
  // dK, dV (and the zero dQ of tokens > 0, and the zero rows of the padding): 16-byte segments, lanes along a row
  for (int it = threadIdx.x; it < tpad * d8; it += 64) {
    const int s = it / d8, j8 = it % d8;
    uint16_t* orow = obase + (long long)s * c3 + hd * d + j8 * 8;
    uint32_t kk[4] = {0, 0, 0, 0}, vv[4] = {0, 0, 0, 0};
    if (s < T) {
      const float dl = dlg[s], ws = w[s];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j8 * 8 + 2 * e;
        kk[e] = adm_pack2(dl * q0[j], dl * q0[j + 1]);
        vv[e] = adm_pack2(ws * da[j], ws * da[j + 1]);
      }
    }
    *reinterpret_cast<uint4*>(orow + c) = make_uint4(kk[0], kk[1], kk[2], kk[3]);
    *reinterpret_cast<uint4*>(orow + 2 * c) = make_uint4(vv[0], vv[1], vv[2], vv[3]);
    if (s > 0) *reinterpret_cast<uint4*>(orow) = make_uint4(0, 0, 0, 0);  // dQ only exists for token 0
  }
  for (int j = threadIdx.x; j < d; j += 64) {
    float acc = 0.f;
    for (int s = 0; s < T; ++s) acc += dlg[s] * adm_h_to_f32(base[(long long)s * c3 + c + hd * d + j]);
    obase[hd * d + j] = adm_f32_to_h(acc);
  }
}

// d_act[n, p, :] = dtok[n, 1+p, :] + dtok[n, 0, :] / HW
__global__ void __launch_bounds__(256)
pool_prep_bwd_kernel(const uint16_t* __restrict__ dtok, uint16_t* __restrict__ dact, int n, int hw, int c, int tpad) {
  const long long items = (long long)n * hw * c;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(it % c);
    const long long r = it / c;
    const int p = (int)(r % hw), img = (int)(r / hw);
    const float v = adm_h_to_f32(dtok[((long long)img * tpad + 1 + p) * c + ch]) +
                    adm_h_to_f32(dtok[((long long)img * tpad) * c + ch]) / (float)hw;
    dact[it] = adm_f32_to_h(v);
  }
}

// backward-data weight image: conv with w'[ci][co][t'] = w[co][ci][taps-1-t'] (swap in/out, flip taps)
__global__ void __launch_bounds__(256)
pack_weight_bwd_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int cout, int cin, int taps, int ntiles16) {
  // the packed image describes a conv with cin' = cout input channels and cout' = cin outputs
  const long long total = (long long)(cout / 32) * taps * ntiles16 * 512;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), ln = (int)((i >> 3) & 63);
    long long r = i >> 9;
    const int tile = (int)(r % ntiles16); r /= ntiles16;
    const int tap = (int)(r % taps);
    const int chunk = (int)(r / taps);
    const int ch = tile * 16 + (ln & 15);          // output channel of the backward conv = ci
    const int k = chunk * 32 + (ln >> 4) * 8 + e;  // input channel of the backward conv = co
    float v = 0.0f;
    if (ch < cin) v = w[((long long)k * cin + ch) * taps + (taps - 1 - tap)];
    out[i] = adm_f32_to_h(v);
  }
}

int grid_for(long long items) {
  long long b = (items + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int adm_gn_bwd_partial(const adm_bf16* x, const adm_bf16* dy, const float* aff_a, const float* aff_b,
                                  float* partial, int n, int h, int w, int c, int slabs, int silu, int dy_half,
                                  void* stream) {
  ADM_REQUIRE(x && dy && aff_a && aff_b && partial, ADM_E_ARG, "adm_gn_bwd_partial: null pointer");
  ADM_REQUIRE(n > 0 && h > 0 && w > 0 && slabs > 0 && slabs <= h * w, ADM_E_ARG, "adm_gn_bwd_partial: bad shape");
  ADM_REQUIRE(c % 32 == 0 && c <= 2048, ADM_E_SHAPE, "adm_gn_bwd_partial: channels %d unsupported", c);
  ADM_REQUIRE(!dy_half || (h % 2 == 0 && w % 2 == 0), ADM_E_SHAPE, "adm_gn_bwd_partial: odd size with dy_half");
  ADM_REQUIRE(adm_aligned16(x) && adm_aligned16(dy), ADM_E_ALIGN, "adm_gn_bwd_partial: unaligned pointer");
  const int lanes = 256 / (c / 8);
  const size_t smem = (size_t)lanes * c * 2 * sizeof(float);
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(slabs, n), dim3(256), smem, (hipStream_t)stream, x, dy, aff_a, aff_b,
                     partial, h, w, c, slabs, silu, dy_half);
  return adm_check_launch("adm_gn_bwd_partial");
}

extern "C" int adm_gn_bwd_finalize(const float* partial, const float* aff_a, const float* stats, const float* add, int add_stride,
                                   float* k1, float* k0, int n, int c, int hw, int slabs, void* stream) {
  ADM_REQUIRE(partial && aff_a && stats && k1 && k0, ADM_E_ARG, "adm_gn_bwd_finalize: null pointer");
  ADM_REQUIRE(n > 0 && c % 32 == 0 && hw > 0 && slabs > 0, ADM_E_SHAPE, "adm_gn_bwd_finalize: bad shape");
  ADM_REQUIRE(!add || add_stride >= c, ADM_E_ARG, "adm_gn_bwd_finalize: add_stride < c");
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(n, 4), dim3(256), 0, (hipStream_t)stream, partial, aff_a, stats, add, add_stride,
                     k1, k0, c, hw, slabs);
  return adm_check_launch("adm_gn_bwd_finalize");
}

extern "C" int adm_gn_bwd_apply(const adm_bf16* x, const adm_bf16* dy, const float* aff_a, const float* aff_b,
                                const float* k1, const float* k0, const adm_bf16* add, adm_bf16* out, int n, int h,
                                int w, int c, int silu, int dy_half, int add_half, void* stream) {
  ADM_REQUIRE(x && dy && aff_a && aff_b && k1 && k0 && out, ADM_E_ARG, "adm_gn_bwd_apply: null pointer");
  ADM_REQUIRE(n > 0 && h > 0 && w > 0 && c % 8 == 0 && c <= 2048, ADM_E_SHAPE, "adm_gn_bwd_apply: bad shape");
  ADM_REQUIRE(!(dy_half || add_half) || (h % 2 == 0 && w % 2 == 0), ADM_E_SHAPE, "adm_gn_bwd_apply: odd size with a half-resolution operand");
  ADM_REQUIRE(adm_aligned16(x) && adm_aligned16(dy) && adm_aligned16(add) && adm_aligned16(out), ADM_E_ALIGN,
              "adm_gn_bwd_apply: unaligned pointer");
  const int hw = h * w;
  const int slabs = hw >= 64 ? hw / 64 : 1;   // 64 pixels per block, measured against 128 / 256 (the pass is elementwise: any split gives the same bits)
  ADM_REQUIRE(slabs <= 65535 && n <= 65535, ADM_E_SHAPE, "adm_gn_bwd_apply: grid too large");
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(slabs, n), dim3(256), 0,
                     (hipStream_t)stream, x, dy, aff_a, aff_b, k1, k0, add, out, h, w, c, slabs, silu, dy_half, add_half);
  return adm_check_launch("adm_gn_bwd_apply");
}

extern "C" int adm_grad_add(const adm_bf16* a, const adm_bf16* b, adm_bf16* out, int n, int h, int w, int c, int b_half,
                            void* stream) {
  ADM_REQUIRE(a && b && out, ADM_E_ARG, "adm_grad_add: null pointer");
  ADM_REQUIRE(n > 0 && h > 0 && w > 0 && c % 8 == 0, ADM_E_SHAPE, "adm_grad_add: bad shape");
  ADM_REQUIRE(adm_aligned16(a) && adm_aligned16(b) && adm_aligned16(out), ADM_E_ALIGN, "adm_grad_add: unaligned pointer");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for((long long)n * h * w * (c / 8))), dim3(256), 0, (hipStream_t)stream, a, b,
                     out, n, h, w, c, b_half);
  return adm_check_launch("adm_grad_add");
}

extern "C" int adm_logsoftmax_grad(const float* logits, const int64_t* y, float* dlogits, float* logp_sel, int n, int k,
                                   float scale, void* stream) {
  ADM_REQUIRE(logits && y && dlogits, ADM_E_ARG, "adm_logsoftmax_grad: null pointer");
  ADM_REQUIRE(n > 0 && k > 0, ADM_E_ARG, "adm_logsoftmax_grad: bad shape");
  hipLaunchKernelGGL(logsoftmax_grad_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, y, dlogits, logp_sel, k, scale);
  return adm_check_launch("adm_logsoftmax_grad");
}

extern "C" int adm_pool_prep(const adm_bf16* h, const float* aff_a, const float* aff_b, const float* pos, adm_bf16* tok,
                             int n, int hw, int c, int tpad, void* stream) {
  ADM_REQUIRE(h && aff_a && aff_b && pos && tok, ADM_E_ARG, "adm_pool_prep: null pointer");
  ADM_REQUIRE(n > 0 && hw > 0 && c > 0 && tpad >= hw + 1, ADM_E_ARG, "adm_pool_prep: bad shape");
  hipLaunchKernelGGL(pool_prep_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, h, aff_a, aff_b, pos, tok, hw, c, tpad);
  return adm_check_launch("adm_pool_prep");
}

extern "C" int adm_pool_attn_fwd(const adm_bf16* qkv, float* a0, float* wts, int n, int t, int tpad, int heads, int d,
                                 void* stream) {
  ADM_REQUIRE(qkv && a0 && wts, ADM_E_ARG, "adm_pool_attn_fwd: null pointer");
  ADM_REQUIRE(n > 0 && t > 0 && tpad >= t && heads > 0 && d > 0 && n < 65536, ADM_E_ARG, "adm_pool_attn_fwd: bad shape");
  hipLaunchKernelGGL(pool_attn_fwd_kernel, dim3(heads, n), dim3(64), (size_t)(d + tpad) * 4, (hipStream_t)stream, qkv, a0,
                     wts, t, tpad, heads, d);
  return adm_check_launch("adm_pool_attn_fwd");
}

extern "C" int adm_pool_attn_bwd(const adm_bf16* qkv, const float* wts, const float* da0, adm_bf16* dqkv, int n, int t,
                                 int tpad, int heads, int d, void* stream) {
  ADM_REQUIRE(qkv && wts && da0 && dqkv, ADM_E_ARG, "adm_pool_attn_bwd: null pointer");
  ADM_REQUIRE(n > 0 && t > 0 && tpad >= t && heads > 0 && d > 0 && d % 8 == 0 && n < 65536, ADM_E_ARG, "adm_pool_attn_bwd: bad shape");
  hipLaunchKernelGGL(pool_attn_bwd_kernel, dim3(heads, n), dim3(64), (size_t)(2 * d + tpad) * 4, (hipStream_t)stream, qkv,
                     wts, da0, dqkv, t, tpad, heads, d);
  return adm_check_launch("adm_pool_attn_bwd");
}

extern "C" int adm_pool_prep_bwd(const adm_bf16* dtok, adm_bf16* dact, int n, int hw, int c, int tpad, void* stream) {
  ADM_REQUIRE(dtok && dact, ADM_E_ARG, "adm_pool_prep_bwd: null pointer");
  ADM_REQUIRE(n > 0 && hw > 0 && c > 0 && tpad >= hw + 1, ADM_E_ARG, "adm_pool_prep_bwd: bad shape");
  hipLaunchKernelGGL(pool_prep_bwd_kernel, dim3(grid_for((long long)n * hw * c)), dim3(256), 0, (hipStream_t)stream, dtok,
                     dact, n, hw, c, tpad);
  return adm_check_launch("adm_pool_prep_bwd");
}

extern "C" int adm_pack_conv_weight_bwd(const float* w, adm_bf16* out, int cout, int cin, int taps, void* stream) {
  ADM_REQUIRE(w && out, ADM_E_ARG, "adm_pack_conv_weight_bwd: null pointer");
  ADM_REQUIRE(cout > 0 && cin > 0 && cout % 32 == 0 && (taps == 1 || taps == 9), ADM_E_SHAPE,
              "adm_pack_conv_weight_bwd: cout=%d cin=%d taps=%d unsupported (cout %% 32 == 0, taps 1|9)", cout, cin, taps);
  const int nt16 = (cin + 15) / 16;
  const long long total = (long long)(cout / 32) * taps * nt16 * 512;
  hipLaunchKernelGGL(pack_weight_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, out, cout, cin,
                     taps, nt16);
  return adm_check_launch("adm_pack_conv_weight_bwd");
}
