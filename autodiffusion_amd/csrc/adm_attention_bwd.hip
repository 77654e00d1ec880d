// K10 (attention part): backward-data of QKVAttention for the classifier-guidance gradient.
//
// The reference obtains d(log p(y|x))/dx with torch.autograd through EncoderUNetModel
// (search_imagenet64_classifier_guidance.py:319-326); its AttentionBlocks re-run their forward
// inside backward (hard-wired checkpoint, guided_diffusion/unet.py:297).  Here the attention
// backward is a flash-style recomputation from (q, k, v, forward output, log-sum-exp):
//   delta[q]   = sum_d dA[q,d] * A[q,d]
//   P          = exp2(S * c - lse),  S = Q K^T,  c = log2(e)/sqrt(D)
//   dV = P^T dA;   dP = dA V^T;   dS = P o (dP - delta);   dQ = dS K / sqrt(D);   dK = dS^T Q / sqrt(D)
// Two kernels, no atomics (bitwise reproducible): dQ per 128-query block (loop over key tiles; it also
// computes delta for its queries and leaves it in the workspace), and
// dK/dV per 128-key block (loop over query tiles).  All MFMAs are v_mfma_f32_16x16x32_bf16; as in the
// forward kernel the second product of each chain takes its B operand straight from the first
// product's accumulators (k-order permuted identically on the LDS-transposed A side).
#include "adm_attn_common.h"

namespace {

constexpr int TT = 64;        // streamed tile (keys for dQ, queries for dK/dV)
constexpr int BW = 32;        // rows (queries / keys) owned per wave
constexpr int BB = 128;       // rows per block
constexpr int PADE = 16;      // see adm_attention.hip: conflict-free row stride

struct AttnBwdK {
  const uint16_t* qkv; const uint16_t* out; const uint16_t* dout; const float* lse; const float* delta;
  float* delta_out;  // written by the dQ kernel, read (as `delta`) by the dK/dV kernel
  uint16_t* dqkv;
  int T, heads, C3, C;
  int q_off, k_off, v_off, head_stride;
  float scale_log2, inv_sqrt_d;
};

// ------------------------------------------------------------------------------------ dQ
template <int D>
__global__ void __launch_bounds__(256, 2)
attn_dq_kernel(const AttnBwdK p) {
  constexpr int KS = D / 32, DT = D / 16, KROW = D + PADE;
  __shared__ __attribute__((aligned(16))) uint16_t Ks[2][TT * KROW];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[2][TT * KROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lq = lane >> 4;
  int bx, by;
  adm_xcd_block(bx, by);
  const int n = by / p.heads, hd = by % p.heads;
  const int qbase = bx * BB + wave * BW;
  const uint16_t* base = p.qkv + (long long)n * p.T * p.C3;
  const uint16_t* dbase = p.dout + (long long)n * p.T * p.C;
  const int qcol = p.q_off + hd * p.head_stride, kcol = p.k_off + hd * p.head_stride, vcol = p.v_off + hd * p.head_stride;

  // descriptors over this image's T rows: rows beyond T read as zeros, so no load carries a bounds branch and no
  // masking is needed -- a key beyond T has K = V = 0, hence dP = 0 and its dS row multiplies a zero K row
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.T * p.C3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, p.T * p.C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (long long)n * p.T * p.C), 0, p.T * p.C * 2, 0x00020000);
  adm_h8 qf[2][KS], gf[2][KS];
  float lse[2], dl[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qbase + qt * 16 + lc;
    const bool ok = q < p.T;
    // delta[q] = sum_d dA[q,d] * A[q,d], from the same fragment shape as dA (lane: 8 d-values per k-step), reduced over
    // the 4 lane quarters; written out for the dK/dV kernel that follows on the stream
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const adm_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * p.C3 + qcol + ks * 32 + lq * 8) * 2, 0, 0);
      const adm_u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rsg, (q * p.C + hd * D + ks * 32 + lq * 8) * 2, 0, 0);
      const adm_u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rso, (q * p.C + hd * D + ks * 32 + lq * 8) * 2, 0, 0);
      qf[qt][ks] = __builtin_bit_cast(adm_h8, a);
      gf[qt][ks] = __builtin_bit_cast(adm_h8, g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dsum += adm_lo_f32(o[e]) * adm_lo_f32(g[e]);
        dsum += adm_hi_f32(o[e]) * adm_hi_f32(g[e]);
      }
    }
    dsum = adm_quarter_sum(dsum);
    const long long si = ((long long)n * p.heads + hd) * p.T + (ok ? q : 0);
    lse[qt] = ok ? p.lse[si] : 0.f;
    dl[qt] = dsum;
    if (ok && lq == 0) p.delta_out[si] = dsum;
  }
  f32x4 acc[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  AdmTileRegs<TT, D, 256> kr, vr;
  kr.load_buf(rs, p.C3, kcol, 0, tid);
  vr.load_buf(rs, p.C3, vcol, 0, tid);
  kr.store(Ks[0], KROW, tid);
  vr.store(Vs[0], KROW, tid);
  __syncthreads();

  const int ntiles = (p.T + TT - 1) / TT;
  for (int t0 = 0; t0 < ntiles; ++t0) {
    const int k0 = t0 * TT, cur = t0 & 1;
    const bool next = t0 + 1 < ntiles;
    if (next) {
      kr.load_buf(rs, p.C3, kcol, k0 + TT, tid);
      vr.load_buf(rs, p.C3, vcol, k0 + TT, tid);
    }
    const uint16_t* Kc = Ks[cur];
    const uint16_t* Vc = Vs[cur];
    adm_h8 dsf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 st[2][2], dp[2][2];  // [key tile within the 32-key block][query tile]
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int kt = 2 * kb + kk;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const adm_h8 kf = *reinterpret_cast<const adm_h8*>(&Kc[(kt * 16 + lc) * KROW + ks * 32 + lq * 8]);
          const adm_h8 vf = *reinterpret_cast<const adm_h8*>(&Vc[(kt * 16 + lc) * KROW + ks * 32 + lq * 8]);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            st[kk][qt] = adm_mfma_16x16x32(kf, qf[qt][ks], ks == 0 ? zero4 : st[kk][qt], 0, 0, 0);
            dp[kk][qt] = adm_mfma_16x16x32(vf, gf[qt][ks], ks == 0 ? zero4 : dp[kk][qt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // P = exp2(S c - lse), dS = P (dP - delta): packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32)
        const adm_f32x2 sc2 = {p.scale_log2, p.scale_log2}, nl2 = {-lse[qt], -lse[qt]}, nd2 = {-dl[qt], -dl[qt]};
        float dsv[8];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const adm_f32x2 a = __builtin_elementwise_fma(adm_f32x2{st[kk][qt][2 * h], st[kk][qt][2 * h + 1]}, sc2, nl2);
            const adm_f32x2 pr = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
            const adm_f32x2 d2 = pr * (adm_f32x2{dp[kk][qt][2 * h], dp[kk][qt][2 * h + 1]} + nd2);
            dsv[kk * 4 + 2 * h] = d2.x;
            dsv[kk * 4 + 2 * h + 1] = d2.y;
          }
        adm_h8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (adm_elem_t)dsv[e];
        dsf[qt][kb] = f;
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const adm_h8 kt_f = adm_tr_frag(Kc, KROW, kb * 32, dt * 16, lc, lq);  // K^T from the row-major tile
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          acc[dt][qt] = adm_mfma_16x16x32(kt_f, dsf[qt][kb], acc[dt][qt], 0, 0, 0);
      }
    if (next) {
      kr.store(Ks[cur ^ 1], KROW, tid);
      vr.store(Vs[cur ^ 1], KROW, tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qbase + qt * 16 + lc;
    if (q >= p.T) continue;
    uint16_t* orow = p.dqkv + ((long long)n * p.T + q) * p.C3 + qcol;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x4 o = acc[dt][qt] * p.inv_sqrt_d;
      uint2 pk;
      pk.x = adm_pack2(o[0], o[1]);
      pk.y = adm_pack2(o[2], o[3]);
      *reinterpret_cast<uint2*>(orow + dt * 16 + lq * 4) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------ dK, dV
template <int D>
__global__ void __launch_bounds__(256, 2)
attn_dkv_kernel(const AttnBwdK p) {
  constexpr int KS = D / 32, DT = D / 16, KROW = D + PADE;
  __shared__ __attribute__((aligned(16))) uint16_t Qs[2][TT * KROW];
  __shared__ __attribute__((aligned(16))) uint16_t Gs[2][TT * KROW];
  __shared__ __attribute__((aligned(16))) float lse_s[2][TT], dl_s[2][TT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lq = lane >> 4;
  int bx, by;
  adm_xcd_block(bx, by);
  const int n = by / p.heads, hd = by % p.heads;
  const int kbase = bx * BB + wave * BW;
  const uint16_t* base = p.qkv + (long long)n * p.T * p.C3;
  const uint16_t* dbase = p.dout + (long long)n * p.T * p.C;
  const int qcol = p.q_off + hd * p.head_stride, kcol = p.k_off + hd * p.head_stride, vcol = p.v_off + hd * p.head_stride;
  const long long sbase = ((long long)n * p.heads + hd) * p.T;

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.T * p.C3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, p.T * p.C * 2, 0x00020000);
  // K^T / V^T B-operand fragments of this wave's 32 keys: lane (key lc, quarter lq) holds row[key][ks*32 + 8*lq ..]
  adm_h8 kf[2][KS], vf[2][KS];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = kbase + kt * 16 + lc;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const adm_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (key * p.C3 + kcol + ks * 32 + lq * 8) * 2, 0, 0);
      const adm_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (key * p.C3 + vcol + ks * 32 + lq * 8) * 2, 0, 0);
      kf[kt][ks] = __builtin_bit_cast(adm_h8, a);
      vf[kt][ks] = __builtin_bit_cast(adm_h8, b);
    }
  }
  f32x4 dv[DT][2], dk[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) { dv[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  AdmTileRegs<TT, D, 256> qr, gr;
  float lse_r = 0.f, dl_r = 0.f;
  auto load_rows = [&](int q0) {
    qr.load_buf(rs, p.C3, qcol, q0, tid);
    gr.load_buf(rsg, p.C, hd * D, q0, tid);
    if (tid < TT) {
      // a query beyond T gets lse = +huge: its P = exp2(S c - lse) is exactly 0, no masking in the loop
      const int q = q0 + tid;
      lse_r = q < p.T ? p.lse[sbase + q] : 1e30f;
      dl_r = q < p.T ? p.delta[sbase + q] : 0.f;
    }
  };
  auto store_rows = [&](int buf) {
    qr.store(Qs[buf], KROW, tid);
    gr.store(Gs[buf], KROW, tid);
    if (tid < TT) { lse_s[buf][tid] = lse_r; dl_s[buf][tid] = dl_r; }
  };
  load_rows(0);
  store_rows(0);
  __syncthreads();

  const int ntiles = (p.T + TT - 1) / TT;
  for (int t0 = 0; t0 < ntiles; ++t0) {
    const int q0 = t0 * TT, cur = t0 & 1;
    const bool next = t0 + 1 < ntiles;
    if (next) load_rows(q0 + TT);
    const uint16_t* Qc = Qs[cur];
    const uint16_t* Gc = Gs[cur];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {     // 32-query block of the tile
      adm_h8 pf[2], dsf[2];             // per key tile
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x4 st[2], dp[2];              // the block's two 16-query tiles
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int qt = 2 * qb + qq;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const adm_h8 qa = *reinterpret_cast<const adm_h8*>(&Qc[(qt * 16 + lc) * KROW + ks * 32 + lq * 8]);
            const adm_h8 ga = *reinterpret_cast<const adm_h8*>(&Gc[(qt * 16 + lc) * KROW + ks * 32 + lq * 8]);
            st[qq] = adm_mfma_16x16x32(qa, kf[kt][ks], ks == 0 ? zero4 : st[qq], 0, 0, 0);
            dp[qq] = adm_mfma_16x16x32(ga, vf[kt][ks], ks == 0 ? zero4 : dp[qq], 0, 0, 0);
          }
        }
        // the 4 query rows of a lane quarter are consecutive: one 16-byte LDS read each for lse and delta
        adm_h8 f, g;
        const adm_f32x2 sc2 = {p.scale_log2, p.scale_log2};
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int ql = (2 * qb + qq) * 16 + lq * 4;  // first query row of this lane within the tile
          const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[cur][ql]);
          const float4 d4 = *reinterpret_cast<const float4*>(&dl_s[cur][ql]);
          const adm_f32x2 a0 = __builtin_elementwise_fma(adm_f32x2{st[qq][0], st[qq][1]}, sc2, adm_f32x2{-l4.x, -l4.y});
          const adm_f32x2 a1 = __builtin_elementwise_fma(adm_f32x2{st[qq][2], st[qq][3]}, sc2, adm_f32x2{-l4.z, -l4.w});
          const adm_f32x2 p0 = {__builtin_amdgcn_exp2f(a0.x), __builtin_amdgcn_exp2f(a0.y)};
          const adm_f32x2 p1 = {__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
          const adm_f32x2 s0 = p0 * (adm_f32x2{dp[qq][0], dp[qq][1]} - adm_f32x2{d4.x, d4.y});
          const adm_f32x2 s1 = p1 * (adm_f32x2{dp[qq][2], dp[qq][3]} - adm_f32x2{d4.z, d4.w});
          f[qq * 4 + 0] = (adm_elem_t)p0.x; f[qq * 4 + 1] = (adm_elem_t)p0.y; f[qq * 4 + 2] = (adm_elem_t)p1.x; f[qq * 4 + 3] = (adm_elem_t)p1.y;
          g[qq * 4 + 0] = (adm_elem_t)s0.x; g[qq * 4 + 1] = (adm_elem_t)s0.y; g[qq * 4 + 2] = (adm_elem_t)s1.x; g[qq * 4 + 3] = (adm_elem_t)s1.y;
        }
        pf[kt] = f;
        dsf[kt] = g;
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const adm_h8 gt_f = adm_tr_frag(Gc, KROW, qb * 32, dt * 16, lc, lq);  // dA^T
        const adm_h8 qt_f = adm_tr_frag(Qc, KROW, qb * 32, dt * 16, lc, lq);  // Q^T
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          dv[dt][kt] = adm_mfma_16x16x32(gt_f, pf[kt], dv[dt][kt], 0, 0, 0);
          dk[dt][kt] = adm_mfma_16x16x32(qt_f, dsf[kt], dk[dt][kt], 0, 0, 0);
        }
      }
    }
    if (next) store_rows(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = kbase + kt * 16 + lc;
    if (key >= p.T) continue;
    uint16_t* orow = p.dqkv + ((long long)n * p.T + key) * p.C3;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x4 a = dk[dt][kt] * p.inv_sqrt_d;
      const f32x4 b = dv[dt][kt];
      uint2 pk;
      pk.x = adm_pack2(a[0], a[1]);
      pk.y = adm_pack2(a[2], a[3]);
      *reinterpret_cast<uint2*>(orow + kcol + dt * 16 + lq * 4) = pk;
      pk.x = adm_pack2(b[0], b[1]);
      pk.y = adm_pack2(b[2], b[3]);
      *reinterpret_cast<uint2*>(orow + vcol + dt * 16 + lq * 4) = pk;
    }
  }
}

}  // namespace

extern "C" int adm_attention_bwd(const adm_bf16* qkv, const adm_bf16* out, const adm_bf16* dout, const float* lse,
                                 float* delta_ws, adm_bf16* dqkv, int n, int t, int heads, int d, int new_order,
                                 void* stream) {
  ADM_REQUIRE(qkv && out && dout && lse && delta_ws && dqkv, ADM_E_ARG, "adm_attention_bwd: null pointer");
  ADM_REQUIRE(n > 0 && t > 0 && heads > 0, ADM_E_ARG, "adm_attention_bwd: bad shape");
  ADM_REQUIRE(d == 32 || d == 64, ADM_E_SHAPE, "adm_attention_bwd: head dim %d unsupported (32, 64)", d);
  ADM_REQUIRE(adm_aligned16(qkv) && adm_aligned16(out) && adm_aligned16(dout) && adm_aligned16(dqkv), ADM_E_ALIGN,
              "adm_attention_bwd: unaligned pointer");
  ADM_REQUIRE((long long)n * heads < 65536, ADM_E_SHAPE, "adm_attention_bwd: n*heads exceeds grid.y");
  AttnBwdK k{};
  k.qkv = qkv; k.out = out; k.dout = dout; k.lse = lse; k.delta = delta_ws; k.delta_out = delta_ws; k.dqkv = dqkv;
  k.T = t; k.heads = heads; k.C = heads * d; k.C3 = 3 * k.C;
  if (new_order) { k.q_off = 0; k.k_off = k.C; k.v_off = 2 * k.C; k.head_stride = d; }
  else           { k.q_off = 0; k.k_off = d;   k.v_off = 2 * d;   k.head_stride = 3 * d; }
  k.inv_sqrt_d = 1.0f / sqrtf((float)d);
  k.scale_log2 = 1.4426950408889634f * k.inv_sqrt_d;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((t + BB - 1) / BB, n * heads);
  if (d == 32) {
    hipLaunchKernelGGL((attn_dq_kernel<32>), grid, dim3(256), 0, s, k);
    hipLaunchKernelGGL((attn_dkv_kernel<32>), grid, dim3(256), 0, s, k);
  } else {
    hipLaunchKernelGGL((attn_dq_kernel<64>), grid, dim3(256), 0, s, k);
    hipLaunchKernelGGL((attn_dkv_kernel<64>), grid, dim3(256), 0, s, k);
  }
  return adm_check_launch("adm_attention_bwd");
}
