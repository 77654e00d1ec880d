// K5: QKVAttention / QKVAttentionLegacy as a flash-style MFMA kernel (gfx950).
//
// Replaces reference guided_diffusion/unet.py:361-393 (new order) and :328-358 (legacy order):
//   w = softmax_fp32((q*s) . (k*s)), s = ch^-1/4;  a = w . v
// without materialising the [T, T] weight matrix.  Input is the token-major output of the qkv 1x1
// projection, bf16 [N][T][3*H*D]; output bf16 [N][T][H*D].
//
// Formulation (all MFMAs are v_mfma_f32_16x16x32_bf16):
//   S^T[key][query] = K . Q^T        A = K rows from LDS, B = Q^T held in registers
//   online softmax over keys: keys live in the 4 accumulator registers and the 4 lane quarters of
//   a query column, so the row max/sum need two cross-lane shuffles (xor 16, 32) per tile
//   O^T[d][query] += V^T . P^T       B = P^T taken straight from the S^T accumulators (the k-order
//   of the contraction is permuted identically on the V^T side, so P never touches LDS);
//   A = V^T read from the row-major LDS tile with the transposing load ds_read_b64_tr_b16
// Block = 4 waves x 32 queries; K/V tiles of 64 keys shared through LDS.
#include "adm_attn_common.h"

namespace {

constexpr int KT = 64;        // keys per tile
constexpr int QW = 32;        // queries per wave
constexpr int QB = 128;       // queries per block
constexpr int PADE = 16;      // bf16 elements of row padding: row stride = D/2 + 8 dwords = 8 (mod 16), which keeps both the
                              // ds_read_b128 fragment reads and the ds_read_b64_tr_b16 transposing reads bank-conflict free

struct AttnK {
  const uint16_t* qkv; uint16_t* out; float* lse;
  const uint16_t* kv;      // keys / values: [N][kv_rows][Ckv], the first Tk rows of an image are attended to
  int T, heads, C3, C;     // T queries per image, row pitch C3 of the query tensor, C = heads * D output columns
  int Tk, kv_rows, Ckv;
  int q_off, k_off, v_off, head_stride, kv_head_stride;
  float scale_log2;  // log2(e) / sqrt(D)
};

// (256, 2): two waves per SIMD caps the kernel at 256 registers, which also makes hipcc keep the MFMA
// accumulators in VGPRs -- with the default bound it parked them in AGPRs and spent ~160 v_accvgpr_read/write
// per key tile moving S^T out for the softmax and O^T through the rescale.
template <int D>
__global__ void __launch_bounds__(256, D <= 64 ? 2 : 0)
attn_kernel(const AttnK p) {
  constexpr int KS = D / 32;   // 32-deep k-steps of QK^T
  constexpr bool R16 = (D % 32) == 16;  // plus one 16-deep step (v_mfma_f32_16x16x16_bf16): head widths 48, 80, 112
  static_assert(D % 16 == 0 && KS >= 1, "head width must be a multiple of 16, at least 32");
  constexpr int DT = D / 16;   // d tiles of the output
  // LDS row pitch = 8 mod 16 dwords (conflict-free for the b128 fragment reads and the transposing reads): 48- and
  // 80-wide heads have it unpadded (24 / 40 dwords), the multiples of 32 need the 16-element pad
  constexpr int KROW = D + (((D / 2) % 16 == 8) ? 0 : PADE);
  // K and V tiles row-major, double-buffered: the next tile's global loads fly during this tile's MFMAs
  __shared__ __attribute__((aligned(16))) uint16_t Ks[2][KT * KROW];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[2][KT * KROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int bx, by;
  adm_xcd_block(bx, by);
  const int n = by / p.heads, hd = by % p.heads;
  const int qbase = bx * QB + wave * QW;
  const uint16_t* base = p.qkv + (long long)n * p.T * p.C3;
  const int qcol = p.q_off + hd * p.head_stride, kcol = p.k_off + hd * p.kv_head_stride,
            vcol = p.v_off + hd * p.kv_head_stride;
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kv + (long long)n * p.kv_rows * p.Ckv), 0,
                                                                       p.Tk * p.Ckv * 2, 0x00020000);

  // one descriptor over this image's T rows: queries / keys beyond T read as zeros (no bounds branches)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.T * p.C3 * 2, 0x00020000);
  // Q^T fragments: lane (query lc, quarter lq) holds Q[query][ks*32 + 8*lq .. +8]
  bf16x8 qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = qbase + qt * 16 + lc;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const adm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * p.C3 + qcol + ks * 32 + lq * 8) * 2, 0, 0);
      qf[qt][ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  adm_s16x4 qf16[2] = {};  // the 16-deep tail: lane (query lc, quarter lq) holds Q[query][32*KS + 4*lq .. +4]
  if constexpr (R16) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = qbase + qt * 16 + lc;
      const adm_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (q * p.C3 + qcol + KS * 32 + lq * 4) * 2, 0, 0);
      qf16[qt] = __builtin_bit_cast(adm_s16x4, v);
    }
  }

  f32x4 oacc[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

  AdmTileRegs<KT, D, 256> kr, vr;
  kr.load_buf(rsk, p.Ckv, kcol, 0, tid);
  vr.load_buf(rsk, p.Ckv, vcol, 0, tid);
  kr.store(Ks[0], KROW, tid);
  vr.store(Vs[0], KROW, tid);
  __syncthreads();

  const int ntiles = (p.Tk + KT - 1) / KT;
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int k0 = kt0 * KT;
    const int cur = kt0 & 1;
    const bool next = kt0 + 1 < ntiles;
    if (next) {
      kr.load_buf(rsk, p.Ckv, kcol, k0 + KT, tid);
      vr.load_buf(rsk, p.Ckv, vcol, k0 + KT, tid);
    }
    const uint16_t* Kc = Ks[cur];
    const uint16_t* Vc = Vs[cur];

    // ---- S^T = K . Q^T  (4 key tiles x 2 query tiles).  All K fragments of the tile are requested first
    //      (scheduling fences keep hipcc from sinking each LDS read next to its two MFMAs, which exposed the
    //      read latency 8 times per tile); the V^T fragments are requested right after the S MFMAs so that
    //      their latency hides under the softmax VALU work.
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kfr[4][KS];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        kfr[kt][ks] = *reinterpret_cast<const bf16x8*>(&Kc[(kt * 16 + lc) * KROW + ks * 32 + lq * 8]);
    adm_s16x4 kfr16[4] = {};
    if constexpr (R16) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        kfr16[kt] = *reinterpret_cast<const adm_s16x4*>(&Kc[(kt * 16 + lc) * KROW + KS * 32 + lq * 4]);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 st[4][2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
        st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[kt][0], qf[qt][0], zero4, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < KS; ++ks)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[kt][ks], qf[qt][ks], st[kt][qt], 0, 0, 0);
      if constexpr (R16) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kfr16[kt], qf16[qt], st[kt][qt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 vfr[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) vfr[dt][kb] = adm_tr_frag(Vc, KROW, kb * 32, dt * 16, lc, lq);
    __builtin_amdgcn_sched_barrier(0);
    // ---- online softmax (per query column)
    const bool ragged = k0 + KT > p.Tk;
    bf16x8 pf[2][2];  // [query tile][32-key block]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = -1e30f;
      if (ragged) {  // only the last tile of a sequence whose length is not a multiple of 64
        asm volatile("" ::: "memory");  // keep this a (wave-uniform) branch: if-converted it is 32 compares + selects per tile
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + lq * 4 + r >= p.Tk) st[kt][qt][r] = -1e30f;
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][qt][r]);
      mx = adm_quarter_max(mx);
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * p.scale_log2);
      const float mneg = -m_new * p.scale_log2;
      m_run[qt] = m_new;
      // exponent arguments and the row sum in packed fp32 (v_pk_fma_f32 / v_pk_add_f32): this loop is VALU-bound
      const adm_f32x2 sc2 = {p.scale_log2, p.scale_log2}, mn2 = {mneg, mneg};
      adm_f32x2 ps2 = {0.f, 0.f};
      float pv[4][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const adm_f32x2 a = __builtin_elementwise_fma(adm_f32x2{st[kt][qt][2 * h], st[kt][qt][2 * h + 1]}, sc2, mn2);
          const adm_f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
          pv[kt][2 * h] = e.x;
          pv[kt][2 * h + 1] = e.y;
          ps2 += e;
        }
      const float psum = ps2.x + ps2.y;
      l_run[qt] = l_run[qt] * alpha + psum;
      if (__any(alpha != 1.0f)) {  // wave-uniform: skip the O rescale once the running max has settled
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[dt][qt] *= alpha;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (__bf16)pv[2 * kb + (e >> 2)][e & 3];
        pf[qt][kb] = f;
      }
    }
    // ---- O^T += V^T . P^T ; contraction slot k = 8*lq + e  <->  key kb*32 + 16*(e>>2) + 4*lq + (e&3);
    //      V^T fragments came from the row-major tile through the transposing LDS read
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt][kb], pf[qt][kb], oacc[dt][qt], 0, 0, 0);
    if (next) {
      kr.store(Ks[cur ^ 1], KROW, tid);
      vr.store(Vs[cur ^ 1], KROW, tid);
    }
    __syncthreads();
  }

  // ---- normalise and store: lane holds d = dt*16 + 4*lq .. +3 of query lc
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float l = adm_quarter_sum(l_run[qt]);
    const float inv = 1.0f / l;
    const int q = qbase + qt * 16 + lc;
    if (q >= p.T) continue;
    // log2-domain log-sum-exp of the scaled logits: P = exp2(s * scale_log2 - lse)
    if (p.lse && lq == 0) p.lse[((long long)n * p.heads + hd) * p.T + q] = m_run[qt] * p.scale_log2 + log2f(l);
    uint16_t* orow = p.out + ((long long)n * p.T + q) * p.C + hd * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x4 o = oacc[dt][qt] * inv;
      uint2 pk;
      pk.x = (uint32_t)adm_f32_to_bf16(o[0]) | ((uint32_t)adm_f32_to_bf16(o[1]) << 16);
      pk.y = (uint32_t)adm_f32_to_bf16(o[2]) | ((uint32_t)adm_f32_to_bf16(o[3]) << 16);
      *reinterpret_cast<uint2*>(orow + dt * 16 + lq * 4) = pk;
    }
  }
}

// ---- wide heads (D = 192, 256: ADM-128's num_heads = 4 gives 128 / 192 / 256 channels per head).  Same
// formulation, sized for the register file instead of for speed: 16 queries per wave (64 per block), 32-key tiles
// single-buffered in LDS, K / V^T fragments streamed one tile at a time.  Not on the benchmarked path.
template <int D>
__global__ void __launch_bounds__(256)
attn_wide_kernel(const AttnK p) {
  constexpr int KT2 = 32, QB2 = 64;
  constexpr int KS = D / 32, DT = D / 16, KROW = D + PADE;
  __shared__ __attribute__((aligned(16))) uint16_t Ks[KT2 * KROW];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[KT2 * KROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int bx, by;
  adm_xcd_block(bx, by);
  const int n = by / p.heads, hd = by % p.heads;
  const int q = bx * QB2 + wave * 16 + lc;
  const uint16_t* base = p.qkv + (long long)n * p.T * p.C3;
  const int qcol = p.q_off + hd * p.head_stride, kcol = p.k_off + hd * p.kv_head_stride,
            vcol = p.v_off + hd * p.kv_head_stride;
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kv + (long long)n * p.kv_rows * p.Ckv), 0,
                                                                       p.Tk * p.Ckv * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, p.T * p.C3 * 2, 0x00020000);

  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const adm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (q * p.C3 + qcol + ks * 32 + lq * 8) * 2, 0, 0);
    qf[ks] = __builtin_bit_cast(bf16x8, v);
  }
  f32x4 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  AdmTileRegs<KT2, D, 256> kr, vr;
  const int ntiles = (p.Tk + KT2 - 1) / KT2;
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int k0 = kt0 * KT2;
    kr.load_buf(rsk, p.Ckv, kcol, k0, tid);
    vr.load_buf(rsk, p.Ckv, vcol, k0, tid);
    __syncthreads();  // the previous tile's readers are done
    kr.store(Ks, KROW, tid);
    vr.store(Vs, KROW, tid);
    __syncthreads();
    // S^T = K . Q^T: 2 key tiles x 1 query tile
    f32x4 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(kt * 16 + lc) * KROW + ks * 32 + lq * 8]);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[kt], 0, 0, 0);
      }
    }
    if (k0 + KT2 > p.Tk) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (k0 + kt * 16 + lq * 4 + r >= p.Tk) st[kt][r] = -1e30f;
    }
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = adm_quarter_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
    const float mneg = -m_new * p.scale_log2;
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pf;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[kt][r] * p.scale_log2 + mneg);
        psum += e;
        pf[kt * 4 + r] = (__bf16)e;
      }
    l_run = l_run * alpha + psum;
    // O^T += V^T . P^T; contraction slot k = 8*lq + e <-> key 16*(e>>2) + 4*lq + (e&3)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const bf16x8 vf = adm_tr_frag(Vs, KROW, 0, dt * 16, lc, lq);
      oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt] * alpha, 0, 0, 0);
    }
  }
  const float l = adm_quarter_sum(l_run);
  const float inv = 1.0f / l;
  if (q >= p.T) return;
  if (p.lse && lq == 0) p.lse[((long long)n * p.heads + hd) * p.T + q] = m_run * p.scale_log2 + log2f(l);
  uint16_t* orow = p.out + ((long long)n * p.T + q) * p.C + hd * D;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const f32x4 o = oacc[dt] * inv;
    uint2 pk;
    pk.x = (uint32_t)adm_f32_to_bf16(o[0]) | ((uint32_t)adm_f32_to_bf16(o[1]) << 16);
    pk.y = (uint32_t)adm_f32_to_bf16(o[2]) | ((uint32_t)adm_f32_to_bf16(o[3]) << 16);
    *reinterpret_cast<uint2*>(orow + dt * 16 + lq * 4) = pk;
  }
}

int launch_attention(const AttnK& k, int n, int t, int heads, int d, hipStream_t s);

}  // namespace

extern "C" int adm_attention_lse(const adm_bf16* qkv, adm_bf16* out, float* lse, int n, int t, int heads, int d,
                                 int new_order, void* stream);

extern "C" int adm_attention(const adm_bf16* qkv, adm_bf16* out, int n, int t, int heads, int d, int new_order,
                             void* stream) {
  return adm_attention_lse(qkv, out, nullptr, n, t, heads, d, new_order, stream);
}

extern "C" int adm_attention_lse(const adm_bf16* qkv, adm_bf16* out, float* lse, int n, int t, int heads, int d,
                                 int new_order, void* stream) {
  ADM_REQUIRE(qkv && out, ADM_E_ARG, "adm_attention: null pointer");
  ADM_REQUIRE(n > 0 && t > 0 && heads > 0, ADM_E_ARG, "adm_attention: bad shape n=%d t=%d heads=%d", n, t, heads);
  ADM_REQUIRE(d == 32 || d == 48 || d == 64 || d == 80 || d == 96 || d == 128 || d == 160 || d == 192 || d == 256, ADM_E_SHAPE,
              "adm_attention: head dim %d unsupported (32, 48, 64, 80, 96, 128, 160, 192, 256)", d);
  ADM_REQUIRE(adm_aligned16(qkv) && adm_aligned16(out), ADM_E_ALIGN, "adm_attention: unaligned pointer");
  ADM_REQUIRE((long long)n * heads < 65536, ADM_E_SHAPE, "adm_attention: n*heads exceeds grid.y");
  AttnK k{};
  k.qkv = qkv; k.out = out; k.lse = lse; k.T = t; k.heads = heads;
  k.C = heads * d; k.C3 = 3 * k.C;
  if (new_order) { k.q_off = 0; k.k_off = k.C; k.v_off = 2 * k.C; k.head_stride = d; }
  else           { k.q_off = 0; k.k_off = d;   k.v_off = 2 * d;   k.head_stride = 3 * d; }
  k.kv = qkv; k.Tk = t; k.kv_rows = t; k.Ckv = k.C3; k.kv_head_stride = k.head_stride;
  k.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
  return launch_attention(k, n, t, heads, d, (hipStream_t)stream);
}

extern "C" int adm_attention_cross(const adm_bf16* q, int q_stride, const adm_bf16* kv, int kv_stride, int kv_rows,
                                   adm_bf16* out, int n, int tq, int tk, int heads, int d, float scale, void* stream) {
  ADM_REQUIRE(q && kv && out, ADM_E_ARG, "adm_attention_cross: null pointer");
  ADM_REQUIRE(n > 0 && tq > 0 && tk > 0 && heads > 0 && kv_rows >= tk, ADM_E_ARG,
              "adm_attention_cross: bad shape n=%d tq=%d tk=%d kv_rows=%d heads=%d", n, tq, tk, kv_rows, heads);
  ADM_REQUIRE(d == 32 || d == 48 || d == 64 || d == 80 || d == 96 || d == 128 || d == 160 || d == 192 || d == 256, ADM_E_SHAPE,
              "adm_attention_cross: head dim %d unsupported (32, 48, 64, 80, 96, 128, 160, 192, 256)", d);
  ADM_REQUIRE(q_stride >= heads * d && kv_stride >= 2 * heads * d && q_stride % 8 == 0 && kv_stride % 8 == 0, ADM_E_SHAPE,
              "adm_attention_cross: row pitches %d / %d too small or not multiples of 8", q_stride, kv_stride);
  ADM_REQUIRE(adm_aligned16(q) && adm_aligned16(kv) && adm_aligned16(out), ADM_E_ALIGN, "adm_attention_cross: unaligned pointer");
  ADM_REQUIRE((long long)n * heads < 65536, ADM_E_SHAPE, "adm_attention_cross: n*heads exceeds grid.y");
  ADM_REQUIRE((long long)tq * q_stride < (1ll << 30) && (long long)tk * kv_stride < (1ll << 30), ADM_E_SHAPE,
              "adm_attention_cross: an image's rows exceed the 32-bit byte offsets of a buffer descriptor");
  AttnK k{};
  k.qkv = q; k.out = out; k.lse = nullptr; k.T = tq; k.heads = heads;
  k.C = heads * d; k.C3 = q_stride;
  k.q_off = 0; k.head_stride = d;
  k.kv = kv; k.Tk = tk; k.kv_rows = kv_rows; k.Ckv = kv_stride;
  k.k_off = 0; k.v_off = heads * d; k.kv_head_stride = d;
  k.scale_log2 = 1.4426950408889634f * (scale > 0.f ? scale : 1.0f / sqrtf((float)d));
  return launch_attention(k, n, tq, heads, d, (hipStream_t)stream);
}

namespace {
int launch_attention(const AttnK& k, int n, int t, int heads, int d, hipStream_t s) {
  dim3 grid((t + QB - 1) / QB, n * heads);
  if (d > 128) {
    dim3 gridw((t + 63) / 64, n * heads);
    if (d == 160) hipLaunchKernelGGL((attn_wide_kernel<160>), gridw, dim3(256), 0, s, k);
    else if (d == 192) hipLaunchKernelGGL((attn_wide_kernel<192>), gridw, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((attn_wide_kernel<256>), gridw, dim3(256), 0, s, k);
    return adm_check_launch("adm_attention");
  }
  if (d == 32) hipLaunchKernelGGL((attn_kernel<32>), grid, dim3(256), 0, s, k);
  else if (d == 48) hipLaunchKernelGGL((attn_kernel<48>), grid, dim3(256), 0, s, k);
  else if (d == 80) hipLaunchKernelGGL((attn_kernel<80>), grid, dim3(256), 0, s, k);
  else if (d == 64) hipLaunchKernelGGL((attn_kernel<64>), grid, dim3(256), 0, s, k);
  else if (d == 96) hipLaunchKernelGGL((attn_kernel<96>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((attn_kernel<128>), grid, dim3(256), 0, s, k);
  return adm_check_launch("adm_attention");
}
}  // namespace
