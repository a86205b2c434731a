// gw_bf16.hip - bf16-MFMA variant of the fused MLP kernels (BASELINE.json configs[2]: "bf16 MFMA node-MLPs").
//
// Same transposed, register-resident scheme as the fp32 kernels (gw_kernels.hip / gw_edge.hip):
//       H_out[feature][column] = W[feature][k] . H_in[k][column],
// but on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): weights are the A operand (lane: row = lane & 15, 8 consecutive
// packed k's selected by lane >> 4), activations the B operand (column = lane & 15), converted fp32 -> bf16 (RNE) in
// registers.  With the K order
//       k(s, q, i) = 32 s + 16 (i >> 2) + 4 q + (i & 3)        (s = K-step, q = lane >> 4, i = 0..7)
// the 8 values a lane must supply for K-step s of the next layer are exactly its accumulator entries of row tiles
// 2s and 2s+1, so the chain of layers + LayerNorm + residual stays in registers as in fp32.
// Everything that is not a matrix product (bias, LayerNorm statistics, residual adds, segment sums, and every tensor
// in HBM) stays fp32.
//
// bf16 matrix cores are 16x faster than the fp32 ones, so the balance moves to the weight stream: one A fragment
// (1 KiB per wave) read from LDS feeds 4 MFMAs here - every wave works on FOUR 16-column groups (64 columns, 256 per
// workgroup) - which keeps LDS reads at a quarter of their peak and the L2->LDS weight DMA at 0.5 KiB per column
// and layer.  That needs ~420 VGPRs per wave: one workgroup (4 waves) per CU.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

#ifdef GW_TUNING
#define GW_TUNE16(a) ((a).tune16)
#else
#define GW_TUNE16(a) 0
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// A workgroup is NW waves x NG 16-column groups per wave.  4 x 2 (128 columns, one wave per SIMD, ~450 registers: every A
// fragment read from LDS feeds two MFMAs) is the form of the general kernels; the row-wise launches of the fused forward are
// bound by their row loads / stores, which one lock-stepped workgroup per CU cannot overlap with anything, and run as 4 x 1
// (64 columns, <= 256 registers, TWO workgroups per CU: twice the weight DMA and LDS reads per column, but one workgroup's
// row traffic under the other's MFMA phases - measured 14-30 % faster; 8 x 1 in one workgroup sits between).
constexpr int kBufBytes = 32768;            // one weight chunk buffer (2 K-steps x 16 tiles x 1 KiB)
constexpr int kStageLd16 = 260;
constexpr int kStageFloats16 = 64 * kStageLd16;
constexpr int kLdsWeights = 2 * kBufBytes;  // double buffered
constexpr int kLdsEdge = kLdsWeights + (kStageFloats16 + 64) * 4;

template <int N>
__device__ __forceinline__ void wait_vm16() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier16() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DMA `bytes` (multiple of 1 KiB) of the packed weight stream into LDS at byte offset lds_off; pieces round-robin
// over the 4 waves.
template <int NW>
__device__ __forceinline__ void issue_bytes(const char* __restrict__ g, int bytes, unsigned lds_off, int lane, int wave) {
  const int npieces = bytes >> 10;
  for (int p = wave; p < npieces; p += NW)
    glds16_asm_s((const float*)(g + (size_t)p * 1024), (unsigned)lane * 16u,
                 __builtin_amdgcn_readfirstlane(lds_off + (unsigned)p * 1024u));
}

__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
  bf16x8 r;
  r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
  r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
  return r;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}
// layer-1 node products as fp16 rows (GW_LAYOUT_ROWS_F16): 11 significant bits in front of the consumer's bf16 rounding of
// the layer-1 activations; clamped to the fp16 range (an overflow would turn into inf, then NaN under LayerNorm)
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stg_half4(void* p, f32x4 v) {
  half4_t h;
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = (_Float16)fminf(fmaxf(v[r], -65504.f), 65504.f);
  *(GW_AS1 half4_t*)p = h;
}

// One layer pass: acc[g][t] += W[16t.., k] . bin[g][k], K = 32 KS, NT row tiles, NTP = tiles per K-step in the packed
// stream (NT rounded up to 4).  The stream of this pass starts at gw; its first chunk has already been issued into
// buffer `parity`; while the last chunk computes, the first chunk of the next pass (next_gw, next_bytes) is issued.
template <int NW, int NG, int KS, int BKS, int NT, int NTP>
__device__ __forceinline__ void pass16(f32x4 (&acc)[NG][NT], const bf16x8 (&bin)[NG][BKS], const char* __restrict__ gw,
                                       const char* __restrict__ next_gw, int next_bytes, const char* lds, int& parity,
                                       int lane, int wave, int tune = 0) {
  constexpr int STEP_BYTES = NTP * 1024;
  constexpr int CS = (2 * STEP_BYTES <= kBufBytes) ? 2 : 1;  // K-steps per chunk
  constexpr int NCH = (KS + CS - 1) / CS;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    constexpr int dummy = 0;
    (void)dummy;
    const int steps_c = (KS - c * CS) < CS ? (KS - c * CS) : CS;
    wait_vm16<0>();
    if (!(tune & 4)) lds_barrier16();  // chunk c has landed for every wave; nobody still reads the other buffer
    if (!(tune & 1)) {
      if (c + 1 < NCH) {
        const int sn = (KS - (c + 1) * CS) < CS ? (KS - (c + 1) * CS) : CS;
        issue_bytes<NW>(gw + (size_t)(c + 1) * CS * STEP_BYTES, sn * STEP_BYTES, (unsigned)((parity ^ 1) * kBufBytes), lane, wave);
      } else if (next_gw != nullptr) {
        issue_bytes<NW>(next_gw, next_bytes, (unsigned)((parity ^ 1) * kBufBytes), lane, wave);
      }
    }
    if (tune & 2) {
      parity ^= 1;
      continue;
    }
    const char* buf = lds + parity * kBufBytes + lane * 16;
    // The A fragments (4 row tiles = one "unit" of 4 ds_read_b128) travel through a ring of three register sets, requested two
    // units ahead of the MFMAs that use them: this kernel runs ONE wave per SIMD, so an LDS round trip that is waited for right
    // after its request (what the compiler schedules on its own: 2 reads, wait, 2 MFMAs) is exposed in full (measured: 6 % of
    // the row-wise launches; the rest of their per-chunk time is not the weight stream either - neither two chunks in flight
    // nor a per-workgroup rotation of the piece order moved it, profiles/r03 notes in DESIGN.md).
    constexpr int UPS = NTP / 4;           // units per K-step
    const int NU = steps_c * UPS;          // units of this chunk (<= 2 * UPS)
    bf16x8 af[3][4];
    auto ldu = [&](bf16x8 (&f)[4], int u) {
      const int su = u / UPS, t4 = u - su * UPS;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) f[tt] = *(const bf16x8*)(buf + su * STEP_BYTES + (t4 * 4 + tt) * 1024);
    };
    ldu(af[0], 0);
    if (NU > 1) ldu(af[1], 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < CS * UPS; ++u) {
      if (u < NU) {
        if (u + 2 < NU) ldu(af[(u + 2) % 3], u + 2);
        const int su = u / UPS, t4 = u - su * UPS;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          if (t4 * 4 + tt < NT) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
              acc[g][t4 * 4 + tt] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u % 3][tt], bin[g][c * CS + su], acc[g][t4 * 4 + tt], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    parity ^= 1;
  }
}

template <int NG, int NT>
__device__ __forceinline__ void init_bias16(f32x4 (&acc)[NG][NT], const float* __restrict__ bias, int q) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 bv = bias ? ldg4(bias + 16 * t + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g][t] = bv;
  }
}

// bin[s] <- bf16(row[k(s,q,i)]) for a raw operand row (valid features [0, kvalid))
template <int KS, bool FULL, int DEPTH = 4>
__device__ __forceinline__ void load_raw16(bf16x8 (&bin)[KS], const float* __restrict__ row, int kvalid, int q) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool pairs = !FULL && (kvalid & 1) == 0 && ((size_t)row & 7) == 0;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f32x4 lo, hi;
    if (FULL) {
      lo = ldg4(row + 32 * s + 4 * q);
      hi = ldg4(row + 32 * s + 16 + 4 * q);
    } else if (pairs) {
      // rows of an even number of floats at an 8-byte aligned base (the 102 input features of the node encoder: 408-byte rows):
      // 8-byte loads - half the load instructions of the scalar form, which is what this launch's input phase is made of
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const int k0 = 32 * s + 4 * q + r, k1 = k0 + 16;
        const f32x2 a = k0 < kvalid ? *(const GW_AS1 f32x2*)(row + k0) : f32x2{0.f, 0.f};
        const f32x2 b = k1 < kvalid ? *(const GW_AS1 f32x2*)(row + k1) : f32x2{0.f, 0.f};
        lo[r] = a.x;
        lo[r + 1] = a.y;
        hi[r] = b.x;
        hi[r + 1] = b.y;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k0 = 32 * s + 4 * q + r, k1 = k0 + 16;
        lo[r] = k0 < kvalid ? ldg1(row + k0) : 0.f;
        hi[r] = k1 < kvalid ? ldg1(row + k1) : 0.f;
      }
    }
    bin[s] = pack8(lo, hi);
    // at most 2 DEPTH row pieces in flight per group (DEPTH = 4 where registers are scarce: one wave per SIMD with two groups)
    if ((s & (DEPTH - 1)) == DEPTH - 1) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NT>
__device__ __forceinline__ void relu_to_bin(bf16x8 (&bin)[NT / 2], const f32x4 (&acc)[NT]) {
#pragma unroll
  for (int s = 0; s < NT / 2; ++s) bin[s] = pack8(relu4(acc[2 * s]), relu4(acc[2 * s + 1]));
}

__device__ __forceinline__ const float* operand_row16(const float* ptr, const int* idx, int rows_pb, int ld, int b, int k) {
  const int r = idx ? ldgi(idx + k) : k;
  return ptr + ((size_t)b * (size_t)rows_pb + (size_t)r) * (size_t)ld;
}

// K1S: 32-wide K-steps of a raw layer-1 operand (8: k = 256, 4: k <= 128, 1: k <= 32); HT / OT: hidden / output row tiles.
template <int K1S, bool K1FULL, int NSEG, int HT, int OT, int EPI, bool SINGLE, bool POST = false, bool HEAD = false, int NW = 4,
          int NG = 2>
__global__ __launch_bounds__(NW * 64, (NW * NG == 4 ? 2 : 1)) void chain16_kernel(const ChainArgs a) {
  constexpr int kCols16 = NW * NG * 16;  // columns per workgroup
  static_assert(EPI != EPI_EDGE || (NW == 4 && NG == 2), "the segment-sum epilogue walks 64 columns per round on 4 waves");
  extern __shared__ __attribute__((aligned(16))) char lds16[];
  constexpr int HTP = (HT + 3) / 4 * 4, OTP = (OT + 3) / 4 * 4;
  constexpr int HKS = HT / 2;  // K-steps of a layer fed by the hidden activations
  constexpr int H_STEP = HTP * 1024, O_STEP = OTP * 1024;
  constexpr int H_CS = (2 * H_STEP <= kBufBytes) ? 2 : 1, O_CS = (2 * O_STEP <= kBufBytes) ? 2 : 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  const int tile_c0 = blockIdx.x * kCols16;
  // (measured and dropped: per-batch tiles walked batch-innermost and XCD-aware, so that batch-shared operand rows come from the
  // XCD's L2 - the head launch 1.19 -> 1.18 ms, the mesh-sized node updates 0.168 -> 0.175 ms)

  int cc[NG], bb[NG], kk[NG];
  bool valid[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c_raw = tile_c0 + g * (NW * 16) + wave * 16 + j;
    valid[g] = c_raw < a.n_cols;
    cc[g] = valid[g] ? c_raw : a.n_cols - 1;
    bb[g] = cc[g] / a.cols_per_batch;
    kk[g] = cc[g] - bb[g] * a.cols_per_batch;
  }

  bool on[3], prj[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    on[i] = (i < NSEG) && (a.seg_k[i] > 0) && (a.seg_proj[i] == 0);
    prj[i] = (i < NSEG) && (a.seg_k[i] > 0) && (a.seg_proj[i] != 0);
  }
  const char* w1[3] = {(const char*)a.w1[0], (const char*)a.w1[1], (const char*)a.w1[2]};
  if (SINGLE) w1[0] = (const char*)a.proj_w[blockIdx.y];
  const char* w_mid = (const char*)a.w_mid;
  const char* w_out = (const char*)a.w_out;
  constexpr int K1_CS = H_CS;
  constexpr int K1FIRST = (K1S < K1_CS ? K1S : K1_CS) * H_STEP;
  const char* after_l1 = SINGLE ? nullptr : (a.n_mid > 0 ? w_mid : w_out);
  const int after_l1_bytes = SINGLE ? 0 : (a.n_mid > 0 ? H_CS * H_STEP : O_CS * O_STEP);
  int parity = 0;
  {
    const char* first = on[0] ? w1[0] : (on[1] ? w1[1] : (on[2] ? w1[2] : after_l1));
    const int first_bytes = (on[0] || on[1] || on[2]) ? K1FIRST : after_l1_bytes;
    issue_bytes<NW>(first, first_bytes, 0u, lane, wave);
  }

  // ---- layer 1 ----
  // One accumulator array and one B-operand array serve every layer (the output layer gets its own accumulator only
  // when its tile count differs from the hidden one): register allocation does not reuse them otherwise.
  constexpr int BKS = K1S > HKS ? K1S : HKS;
  constexpr bool SHARE_ACC = (OT == HT);
  f32x4 acc[NG][HT];
  bf16x8 bin[NG][BKS];
  init_bias16<NG, HT>(acc, a.b1, q);
  {
#pragma unroll
    for (int i = 0; i < NSEG; ++i) {
      if (on[i]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (K1S == 8 && a.seg_bf16k[i]) {
            // bf16 rows in K order (GW_LAYOUT_ROWS_BF16K: the aggregate written by the segment-aligned edge kernel): position
            // 32 s + 8 q + i holds feature k(s, q, i) - the lane's B fragment of K-step s is one 16-byte load, no conversion
            const __bf16* rowb = (const __bf16*)a.seg_ptr[i] + ((size_t)bb[g] * (size_t)a.seg_rows_pb[i] + (size_t)kk[g]) * (size_t)a.seg_ld[i];
#pragma unroll
            for (int s = 0; s < (K1S == 8 ? 8 : 0); ++s) bin[g][s] = *(const GW_AS1 bf16x8*)(rowb + 32 * s + 8 * q);
          } else {
            const float* row = operand_row16(a.seg_ptr[i], a.seg_idx[i], a.seg_rows_pb[i], a.seg_ld[i], bb[g], kk[g]);
            load_raw16<K1S, K1FULL, (NG == 1 ? 8 : 4)>(reinterpret_cast<bf16x8(&)[K1S]>(bin[g]), row, a.seg_k[i], q);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const char* nx = after_l1;
        int nb = after_l1_bytes;
#pragma unroll
        for (int i2 = NSEG - 1; i2 > i; --i2)
          if (on[i2]) {
            nx = w1[i2];
            nb = K1FIRST;
          }
        pass16<NW, NG, K1S, BKS, HT, HTP>(acc, bin, w1[i], nx, nb, lds16, parity, lane, wave, GW_TUNE16(a));
      } else if (prj[i]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (a.seg_half[i]) {  // fp16 product rows (GW_LAYOUT_ROWS_F16): 8 bytes per row tile and lane
            const int r = a.seg_idx[i] ? ldgi(a.seg_idx[i] + kk[g]) : kk[g];
            const _Float16* row = (const _Float16*)a.seg_ptr[i] + ((size_t)bb[g] * (size_t)a.seg_rows_pb[i] + (size_t)r) * (size_t)a.seg_ld[i];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const half4_t h = *(const GW_AS1 half4_t*)(row + 16 * t + 4 * q);
              acc[g][t] += f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
              if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
          } else {
            const float* row = operand_row16(a.seg_ptr[i], a.seg_idx[i], a.seg_rows_pb[i], a.seg_ld[i], bb[g], kk[g]);
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              acc[g][t] += ldg4(row + 16 * t + 4 * q);
              if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
    }
  }

  f32x4 o_sep[SHARE_ACC ? 1 : NG][SHARE_ACC ? 1 : OT];
  f32x4 (&o)[NG][OT] = *reinterpret_cast<f32x4(*)[NG][OT]>(SHARE_ACC ? (void*)acc : (void*)o_sep);
  if constexpr (!SINGLE) {
    // ---- middle layers (hidden -> hidden) ----
#pragma unroll 1
    for (int l = 0; l < a.n_mid; ++l) {
#pragma unroll
      for (int g = 0; g < NG; ++g) relu_to_bin<HT>(reinterpret_cast<bf16x8(&)[HKS]>(bin[g]), acc[g]);
      __builtin_amdgcn_sched_barrier(0);
      init_bias16<NG, HT>(acc, a.b_mid + l * (HT * 16), q);
      const bool last = (l + 1 == a.n_mid);
      const char* nx = last ? w_out : w_mid + (size_t)(l + 1) * HKS * H_STEP;
      const int nb = last ? O_CS * O_STEP : H_CS * H_STEP;
      pass16<NW, NG, HKS, BKS, HT, HTP>(acc, bin, w_mid + (size_t)l * HKS * H_STEP, nx, nb, lds16, parity, lane, wave, GW_TUNE16(a));
    }
    // ---- output layer ----
#pragma unroll
    for (int g = 0; g < NG; ++g) relu_to_bin<HT>(reinterpret_cast<bf16x8(&)[HKS]>(bin[g]), acc[g]);
    __builtin_amdgcn_sched_barrier(0);
    init_bias16<NG, OT>(o, a.b_out, q);
    pass16<NW, NG, HKS, BKS, OT, OTP>(o, bin, w_out, POST ? (const char*)a.proj_w[0] : (HEAD ? (const char*)a.hd_w1 : nullptr),
                              POST ? H_CS * H_STEP : (HEAD ? 2 * 8 * 1024 : 0), lds16, parity, lane, wave, GW_TUNE16(a));
  }

  // ---- LayerNorm over the OT*16 features of each column (eps 1e-5, biased variance), fp32 ----
  if (!SINGLE && a.gamma != nullptr) {
    constexpr float inv_n = 1.0f / (OT * 16);
    float mean[NG], rstd[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < OT; ++t) s += (o[g][t].x + o[g][t].y) + (o[g][t].z + o[g][t].w);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      mean[g] = s * inv_n;
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = o[g][t][r] - mean[g];
          v += d * d;
        }
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      rstd[g] = 1.0f / sqrtf(v * inv_n + 1e-5f);
    }
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      const f32x4 gm = ldg4(a.gamma + 16 * t + 4 * q);
      const f32x4 bt = ldg4(a.beta + 16 * t + 4 * q);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[g][t][r] = (o[g][t][r] - mean[g]) * rstd[g] * gm[r] + bt[r];
    }
  }

  // ---- residual ----
  if (!SINGLE && !HEAD && a.res_ptr != nullptr && !(GW_TUNE16(a) & 16)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float* rrow = operand_row16(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, bb[g], kk[g]);
#pragma unroll
      for (int t = 0; t < OT; ++t) {
        const int f0 = 16 * t + 4 * q;
        if (EPI == EPI_DEC) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f0 + r < a.out_cols) o[g][t][r] += ldg1(rrow + f0 + r);
        } else {
          o[g][t] += ldg4(rrow + f0);
          if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  // ---- store ----
  float* outp = SINGLE ? a.proj_out[blockIdx.y] : a.out;
  if (!HEAD && outp != nullptr && !(GW_TUNE16(a) & 8)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (valid[g]) {
        float* orow = outp + (size_t)cc[g] * (size_t)a.out_ld;
#pragma unroll
        for (int t = 0; t < OT; ++t) {
          const int f0 = 16 * t + 4 * q;
          if (SINGLE && a.proj_half) {  // (out_ld counts halves)
            stg_half4((_Float16*)outp + (size_t)cc[g] * (size_t)a.out_ld + f0, o[g][t]);
          } else if (EPI == EPI_DEC) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f0 + r < a.out_cols) stg1(orow + f0 + r, o[g][t][r]);
          } else {
            stg4(orow + f0, o[g][t]);
          }
        }
      }
    }
  }


  // ---- HEAD: the output head on the new rows while they are still in registers (ChainArgs): 256 -> 128 relu -> 128 relu ->
  // <= 80 features (+ residual rows), e.g. AssimilatorDecoder.node_decoder + the Decoder residual (assimilator_decoder.py:197,
  // decoder.py:93) behind the decoder's node update: the [rows, 256] table between them (1 GB at 1 degree, batch 16) is never
  // written or read ----
  if constexpr (HEAD) {
    static_assert(!HEAD || (OT == 16 && HT == 16 && SHARE_ACC && !POST && !SINGLE), "HEAD follows a 256-wide node update");
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int s = 0; s < 8; ++s) bin[g][s] = pack8(o[g][2 * s], o[g][2 * s + 1]);  // (no relu: the head's input is the LayerNorm output)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int HS = 8 * 1024;  // bytes of one K-step of a packed slice with <= 8 row tiles
    f32x4 hh[NG][8];
    init_bias16<NG, 8>(hh, a.hd_b1, q);
    pass16<NW, NG, 8, BKS, 8, 8>(hh, bin, (const char*)a.hd_w1, (const char*)a.hd_w2, 2 * HS, lds16, parity, lane, wave, GW_TUNE16(a));
#pragma unroll
    for (int g = 0; g < NG; ++g) relu_to_bin<8>(reinterpret_cast<bf16x8(&)[4]>(bin[g]), hh[g]);
    __builtin_amdgcn_sched_barrier(0);
    init_bias16<NG, 8>(hh, a.hd_b2, q);
    pass16<NW, NG, 4, BKS, 8, 8>(hh, bin, (const char*)a.hd_w2, (const char*)a.hd_w3, 2 * HS, lds16, parity, lane, wave, GW_TUNE16(a));
#pragma unroll
    for (int g = 0; g < NG; ++g) relu_to_bin<8>(reinterpret_cast<bf16x8(&)[4]>(bin[g]), hh[g]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 y[NG][5];
    init_bias16<NG, 5>(y, a.hd_b3, q);
    pass16<NW, NG, 4, BKS, 5, 8>(y, bin, (const char*)a.hd_w3, nullptr, 0, lds16, parity, lane, wave, GW_TUNE16(a));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (valid[g]) {
        const float* rrow = a.res_ptr ? operand_row16(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, bb[g], kk[g]) : nullptr;
        float* orow = a.out + (size_t)cc[g] * (size_t)a.out_ld;
        // rows of an even number of floats at 8-byte aligned bases (78 outputs, 102-float feature rows): 8-byte accesses
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const bool pairs = (a.out_cols & 1) == 0 && ((size_t)orow & 7) == 0 && ((size_t)rrow & 7) == 0;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const int f0 = 16 * t + 4 * q;
          if (pairs) {
#pragma unroll
            for (int r = 0; r < 4; r += 2)
              if (f0 + r < a.out_cols) {
                const f32x2 rv = rrow ? *(const GW_AS1 f32x2*)(rrow + f0 + r) : f32x2{0.f, 0.f};
                *(GW_AS1 f32x2*)(orow + f0 + r) = f32x2{y[g][t][r] + rv.x, y[g][t][r + 1] + rv.y};
              }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f0 + r < a.out_cols) stg1(orow + f0 + r, y[g][t][r] + (rrow ? ldg1(rrow + f0 + r) : 0.f));
          }
        }
      }
    }
  }

  // ---- POST: the next block's layer-1 products of the new rows, while they are still in registers (see ChainArgs) ----
  if constexpr (POST) {
    static_assert(!POST || (OT == 16 && HT == 16 && SHARE_ACC), "POST works on 256-wide rows");
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int s = 0; s < 8; ++s) bin[g][s] = pack8(o[g][2 * s], o[g][2 * s + 1]);
    __builtin_amdgcn_sched_barrier(0);
    if (a.zero_rows != nullptr) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (valid[g]) {
          float* zrow = a.zero_rows + (size_t)cc[g] * 256;
#pragma unroll
          for (int t = 0; t < 16; ++t) stg4(zrow + 16 * t + 4 * q, f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
#pragma unroll 1
    for (int sl = 0; sl < a.n_post; ++sl) {
      init_bias16<NG, HT>(acc, nullptr, q);  // (acc aliases o: the new rows have been stored and packed into bin)
      const char* nx = sl + 1 < a.n_post ? (const char*)a.proj_w[sl + 1] : nullptr;
      pass16<NW, NG, 8, BKS, HT, HTP>(acc, bin, (const char*)a.proj_w[sl], nx, H_CS * H_STEP, lds16, parity, lane, wave, GW_TUNE16(a));
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (valid[g] && !(GW_TUNE16(a) & 32)) {
          if (a.proj_half) {
            _Float16* prow = (_Float16*)a.proj_out[sl] + (size_t)cc[g] * 256;
#pragma unroll
            for (int t = 0; t < HT; ++t) stg_half4(prow + 16 * t + 4 * q, acc[g][t]);
          } else {
            float* prow = a.proj_out[sl] + (size_t)cc[g] * 256;
#pragma unroll
            for (int t = 0; t < HT; ++t) stg4(prow + 16 * t + 4 * q, acc[g][t]);
          }
        }
    }
  }

  // ---- segment sum over destination-sorted columns, 64 columns at a time through LDS (see gw_edge.hip) ----
  if (EPI == EPI_EDGE) {
    float* stage = (float*)(lds16 + kLdsWeights);
    int* gdl = (int*)(stage + kStageFloats16);
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      __syncthreads();  // previous round's readers are done
      {
        float* srow = stage + (wave * 16 + j) * kStageLd16 + 4 * q;
        // (dynamic g: select the group's registers with a fully unrolled compare chain)
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2)
          if (g2 == g) {
#pragma unroll
            for (int t = 0; t < OT; ++t) *(f32x4*)(srow + 16 * t) = o[g2][t];
            if (q == 0) gdl[wave * 16 + j] = valid[g2] ? bb[g2] * a.agg_rows_pb + ldgi(a.agg_idx + kk[g2]) : -1;
          }
      }
      __syncthreads();
      // all 64 LDS reads first; run ends are a 64-bit scalar mask (lane i compares column i with column i + 1): the walk
      // is straight-line code testing one bit per column (see gw_edge.hip)
      const int f = threadIdx.x;
      float vv[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) vv[i] = stage[i * kStageLd16 + f];
      const int gdv = gdl[lane];
      const int gdn = gdl[lane < 63 ? lane + 1 : lane];
      const unsigned long long ends = __ballot(lane == 63 || gdn != gdv);
      // deterministic mode: carry records instead of atomics (gw_internal.hpp; same scheme as gw_edge.hip)
      bool open_lo = true, open_hi = true;
      float* rec = nullptr;
      if (a.carry != nullptr) {
        const int chunk_c0 = tile_c0 + g * 64;
        rec = a.carry + (size_t)(chunk_c0 >> 6) * kCarryFloats;
        const int c_prev = chunk_c0 - 1, c_next = chunk_c0 + 64;
        int gd_prev = -2, gd_next = -2;
        if (c_prev >= 0 && c_prev < a.n_cols) {
          const int bp = c_prev / a.cols_per_batch;
          gd_prev = bp * a.agg_rows_pb + ldgi(a.agg_idx + (c_prev - bp * a.cols_per_batch));
        }
        if (c_next < a.n_cols) {
          const int bn = c_next / a.cols_per_batch;
          gd_next = bn * a.agg_rows_pb + ldgi(a.agg_idx + (c_next - bn * a.cols_per_batch));
        }
        open_lo = gd_prev == __builtin_amdgcn_readlane(gdv, 0);
        open_hi = gd_next == __builtin_amdgcn_readlane(gdv, 63);
        if (f == 0 && chunk_c0 < a.n_cols) {
          rec[512] = __int_as_float(-1);
          rec[513] = __int_as_float(-1);
          rec[514] = __int_as_float(0);
        }
      }
      float run = 0.f;
      bool first = true;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        run += vv[i];
        if (__builtin_expect((ends >> i) & 1ull, 0)) {
          const int cur = __builtin_amdgcn_readlane(gdv, i);
          if (cur >= 0) {
            float* dstp = a.agg + (size_t)cur * 256 + f;
            if (rec != nullptr) {
              const bool lo = first && open_lo, hi = i == 63 && open_hi;
              if (lo) {
                rec[f] = run;
                if (f == 0) {
                  rec[512] = __int_as_float(cur);
                  if (hi) rec[514] = __int_as_float(1);
                }
              } else if (hi) {
                rec[256 + f] = run;
                if (f == 0) rec[513] = __int_as_float(cur);
              } else {
                stg1(dstp, run);
              }
            } else if (first || i == 63) {
              __hip_atomic_fetch_add((GW_AS1 float*)dstp, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              stg1(dstp, run);
            }
          }
          first = false;
          run = 0.f;
        }
      }
    }
  }
}

// ---- weight packing: nn.Linear [n_out, k_total] slice -> bf16 MFMA A-operand stream -------------------------------
// out[s][tile][lane][i] = bf16(W[16 tile + (lane & 15)][k_lo + 32 s + 16 (i >> 2) + 4 (lane >> 4) + (i & 3)]), 0 outside
__global__ void pack_linear_bf16_kernel(const float* __restrict__ w, int n_out, int k_total, int k_lo, int kseg, int ntp,
                                        int nsteps, __bf16* __restrict__ out) {
  const size_t total = (size_t)nsteps * ntp * 512;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const int tile = (int)((e >> 9) % ntp);
    const int s = (int)((e >> 9) / ntp);
    const int f = 16 * tile + (lane & 15);
    const int kx = 32 * s + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3);
    out[e] = (__bf16)((f < n_out && kx < kseg) ? w[(size_t)f * k_total + k_lo + kx] : 0.f);
  }
}

template <typename K>
int launch16(K kernel, ChainArgs& a, void* stream, int grid_y, int lds_bytes, int threads = 256, int cols = 128) {
  static DeviceOnce once;  // per template instantiation and device
  if (once.first()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsEdge);
  const int grid = (a.n_cols + cols - 1) / cols;
  hipLaunchKernelGGL(kernel, dim3(grid, grid_y), dim3(threads), lds_bytes, (hipStream_t)stream, a);
  return check_launch("chain16_kernel launch");
}

}  // namespace

namespace gw {

int chain16_launch(int kind, ChainArgs& a, int k_in, int hidden, int n_out, int grid_y, void* stream) {
#ifdef GW_TUNING
  {
    static const int t16 = GW_TUNE("GW_CHAIN16_TUNE", 0);
    a.tune16 = t16;
  }
#endif
  // The row-wise launches of the fused forward (node update, + post products, + head, node encoder + post products) run as
  // 4 waves x 1 group with two workgroups per CU (see the note above).  Tuning builds: GW_CHAIN16_NW=4 selects 4 x 2, 8 selects 8 x 1.
#ifdef GW_TUNING
  static const int nw = GW_TUNE("GW_CHAIN16_NW", 41);
  if (nw == 8) {
    switch (kind) {
      case 2:
        return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, false, 8, 1>, a, stream, 1, kLdsWeights, 512);
      case 4:
        return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, true, false, 8, 1>, a, stream, 1, kLdsWeights, 512);
      case 6:
        return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, true, 8, 1>, a, stream, 1, kLdsWeights, 512);
      case 5:
        if (hidden == 256 && n_out == 256 && k_in <= 128 && k_in > 32)
          return launch16(chain16_kernel<4, false, 1, 16, 16, EPI_ROWS, false, true, false, 8, 1>, a, stream, 1, kLdsWeights, 512);
        break;
      default:
        break;
    }
  }
  if (nw == 12 || nw == 16) {  // one workgroup of 12 / 16 waves per CU: 192 / 256 columns share one weight stream
    const int th = nw * 64, cl = nw * 16;
    switch (kind) {
      case 2:
        return nw == 12 ? launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, false, 12, 1>, a, stream, 1, kLdsWeights, th, cl)
                        : launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, false, 16, 1>, a, stream, 1, kLdsWeights, th, cl);
      case 4:
        return nw == 12 ? launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, true, false, 12, 1>, a, stream, 1, kLdsWeights, th, cl)
                        : launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, true, false, 16, 1>, a, stream, 1, kLdsWeights, th, cl);
      case 6:
        return nw == 12 ? launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, true, 12, 1>, a, stream, 1, kLdsWeights, th, cl)
                        : launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, true, 16, 1>, a, stream, 1, kLdsWeights, th, cl);
      case 5:
        if (hidden == 256 && n_out == 256 && k_in <= 128 && k_in > 32)
          return nw == 12 ? launch16(chain16_kernel<4, false, 1, 16, 16, EPI_ROWS, false, true, false, 12, 1>, a, stream, 1, kLdsWeights, th, cl)
                          : launch16(chain16_kernel<4, false, 1, 16, 16, EPI_ROWS, false, true, false, 16, 1>, a, stream, 1, kLdsWeights, th, cl);
        break;
      default:
        break;
    }
  }
  if (nw == 4) {
    switch (kind) {
      case 2:
        return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false>, a, stream, 1, kLdsWeights);
      case 4:
        return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, true>, a, stream, 1, kLdsWeights);
      case 6:
        return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, true>, a, stream, 1, kLdsWeights);
      case 5:
        if (hidden == 256 && n_out == 256 && k_in <= 128 && k_in > 32)
          return launch16(chain16_kernel<4, false, 1, 16, 16, EPI_ROWS, false, true>, a, stream, 1, kLdsWeights);
        break;
      default:
        break;
    }
  }
#endif
  switch (kind) {
    case 0:  // mlp rows
      if (hidden == 256 && n_out == 256) {
        if (k_in <= 32) return launch16(chain16_kernel<1, false, 1, 16, 16, EPI_ROWS, false>, a, stream, 1, kLdsWeights);
        if (k_in <= 128) return launch16(chain16_kernel<4, false, 1, 16, 16, EPI_ROWS, false>, a, stream, 1, kLdsWeights);
        if (k_in == 256) return launch16(chain16_kernel<8, true, 1, 16, 16, EPI_ROWS, false>, a, stream, 1, kLdsWeights);
      } else if (hidden == 256 && n_out <= 80 && k_in == 256) {
        return launch16(chain16_kernel<8, true, 1, 16, 5, EPI_DEC, false>, a, stream, 1, kLdsWeights);
      } else if (hidden == 128 && n_out <= 80 && k_in == 256) {
        return launch16(chain16_kernel<8, true, 1, 8, 5, EPI_DEC, false>, a, stream, 1, kLdsWeights);
      }
      return set_error(GW_E_UNSUPPORTED, "bf16 mlp: unsupported (hidden, n_out, k) combination");
    case 1:
      return launch16(chain16_kernel<8, true, 3, 16, 16, EPI_EDGE, false>, a, stream, 1, kLdsEdge);
    case 2:
      return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, false, 4, 1>, a, stream, 1, kLdsWeights, 256, 64);
    case 3:
      return launch16(chain16_kernel<8, true, 1, 16, 16, EPI_ROWS, true>, a, stream, grid_y, kLdsWeights);
    case 4:
      return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, true, false, 4, 1>, a, stream, 1, kLdsWeights, 256, 64);
    case 6:  // node update + output head (decoder)
      return launch16(chain16_kernel<8, true, 2, 16, 16, EPI_ROWS, false, false, true, 4, 1>, a, stream, 1, kLdsWeights, 256, 64);
    case 5:  // mlp rows + POST products of the output rows (node encoder -> layer-1 products of the encoder's edge MLP)
      if (hidden == 256 && n_out == 256 && k_in <= 128 && k_in > 32)
        return launch16(chain16_kernel<4, false, 1, 16, 16, EPI_ROWS, false, true, false, 4, 1>, a, stream, 1, kLdsWeights, 256, 64);
      return set_error(GW_E_UNSUPPORTED, "bf16 mlp + post products: hidden 256, 256 outputs, 33..128 inputs");
  }
  return set_error(GW_E_BADARG, "chain16_launch: bad kind");
}

}  // namespace gw

extern "C" {

// K-steps (32 input features each) of a packed bf16 slice: the layer-1 kernels exist for 1, 4 and 8 steps (k <= 32, k <= 128,
// 256) and stream exactly that many, so a narrower slice is zero-padded to its variant's step count (see gw_packed_floats).
static int packed_steps_bf16(int kseg) { return kseg <= 32 ? 1 : (kseg <= 128 ? 4 : (kseg + 31) / 32); }

size_t gw_packed_bytes_bf16(int n_out, int k_lo, int k_hi) {
  const int kseg = k_hi - k_lo;
  const int nsteps = packed_steps_bf16(kseg);
  const int ntp = (((n_out + 15) / 16) + 3) / 4 * 4;
  return (size_t)nsteps * ntp * 1024;
}

int gw_pack_linear_bf16(const float* w, int n_out, int k_total, int k_lo, int k_hi, void* out, void* stream) {
  if (!w || !out || n_out <= 0 || k_lo < 0 || k_hi <= k_lo || k_hi > k_total)
    return gw::set_error(GW_E_BADARG, "gw_pack_linear_bf16: bad arguments");
  const int kseg = k_hi - k_lo;
  const int nsteps = packed_steps_bf16(kseg);
  const int ntp = (((n_out + 15) / 16) + 3) / 4 * 4;
  const size_t total = (size_t)nsteps * ntp * 512;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_linear_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, n_out, k_total, k_lo, kseg, ntp,
                     nsteps, (__bf16*)out);
  return gw::check_launch("pack_linear_bf16_kernel launch");
}

}  // extern "C"
