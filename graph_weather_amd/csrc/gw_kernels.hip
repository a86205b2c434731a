// gw_kernels.hip - CDNA4 (gfx950) kernels for the graph_weather message-passing hot path.
//
// Design:  every MLP of the model is evaluated in the *transposed* form
//       H_out[feature][column] = W[feature][k] * H_in[k][column]
// with `column` = one edge / node, 16 columns per 64-lane wave.  With v_mfma_f32_16x16x4_f32 the weight matrix
// is the A operand (A[i = lane&15][k = lane>>4]) and the activations are the B operand (B[k = lane>>4][j = lane&15]).
// The accumulator layout (col = lane&15, row = 4*(lane>>4) + r) holds, for column j, exactly the features a lane
// must supply as B operand of the next layer if the K dimension is walked in the order
//       k(s, q) = 16*(s>>2) + 4*q + (s&3)        (s = K-step, q = lane>>4)
// so a 3-layer MLP + LayerNorm + residual runs register-resident: no transposes, no LDS traffic for activations.
// LDS carries only the packed weight stream (shared by the 4 waves of a workgroup, filled by global_load_lds DMA,
// double buffered).  A wave needs < 256 registers, so two independent 64-column workgroups share a CU and one's
// gathers / LayerNorm / segment-sum epilogue overlap the other's MFMAs.
// fp32 in / fp32 accumulate MFMA == an fmaf chain, so the result is fp32-exact up to summation order.
//
// Reference statements implemented: graph_net_block.py:45-61 (MLP), :131-137 (EdgeProcessor), :184-193
// (NodeProcessor incl. scatter_sum :188), losses.py:66-94 (NormalizedMSELoss).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/gw_amd.h"
#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

// in[8c .. 8c+7] <- row[k(s,q)] for the K-steps of chunk c (full 16-byte aligned rows)
template <int NSTEPS>
__device__ __forceinline__ void load_operand_slice(float (&in)[NSTEPS], const float* __restrict__ row, int c, int q) {
#pragma unroll
  for (int i = 2 * c; i < 2 * c + 2; ++i) {
    if (4 * i + 3 < NSTEPS) {
      const f32x4 v = ldg4(row + 16 * i + 4 * q);
      in[4 * i + 0] = v.x;
      in[4 * i + 1] = v.y;
      in[4 * i + 2] = v.z;
      in[4 * i + 3] = v.w;
    }
  }
}

// This wave's DMA pieces (p = wave, wave + 4, ...) of a chunk of NP 1-KiB pieces that are due behind K-step s when the pieces are
// spread over the first SPREAD K-steps of the chunk that is computing (everything is a compile-time constant after unrolling: no
// branch in the pass).  Round 6 (csrc/gw_noders.hip, scripts/gpu_timeline_rs.py): a global_load_lds blocks its wave for 60 - 180
// cycles; issued all at once behind the barrier, every wave of a SIMD is blocked at the same time and the matrix pipe idles -
// spread between the MFMA groups, the partner wave owns the pipe meanwhile (gw_edge.hip has done this since round 1).
__device__ __forceinline__ void issue_chunk_step(const float* __restrict__ g, int np /* pieces of the chunk */, int spread, float* ldsbuf,
                                                 int lane, int wave, int s) {
  const int pw = (np + 3) >> 2;  // pieces per wave (<= 8: a chunk is at most 32 KiB)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < pw && i * spread / pw == s) {
      const int pc = wave + 4 * i;
      if ((np & 3) == 0 || pc < np)
        glds16_asm_s(g + (size_t)pc * 256, (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)pc * 1024u));
    }
  }
}

// One K-pass of a layer: acc[t] += W[16t.., k] * in[k], K = 4*NSTEPS, NT row tiles of 16 features.
// Protocol: the first chunk of this pass has already been issued into buffer `parity`.
// With RELOAD, the 8 operand registers a chunk has consumed are refilled (one chunk later) with the same
// k-slice of the NEXT layer-1 operand (rows are full 256-float rows), so the gather of operand i+1 streams in
// underneath the MFMAs of operand i at no extra register cost; the slice consumed by the last chunk is
// refilled during chunk 0 of the next pass (`tail_row`).
template <int NSTEPS, int NT, bool RELOAD>
__device__ __forceinline__ void mma_pass(f32x4 (&acc)[NT], float (&in)[NSTEPS], const float* __restrict__ gw,
                                         const float* __restrict__ next_gw, int next_floats, float* lds, int& parity,
                                         int lane, int wave, const float* __restrict__ tail_row, bool do_tail,
                                         const float* __restrict__ next_row, bool do_next, int q) {
  constexpr int NT4 = (NT + 3) / 4;
  constexpr int STEPF = NT4 * 256;
  constexpr int NCH = (NSTEPS + kChunkSteps - 1) / kChunkSteps;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int nsteps_c = (NSTEPS - c * kChunkSteps) < kChunkSteps ? (NSTEPS - c * kChunkSteps) : kChunkSteps;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunk c has landed for every wave; nobody still reads the other buffer
    float* other = lds + (parity ^ 1) * kLdsBufFloats;
    // the next chunk of this pass: its pieces go out between the K-steps below; the first chunk of the NEXT pass (a size only
    // known at run time) all at once
    const int nn_c = (c + 1 < NCH) ? (((NSTEPS - (c + 1) * kChunkSteps) < kChunkSteps) ? (NSTEPS - (c + 1) * kChunkSteps) : kChunkSteps) : 0;
    const int np_c = nn_c * STEPF / 256;                           // 1-KiB pieces of that chunk (a constant once unrolled)
    const int spread_c = nsteps_c * 3 / 4 > 0 ? nsteps_c * 3 / 4 : 1;  // K-steps of THIS chunk that carry them
    if (c + 1 >= NCH && next_gw != nullptr) issue_chunk(next_gw, next_floats, other, lane, wave);
    if (RELOAD) {
      // Operand registers are refilled one chunk behind their consumption, right after the barrier, so the
      // gathers have a whole chunk of MFMAs to land before the next vmcnt(0).
      if (c == 0) {
        if (do_tail) load_operand_slice<NSTEPS>(in, tail_row, NCH - 1, q);   // this operand's last 8 registers
      } else {
        if (do_next) load_operand_slice<NSTEPS>(in, next_row, c - 1, q);     // next operand, slice consumed last chunk
      }
    }
    const float* buf = lds + parity * kLdsBufFloats + lane * 4;
    f32x4 a_cur[NT4];
#pragma unroll
    for (int b4 = 0; b4 < NT4; ++b4) a_cur[b4] = *(const f32x4*)(buf + b4 * 256);
#pragma unroll
    for (int s = 0; s < kChunkSteps; ++s) {
      if (s < nsteps_c) {
        f32x4 a_nxt[NT4];
        if (s + 1 < nsteps_c) {
#pragma unroll
          for (int b4 = 0; b4 < NT4; ++b4) a_nxt[b4] = *(const f32x4*)(buf + (s + 1) * STEPF + b4 * 256);
        }
        const float b = in[c * kChunkSteps + s];
        __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of step s+1 ahead of the MFMAs of step s
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t >> 2][t & 3], b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (np_c > 0) issue_chunk_step(gw + (size_t)(c + 1) * kChunkSteps * STEPF, np_c, spread_c, other, lane, wave, s);
        if (s + 1 < nsteps_c) {
#pragma unroll
          for (int b4 = 0; b4 < NT4; ++b4) a_cur[b4] = a_nxt[b4];
        }
      }
    }
    parity ^= 1;
  }
}

// First pass after layer 1.  Its B operand is relu(layer-1 accumulator + gathered pre-projected operand rows),
// and it is *produced slice by slice*: chunk c only needs hidden features 32c..32c+31, i.e. accumulator tiles 2c,
// 2c+1 plus the matching 2 x 16-byte pieces of each projected row.  The pieces for slice c+1 are requested while
// chunk c computes, so the per-edge gathers (1 KiB per operand and column) stream in underneath the MFMAs instead
// of forming a serial prologue (measured: 38k of 209k cycles per tile in the decoder edge update).
template <int HTI, int NT>
__device__ __forceinline__ void mma_pass_produce(f32x4 (&acc)[NT], const f32x4 (&src)[HTI], f32x4 (&tmp)[3][2],
                                                 const float* __restrict__ prow0, const float* __restrict__ prow1,
                                                 const float* __restrict__ prow2, bool p0, bool p1, bool p2,
                                                 const float* __restrict__ gw, const float* __restrict__ next_gw,
                                                 int next_floats, float* lds, int& parity, int lane, int wave, int q,
                                                 float* __restrict__ save_row) {
  constexpr int NT4 = (NT + 3) / 4;
  constexpr int STEPF = NT4 * 256;
  constexpr int NCH = HTI / 2;  // 8 K-steps (two 16-feature tiles) per chunk
  static_assert(HTI % 2 == 0 && kChunkSteps == 8, "slice = one chunk = two accumulator tiles");
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* other = lds + (parity ^ 1) * kLdsBufFloats;
    if (c + 1 >= NCH && next_gw != nullptr) issue_chunk(next_gw, next_floats, other, lane, wave);  // (next pass: all at once)
    f32x4 v0 = src[2 * c], v1 = src[2 * c + 1];
    if (p0) { v0 += tmp[0][0]; v1 += tmp[0][1]; }
    if (p1) { v0 += tmp[1][0]; v1 += tmp[1][1]; }
    if (p2) { v0 += tmp[2][0]; v1 += tmp[2][1]; }
    float in8[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      in8[r] = fmaxf(v0[r], 0.f);
      in8[4 + r] = fmaxf(v1[r], 0.f);
    }
    if (save_row != nullptr) {  // training: relu output of layer 1 (features 32c + 4q.., 32c + 16 + 4q..)
      stg4(save_row + 32 * c + 4 * q, f32x4{in8[0], in8[1], in8[2], in8[3]});
      stg4(save_row + 32 * c + 16 + 4 * q, f32x4{in8[4], in8[5], in8[6], in8[7]});
    }
    if (c + 1 < NCH) {
      const int f0 = 16 * (2 * c + 2) + 4 * q;
      if (p0) { tmp[0][0] = ldg4(prow0 + f0); tmp[0][1] = ldg4(prow0 + f0 + 16); }
      if (p1) { tmp[1][0] = ldg4(prow1 + f0); tmp[1][1] = ldg4(prow1 + f0 + 16); }
      if (p2) { tmp[2][0] = ldg4(prow2 + f0); tmp[2][1] = ldg4(prow2 + f0 + 16); }
    }
    const float* buf = lds + parity * kLdsBufFloats + lane * 4;
    f32x4 a_cur[NT4];
#pragma unroll
    for (int b4 = 0; b4 < NT4; ++b4) a_cur[b4] = *(const f32x4*)(buf + b4 * 256);
#pragma unroll
    for (int s = 0; s < kChunkSteps; ++s) {
      f32x4 a_nxt[NT4];
      if (s + 1 < kChunkSteps) {
#pragma unroll
        for (int b4 = 0; b4 < NT4; ++b4) a_nxt[b4] = *(const f32x4*)(buf + (s + 1) * STEPF + b4 * 256);
      }
      const float b = in8[s];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t >> 2][t & 3], b, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < NCH)  // the next chunk of this pass: its pieces between the K-steps (see issue_chunk_step)
        issue_chunk_step(gw + (size_t)(c + 1) * kChunkSteps * STEPF, kChunkSteps * STEPF / 256, 6, other, lane, wave, s);
      if (s + 1 < kChunkSteps) {
#pragma unroll
        for (int b4 = 0; b4 < NT4; ++b4) a_cur[b4] = a_nxt[b4];
      }
    }
    parity ^= 1;
  }
}

template <int NT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[NT], const float* __restrict__ bias, int q) {
  if (bias == nullptr) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = ldg4(bias + 16 * t + 4 * q);
}

// acc += P[row] for an operand that is already projected through its layer-1 weight slice (accumulator layout)
template <int NT>
__device__ __forceinline__ void add_projected(f32x4 (&acc)[NT], const float* __restrict__ row, int q) {
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] += ldg4(row + 16 * t + 4 * q);
}

// in[s] <- row[k(s,q)], k(s,q) = 16*(s>>2) + 4*q + (s&3)
template <int KSTEPS, bool FULL>
__device__ __forceinline__ void load_operand(float (&in)[KSTEPS], const float* __restrict__ row, int kvalid, int q) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool pairs = !FULL && (kvalid & 1) == 0 && ((size_t)row & 7) == 0;
#pragma unroll
  for (int i = 0; i < KSTEPS / 4; ++i) {
    if (FULL) {
      const f32x4 v = ldg4(row + 16 * i + 4 * q);
      in[4 * i + 0] = v.x;
      in[4 * i + 1] = v.y;
      in[4 * i + 2] = v.z;
      in[4 * i + 3] = v.w;
    } else if (pairs) {
      // rows of an even number of floats at an 8-byte aligned base (102 input features: 408-byte rows): 8-byte loads
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const int k = 16 * i + 4 * q + r;
        const f32x2 v = (k < kvalid) ? *(const GW_AS1 f32x2*)(row + k) : f32x2{0.f, 0.f};
        in[4 * i + r] = v.x;
        in[4 * i + r + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * i + 4 * q + r;
        in[4 * i + r] = (k < kvalid) ? ldg1(row + k) : 0.f;
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void relu_to_in(float (&in)[NT * 4], const f32x4 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) in[4 * t + r] = fmaxf(acc[t][r], 0.f);
}

__device__ __forceinline__ const float* operand_row(const float* ptr, const int* idx, int rows_pb, int ld, int b, int k) {
  const int r = idx ? ldgi(idx + k) : k;
  return ptr + ((size_t)b * (size_t)rows_pb + (size_t)r) * (size_t)ld;
}

template <int K1S, bool K1FULL, int NSEG, int HT, int OT, int EPI, bool SINGLE = false, bool POST = false, bool HEAD = false>
__global__ __launch_bounds__(kThreads, 2) void chain_kernel(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HS = HT * 4;                   // K-steps of a hidden layer
  constexpr int HSTEPF = ((HT + 3) / 4) * 256;  // floats per step, hidden-row layers
  constexpr int OSTEPF = ((OT + 3) / 4) * 256;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  const int c_raw = blockIdx.x * kColsPerWG + wave * kColsPerWave + j;
  const bool valid = c_raw < a.n_cols;
  const int c = valid ? c_raw : a.n_cols - 1;
  const int b = c / a.cols_per_batch;
  const int k = c - b * a.cols_per_batch;

  // Two workgroups share a CU (2 waves per SIMD).  Dispatched together and doing identical work they would run
  // in lockstep - both gathering, then both queueing on the matrix pipe - so neither hides the other's memory
  // phase (measured: MFMA busy 57 %).  Delaying the second batch of 256 workgroups by about half a tile puts the
  // pair on each CU in anti-phase: one gathers / normalises / scatters while the other owns the matrix pipe.
  if (a.stagger > 0 && (blockIdx.x >> 8) == 1) {  // only the second batch of the first wave: later workgroups start when a slot frees up, already out of phase
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  GW_STAMP(0)

  // ---- weight-stream schedule (wave uniform) ----
  // on[i]: operand i takes part in an MFMA pass (raw rows);  prj[i]: operand i is pre-projected (gather-add only)
  bool on[3], prj[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    on[i] = (i < NSEG) && (a.seg_k[i] > 0) && (a.seg_proj[i] == 0);
    prj[i] = (i < NSEG) && (a.seg_k[i] > 0) && (a.seg_proj[i] != 0);
  }
  const float* w1[3] = {a.w1[0], a.w1[1], a.w1[2]};
  if (SINGLE) w1[0] = a.proj_w[blockIdx.y];
  const float* after_l1 = SINGLE ? nullptr : (a.n_mid > 0 ? a.w_mid : a.w_out);
  const int after_l1_floats = SINGLE ? 0 : (a.n_mid > 0 ? kChunkSteps * HSTEPF : kChunkSteps * OSTEPF);
  constexpr int K1FIRST = (K1S < kChunkSteps ? K1S : kChunkSteps) * HSTEPF;
  int parity = 0;
  {
    const float* first = on[0] ? w1[0] : (on[1] ? w1[1] : (on[2] ? w1[2] : after_l1));
    const int first_floats = (on[0] || on[1] || on[2]) ? K1FIRST : after_l1_floats;
    issue_chunk(first, first_floats, lds, lane, wave);
  }

  // ---- layer 1 ----
  f32x4 acc[HT];
  f32x4 ptmp[3][2];
  const float* prow[3] = {nullptr, nullptr, nullptr};
  {
    const float* row[3] = {nullptr, nullptr, nullptr};
#pragma unroll
    for (int i = 0; i < NSEG; ++i)
      if (on[i] || prj[i]) row[i] = operand_row(a.seg_ptr[i], a.seg_idx[i], a.seg_rows_pb[i], a.seg_ld[i], b, k);
    float x[K1S];
    {
      const int f = on[0] ? 0 : (on[1] ? 1 : 2);
      if (on[0] || on[1] || on[2]) load_operand<K1S, K1FULL>(x, row[f], a.seg_k[f], q);
    }
    init_bias<HT>(acc, a.b1, q);
    if (!SINGLE) {
      // first 32-feature slice of every pre-projected operand (the rest streams in during the first hidden pass)
#pragma unroll
      for (int i = 0; i < NSEG; ++i) {
        prow[i] = row[i];
        if (prj[i]) {
          ptmp[i][0] = ldg4(row[i] + 4 * q);
          ptmp[i][1] = ldg4(row[i] + 16 + 4 * q);
        }
      }
    }
    GW_STAMP(1)
    constexpr bool RL = K1FULL && (NSEG > 1);
    bool tail_pending = false;  // the current operand's last register slice still has to be gathered
    if (on[0]) {
      const float* nx = on[1] ? w1[1] : (on[2] ? w1[2] : after_l1);
      const int nf = (on[1] || on[2]) ? K1FIRST : after_l1_floats;
      const bool more = on[1] || on[2];
      mma_pass<K1S, HT, RL>(acc, x, w1[0], nx, nf, lds, parity, lane, wave, nullptr, false, on[1] ? row[1] : row[2], more, q);
      tail_pending = more;
    }
    if (NSEG > 1 && on[1]) {
      const float* nx = on[2] ? w1[2] : after_l1;
      const int nf = on[2] ? K1FIRST : after_l1_floats;
      mma_pass<K1S, HT, RL>(acc, x, w1[1], nx, nf, lds, parity, lane, wave, row[1], tail_pending, row[2], on[2], q);
      tail_pending = on[2];
    }
    if (NSEG > 2 && on[2])
      mma_pass<K1S, HT, RL>(acc, x, w1[2], after_l1, after_l1_floats, lds, parity, lane, wave, row[2], tail_pending, nullptr, false, q);
  }

  GW_STAMP(2)
  f32x4 o[OT];
  f32x4 rres[OT];  // residual rows (prefetched during the last pass)
  if constexpr (SINGLE) {
    static_assert(!SINGLE || HT == OT, "single-layer mode stores the layer-1 accumulator");
#pragma unroll
    for (int t = 0; t < OT; ++t) o[t] = acc[t < HT ? t : 0];
    if (a.zero_rows != nullptr && valid && blockIdx.y == 0) {
      float* zrow = a.zero_rows + (size_t)c * 256;
#pragma unroll
      for (int t = 0; t < OT; ++t) stg4(zrow + 16 * t + 4 * q, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    if (a.relu_mask != nullptr) {  // out *= (mask > 0): the ReLU backward of the layer whose output gradient this product is
      const float* mrow = a.relu_mask + (size_t)c * 256;
#pragma unroll
      for (int t = 0; t < OT; ++t) {
        const f32x4 mv = ldg4(mrow + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(mv[r] > 0.f)) o[t][r] = 0.f;
      }
    }
  } else {
    {
      // ---- first middle layer (n_mid >= 1 is checked on the host): input produced slice by slice from the layer-1 accumulator ----
      f32x4 acc2[HT];
      init_bias<HT>(acc2, a.b_mid, q);
      {
        const bool last = (a.n_mid == 1);
        const float* nx = last ? a.w_out : a.w_mid + (size_t)HS * HSTEPF;
        const int nf = last ? kChunkSteps * OSTEPF : kChunkSteps * HSTEPF;
        float* srow = (a.save_h != nullptr && valid) ? a.save_h + (size_t)c * (size_t)a.save_ld : nullptr;
        mma_pass_produce<HT, HT>(acc2, acc, ptmp, prow[0], prow[1], prow[2], prj[0], prj[1], prj[2], a.w_mid, nx, nf, lds,
                                 parity, lane, wave, q, srow);
      }
      // ---- further middle layers (hidden -> hidden) ----
      float hin[HS];
#pragma unroll 1
      for (int l = 1; l < a.n_mid; ++l) {
        relu_to_in<HT>(hin, acc2);
        if (a.save_h != nullptr && valid) {
          float* srow = a.save_h + (size_t)l * (size_t)a.save_stride + (size_t)c * (size_t)a.save_ld;
#pragma unroll
          for (int t = 0; t < HT; ++t) stg4(srow + 16 * t + 4 * q, f32x4{hin[4 * t], hin[4 * t + 1], hin[4 * t + 2], hin[4 * t + 3]});
        }
        init_bias<HT>(acc2, a.b_mid + l * (HT * 16), q);
        const bool last = (l + 1 == a.n_mid);
        const float* nx = last ? a.w_out : a.w_mid + (size_t)(l + 1) * HS * HSTEPF;
        const int nf = last ? kChunkSteps * OSTEPF : kChunkSteps * HSTEPF;
        mma_pass<HS, HT, false>(acc2, hin, a.w_mid + (size_t)l * HS * HSTEPF, nx, nf, lds, parity, lane, wave, nullptr, false, nullptr, false, q);
      }
      GW_STAMP(3)
      // ---- output layer ----
      relu_to_in<HT>(hin, acc2);
      if (a.save_h != nullptr && valid) {
        float* srow = a.save_h + (size_t)a.n_mid * (size_t)a.save_stride + (size_t)c * (size_t)a.save_ld;
#pragma unroll
        for (int t = 0; t < HT; ++t) stg4(srow + 16 * t + 4 * q, f32x4{hin[4 * t], hin[4 * t + 1], hin[4 * t + 2], hin[4 * t + 3]});
      }
      init_bias<OT>(o, a.b_out, q);
      if (EPI != EPI_DEC && a.res_ptr != nullptr) {
        // fetch the residual rows underneath the last pass
        const float* rrow = operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, b, k);
#pragma unroll
        for (int t = 0; t < OT; ++t) rres[t] = ldg4(rrow + 16 * t + 4 * q);
      }
      // (HEAD: the head's first Linear follows - packed [128, 256]: 8 row tiles = 512 floats per K-step)
      mma_pass<HS, OT, false>(o, hin, a.w_out, POST ? a.proj_w[0] : (HEAD ? a.hd_w1 : nullptr),
                              POST ? kChunkSteps * HSTEPF : (HEAD ? kChunkSteps * 512 : 0), lds, parity, lane, wave, nullptr, false,
                              nullptr, false, q);
    }
  }

  GW_STAMP(4)
  // ---- LayerNorm over the OT*16 features of each column (eps 1e-5, biased variance) ----
  if (!SINGLE && a.gamma != nullptr) {
    if (a.save_y != nullptr && valid) {
      float* srow = a.save_y + (size_t)c * (size_t)(OT * 16);
#pragma unroll
      for (int t = 0; t < OT; ++t) stg4(srow + 16 * t + 4 * q, o[t]);
    }
    // heads and zero-padded narrow models normalise over their ln_width <= OT*16 real features: the padding rows of the
    // last Linear are zero, so they drop out of the sum and are masked out of the variance
    const int nfeat = a.ln_width;
    const bool narrow = nfeat != OT * 16;  // uniform: heads and zero-padded narrow models
    const float inv_n = narrow ? 1.0f / (float)nfeat : 1.0f / (OT * 16);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < OT; ++t) s += (o[t].x + o[t].y) + (o[t].z + o[t].w);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * inv_n;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = (narrow && 16 * t + 4 * q + r >= nfeat) ? 0.f : o[t][r] - mean;
        v += d * d;
      }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const float rstd = 1.0f / sqrtf(v * inv_n + 1e-5f);
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      const f32x4 gm = ldg4(a.gamma + 16 * t + 4 * q);
      const f32x4 bt = ldg4(a.beta + 16 * t + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] = (o[t][r] - mean) * rstd * gm[r] + bt[r];
    }
  }

  // ---- residual ----
  if (!SINGLE && !HEAD && a.res_ptr != nullptr) {
    if (EPI == EPI_DEC) {
      const float* rrow = operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, b, k);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const bool pairs = (a.out_cols & 1) == 0 && ((size_t)rrow & 7) == 0;  // (78 of the 102 features of a 408-byte row)
#pragma unroll
      for (int t = 0; t < OT; ++t) {
        const int f0 = 16 * t + 4 * q;
        if (pairs) {
#pragma unroll
          for (int r = 0; r < 4; r += 2)
            if (f0 + r < a.out_cols) {
              const f32x2 v = *(const GW_AS1 f32x2*)(rrow + f0 + r);
              o[t][r] += v.x;
              o[t][r + 1] += v.y;
            }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f0 + r < a.out_cols) o[t][r] += ldg1(rrow + f0 + r);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < OT; ++t) o[t] += rres[t];
    }
  }

  // ---- store ----
  float* outp = SINGLE ? a.proj_out[blockIdx.y] : a.out;
  if (!HEAD && outp != nullptr && valid) {
    float* orow = outp + (size_t)c * (size_t)a.out_ld;
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      const int f0 = 16 * t + 4 * q;
      if (EPI == EPI_DEC) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        if ((a.out_cols & 1) == 0 && ((size_t)orow & 7) == 0) {  // 78-float rows: 8-byte stores
#pragma unroll
          for (int r = 0; r < 4; r += 2)
            if (f0 + r < a.out_cols) *(GW_AS1 f32x2*)(orow + f0 + r) = f32x2{o[t][r], o[t][r + 1]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f0 + r < a.out_cols) stg1(orow + f0 + r, o[t][r]);
        }
      } else {
        stg4(orow + f0, o[t]);
      }
    }
  }

  // ---- HEAD: the output head on the new rows while they are still in registers (ChainArgs): 256 -> 128 relu -> 128 relu ->
  // <= 80 features (+ residual rows): AssimilatorDecoder.node_decoder + the Decoder residual (assimilator_decoder.py:197,
  // decoder.py:93) behind the decoder's node update; the [rows, 256] table between them (133 MB at 1 degree, batch 2) is never
  // written or read.  Same arithmetic, same order as the two launches (gw_node_update_forward, then gw_mlp_forward): bitwise. ----
  if constexpr (HEAD) {
    static_assert(!HEAD || (OT == 16 && HT == 16 && !POST && !SINGLE && EPI == EPI_ROWS), "HEAD follows a 256-wide node update");
    float xin[64];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xin[4 * t + r] = o[t < OT ? t : 0][r];
    f32x4 h1[8];
    init_bias<8>(h1, a.hd_b1, q);
    mma_pass<64, 8, false>(h1, xin, a.hd_w1, a.hd_w2, kChunkSteps * 512, lds, parity, lane, wave, nullptr, false, nullptr, false, q);
    f32x4 h2[8];
    init_bias<8>(h2, a.hd_b2, q);
    f32x4 none[3][2];
    mma_pass_produce<8, 8>(h2, h1, none, nullptr, nullptr, nullptr, false, false, false, a.hd_w2, a.hd_w3, kChunkSteps * 512, lds, parity,
                           lane, wave, q, nullptr);
    float hin2[32];
    relu_to_in<8>(hin2, h2);
    f32x4 y[5];
    init_bias<5>(y, a.hd_b3, q);
    mma_pass<32, 5, false>(y, hin2, a.hd_w3, nullptr, 0, lds, parity, lane, wave, nullptr, false, nullptr, false, q);
    if (valid) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float* rrow = a.res_ptr ? operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, b, k) : nullptr;
      float* orow = a.out + (size_t)c * (size_t)a.out_ld;
      const bool pairs = (a.out_cols & 1) == 0 && ((size_t)orow & 7) == 0 && ((size_t)rrow & 7) == 0;  // 78-float rows: 8-byte accesses
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const int f0 = 16 * t + 4 * q;
        if (pairs) {
#pragma unroll
          for (int r = 0; r < 4; r += 2)
            if (f0 + r < a.out_cols) {
              f32x2 v = f32x2{y[t][r], y[t][r + 1]};
              if (rrow) {
                const f32x2 rv = *(const GW_AS1 f32x2*)(rrow + f0 + r);
                v.x += rv.x;
                v.y += rv.y;
              }
              *(GW_AS1 f32x2*)(orow + f0 + r) = v;
            }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f0 + r < a.out_cols) stg1(orow + f0 + r, y[t][r] + (rrow ? ldg1(rrow + f0 + r) : 0.f));
        }
      }
    }
  }

  // ---- POST: the next block's layer-1 products of the new rows, while they are still in registers ----
  if constexpr (POST) {
    static_assert(!POST || (OT == 16 && HT == 16), "POST works on 256-wide rows");
    float xin[64];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xin[4 * t + r] = o[t < OT ? t : 0][r];
    if (a.zero_rows != nullptr && valid) {  // the aggregate buffer of the next block's edge update, zero-filled on the side
      float* zrow = a.zero_rows + (size_t)c * 256;
#pragma unroll
      for (int t = 0; t < 16; ++t) stg4(zrow + 16 * t + 4 * q, f32x4{0.f, 0.f, 0.f, 0.f});
    }
#pragma unroll 1
    for (int sl = 0; sl < a.n_post; ++sl) {
      f32x4 pacc[16];
      init_bias<16>(pacc, nullptr, q);
      const float* nx = sl + 1 < a.n_post ? a.proj_w[sl + 1] : nullptr;
      mma_pass<64, 16, false>(pacc, xin, a.proj_w[sl], nx, kChunkSteps * HSTEPF, lds, parity, lane, wave, nullptr, false, nullptr, false, q);
      if (valid) {
        float* prow = a.proj_out[sl] + (size_t)c * 256;
#pragma unroll
        for (int t = 0; t < 16; ++t) stg4(prow + 16 * t + 4 * q, pacc[t]);
      }
    }
  }

  GW_STAMP(5)
  // ---- segment sum over destination-sorted columns: shuffle scan + one atomicAdd per segment tail ----
  if (EPI == EPI_EDGE) {
    const int gd = valid ? (b * a.agg_rows_pb + ldgi(a.agg_idx + k)) : (-1 - j);
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int gu = __shfl_up(gd, off, 16);
      const bool take = (j >= off) && (gu == gd);
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float u = __shfl_up(o[t][r], off, 16);
          o[t][r] += take ? u : 0.f;
        }
    }
    const int gn = __shfl_down(gd, 1, 16);
    const bool tail = valid && (j == 15 || gn != gd);
    if (tail) {
      float* arow = a.agg + (size_t)gd * (size_t)(OT * 16);
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __hip_atomic_fetch_add((GW_AS1 float*)(arow + 16 * t + 4 * q + r), o[t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (a.dbg != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = gw_clock();
    if (threadIdx.x == 0 && (int)blockIdx.x < a.dbg_cap && blockIdx.y == 0) {
      unsigned long long* rec = a.dbg + (size_t)blockIdx.x * 16;
      for (int i = 0; i < 7; ++i) rec[i] = ts[i];
      rec[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      rec[9] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID
      rec[10] = blockIdx.x;
    }
  }
}

// ---- weight packing: nn.Linear [n_out, k_total] slice -> MFMA A-operand stream -------------------------
// out[s][b4][lane][i] = W[16*(4*b4+i) + (lane&15)][k_lo + 16*(s>>2) + 4*(lane>>4) + (s&3)]  (0 outside)
__global__ void pack_linear_kernel(const float* __restrict__ w, int n_out, int k_total, int k_lo, int kseg, int nt4,
                                   int nsteps, float* __restrict__ out) {
  const size_t total = (size_t)nsteps * nt4 * 256;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ti = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    const int b4 = (int)((i >> 8) % nt4);
    const int s = (int)((i >> 8) / nt4);
    const int f = 16 * (4 * b4 + ti) + (lane & 15);
    const int kk = 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3);
    out[i] = (f < n_out && kk < kseg) ? w[(size_t)f * k_total + k_lo + kk] : 0.f;
  }
}

__global__ void pad_vector_kernel(const float* __restrict__ v, int n, int npad, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) out[i] = i < n ? v[i] : 0.f;
}

// ---- gw_mlp_chain_backward: the Linear / ReLU chain of an MLP walked backwards, register-resident ---------------------------
// d_0 = d;  d_{i+1} = (d_i . W_i) * (h_i > 0) for the n_chain Linear layers above layer 1 (W_i: the layer's weight, streamed as
// the packed transposed block; h_i: the ReLU output that fed the layer, from the forward's activation save); then the input
// gradients of layer 1, d_n . W1[:, block], for up to 3 operand blocks.  The forward kernel's structure: the accumulator of one
// product is the B operand of the next, weights stream through LDS (double buffered DMA, two workgroups per CU); every d_i is
// stored once (the weight-gradient GEMMs read it) and never read back by this chain.
// LN (gw_mlp_ln_chain_backward): a.d is the gradient at the output of the MLP's LayerNorm; the kernel walks back through the norm
// in registers first (gw_device.hpp: ln_backward_rows16), stores the gradient at the norm's input once and feeds it to the first
// product.  The extras of that entry point: column sums of the last chain gradient (Linear_0's bias gradient) and rows added to a
// fan product before it is stored.  8 KiB of LDS behind the two weight buffers hold the four waves' column sums.
constexpr int kBwdScratchBytes = 4 * 512 * 4;
template <bool LN, bool EXTRA>
__global__ __launch_bounds__(kThreads, 2) void bwd_chain_kernel(const BwdChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HS = 64, HSTEPF = 1024;  // K-steps / floats per step of a 256 -> 256 product
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  const long long c_raw = (long long)blockIdx.x * kColsPerWG + wave * kColsPerWave + j;
  const bool valid = c_raw < a.n_rows;
  const long long c = valid ? c_raw : a.n_rows - 1;
  const int n_prod = a.n_chain + a.n_fan;
  float* red_all = lds + 2 * kLdsBufFloats;
  auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  int parity = 0;
  issue_chunk((const float*)a.w[0], kChunkSteps * HSTEPF, lds, lane, wave);
  float x[HS];
  // this row of the launch's input gradient: row c of d, or (LN launches) gathered through d_idx, plus a row of d_add
  const float* drow = a.d + (size_t)c * (size_t)a.d_ld;
  const float* drow2 = nullptr;
  if (LN && a.d_idx != nullptr) {
    const long long b = c / a.d_idx_n;
    const int k = (int)(c - b * a.d_idx_n);
    drow = a.d + ((size_t)b * (size_t)a.d_tab_rows_pb + (size_t)ldgi(a.d_idx + k)) * (size_t)a.d_ld;
    if (a.d_add != nullptr) drow2 = a.d_add + (size_t)c * (size_t)a.d_add_ld;
  }
  if constexpr (LN) {
    f32x4 g[16];
    ln_backward_rows16(g, a.ln_y + (size_t)c * 256, drow, drow2, a.ln_gamma, valid, q, j, red_all + wave * 512);
    float* orow = a.ln_dy + (size_t)c * 256;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (valid) stg4(orow + 16 * t + 4 * q, g[t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) x[4 * t + r] = g[t][r];
    }
    barrier();  // the four waves' column sums are in LDS
    ln_backward_flush(red_all, a.ln_dgamma, a.ln_dbeta, threadIdx.x);
  } else {
    load_operand<HS, true>(x, a.d + (size_t)c * (size_t)a.d_ld, 256, q);
  }
#pragma unroll 1
  for (int p = 0; p < n_prod; ++p) {
    const bool chain = p < a.n_chain;
    f32x4 mv[16];
    if (chain && !EXTRA) {  // the ReLU output that gates this product: fetched underneath its MFMAs
      const float* mrow = a.mask[p] + (size_t)c * 256;
#pragma unroll
      for (int t = 0; t < 16; ++t) mv[t] = ldg4(mrow + 16 * t + 4 * q);
    }
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* nx = p + 1 < n_prod ? (const float*)a.w[p + 1] : nullptr;
    mma_pass<HS, 16, false>(acc, x, (const float*)a.w[p], nx, nx ? kChunkSteps * HSTEPF : 0, lds, parity, lane, wave, nullptr, false, nullptr,
                            false, q);
    if (chain) {
      if (EXTRA) {  // (the instantiation with the extras has no registers for the prefetch: the mask is fetched behind the pass)
        const float* mrow = a.mask[p] + (size_t)c * 256;
#pragma unroll
        for (int t = 0; t < 16; ++t) mv[t] = ldg4(mrow + 16 * t + 4 * q);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = mv[t][r] > 0.f ? acc[t][r] : 0.f;
          acc[t][r] = v;
          x[4 * t + r] = v;  // (accumulator layout == B-operand layout of the next product)
        }
    } else if (EXTRA && a.add[p] != nullptr) {  // rows that join this input gradient (gw_mlp_ln_chain_backward: fan_add)
      const float* arow = a.add[p] + (size_t)c * (size_t)a.add_ld;
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] += ldg4(arow + 16 * t + 4 * q);
    } else if (EXTRA && !chain && ((a.add_d_mask >> p) & 1u)) {  // ... the launch's own input gradient (never materialised when gathered)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] += ldg4(drow + 16 * t + 4 * q);
      if (drow2 != nullptr) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] += ldg4(drow2 + 16 * t + 4 * q);
      }
    }
    if (valid) {
      float* orow = a.out[p] + (size_t)c * 256;
#pragma unroll
      for (int t = 0; t < 16; ++t) stg4(orow + 16 * t + 4 * q, acc[t]);
    }
    if (EXTRA && chain && p + 1 == a.n_chain && a.colsum != nullptr)  // (uniform) Linear_0's bias gradient: column sums of this gradient
      colsum_rows64(acc, valid, q, j, wave, threadIdx.x, red_all, a.colsum, barrier);
  }
}

template <bool LN, bool EXTRA>
int bwd_chain_launch_as(const BwdChainArgs& a, void* stream) {
  constexpr int ldsb = kLdsBytes + kBwdScratchBytes;
  const long long grid = (a.n_rows + kColsPerWG - 1) / kColsPerWG;
  static DeviceOnce once;  // per instantiation and device
  if (once.first()) (void)hipFuncSetAttribute((const void*)bwd_chain_kernel<LN, EXTRA>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipLaunchKernelGGL((bwd_chain_kernel<LN, EXTRA>), dim3((unsigned)grid), dim3(kThreads), ldsb, (hipStream_t)stream, a);
  return check_launch("bwd_chain_kernel launch");
}
int bwd_chain_launch(const BwdChainArgs& a, void* stream) {
  bool extra = a.colsum != nullptr || a.add_d_mask != 0;
  for (int i = 0; i < 5; ++i) extra = extra || a.add[i] != nullptr;
  if (a.ln_y != nullptr) return extra ? bwd_chain_launch_as<true, true>(a, stream) : bwd_chain_launch_as<true, false>(a, stream);
  return extra ? bwd_chain_launch_as<false, true>(a, stream) : bwd_chain_launch_as<false, false>(a, stream);
}

// ---- gw_pack_many: all slices / vectors of an MLP in one launch (blockIdx.y = item; the last y packs the vectors) --------
struct PackManyArgs {
  gw_pack_item m[GW_PACK_MAX_ITEMS];
  gw_pad_item v[GW_PACK_MAX_ITEMS];
  int nsteps[GW_PACK_MAX_ITEMS];  // K-steps of the item's stream
  int ntq[GW_PACK_MAX_ITEMS];     // fp32: row-tile quads (nt4); bf16: row tiles rounded up to 4 (ntp)
  int n_mats, n_vecs, bf16;
};

__global__ void pack_many_kernel(const PackManyArgs a) {
  const int it = blockIdx.y;
  if (it < a.n_mats) {
    const float* __restrict__ w = a.m[it].w;
    const long long sf = a.m[it].stride_f, sk = a.m[it].stride_k;
    const int n_out = a.m[it].n_out, kseg = a.m[it].kseg, ntq = a.ntq[it];
    if (!a.bf16) {  // (pack_linear_kernel's order)
      float* __restrict__ out = (float*)a.m[it].out;
      const size_t total = (size_t)a.nsteps[it] * ntq * 256;
      for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ti = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        const int b4 = (int)((i >> 8) % ntq);
        const int s = (int)((i >> 8) / ntq);
        const int f = 16 * (4 * b4 + ti) + (lane & 15);
        const int kk = 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3);
        out[i] = (f < n_out && kk < kseg) ? w[(long long)f * sf + (long long)kk * sk] : 0.f;
      }
    } else {        // (pack_linear_bf16_kernel's order, gw_bf16.hip; bf16 == 2: the split stream of gw_split.hip - per K-step the hi
                    // fragments of all tiles, then their lo fragments)
      __bf16* __restrict__ out = (__bf16*)a.m[it].out;
      const size_t total = (size_t)a.nsteps[it] * ntq * 512;
      const bool x3 = a.bf16 == 2;
      for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7);
        const int lane = (int)((e >> 3) & 63);
        const int tile = (int)((e >> 9) % ntq);
        const int s = (int)((e >> 9) / ntq);
        const int f = 16 * tile + (lane & 15);
        const int kx = 32 * s + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3);
        const float v = (f < n_out && kx < kseg) ? w[(long long)f * sf + (long long)kx * sk] : 0.f;
        if (x3) {
          const __bf16 h = (__bf16)v;
          const size_t o = ((size_t)s * 2 * ntq + tile) * 512 + (size_t)lane * 8 + i;
          out[o] = h;
          out[o + (size_t)ntq * 512] = (__bf16)(v - (float)h);
        } else {
          out[e] = (__bf16)v;
        }
      }
    }
  } else {
    for (int vi = 0; vi < a.n_vecs; ++vi) {
      const int n = a.v[vi].n, nw = a.v[vi].n_out > n ? a.v[vi].n_out : n, npad = ((nw + 31) / 32) * 32;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += gridDim.x * blockDim.x)
        a.v[vi].out[i] = i < n ? a.v[vi].v[i] : 0.f;
    }
  }
}

// ---- NormalizedMSELoss (losses.py:66-94) ----------------------------------------------------------------
__global__ void nmse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                            const float* __restrict__ inv_var, int inv_var_full, const float* __restrict__ lat_w, int num_lon,
                            int nodes, int channels, size_t total, float scale, float* __restrict__ loss) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / channels;
    const int ch = (int)(i - row * channels);
    const int n = (int)(row % nodes);
    float d = pred[i] - target[i];
    d = d * d;
    if (inv_var) d *= inv_var_full ? inv_var[i] : inv_var[ch];
    acc += d * lat_w[n / num_lon];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float part[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) part[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i];
    __hip_atomic_fetch_add(loss, s * scale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

thread_local char g_err[512] = "";

}  // namespace

namespace gw {
unsigned long long* g_dbg = nullptr;
int g_dbg_cap = 0;
int g_dbg_kind = -1;  // which launch family to stamp: 0 mlp, 1 edge, 2 node, 3 project

int set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return GW_E_LAUNCH;
  }
  return GW_OK;
}

#ifdef GW_TUNING
int env_int(const char* name, int fallback) {
  const char* e = getenv(name);
  return e ? atoi(e) : fallback;
}
#endif
}  // namespace gw

namespace {

int fail(int code, const char* msg) { return gw::set_error(code, msg); }

// 16-bit matrix-core modes: GW_DTYPE_BF16 (gw_bf16.hip, + the resident kernels) and GW_DTYPE_BF16X3 (gw_split.hip: split operands,
// fp32 rows everywhere - none of the bf16 mode's 16-bit table formats)
bool is16(int dt) { return dt == GW_DTYPE_BF16 || dt == GW_DTYPE_BF16X3; }
int launch16(int dt, int kind, ChainArgs& a, int k_in, int hidden, int n_out, int grid_y, void* stream) {
  return dt == GW_DTYPE_BF16X3 ? gw::chainx3_launch(kind, a, k_in, hidden, n_out, grid_y, stream)
                               : gw::chain16_launch(kind, a, k_in, hidden, n_out, grid_y, stream);
}

int g_stagger_override = -1;  // GW_STAGGER env (tuning): -1 = automatic

template <typename K>
int launch_chain(K kernel, ChainArgs& a, void* stream, int grid_y = 1, int kind = 0) {
  if (g_dbg != nullptr && kind == g_dbg_kind) {
    a.dbg = g_dbg;
    a.dbg_cap = g_dbg_cap;
  }
  {
    static bool env_read = false;
    if (!env_read) {
      g_stagger_override = GW_TUNE("GW_STAGGER", -1);
      env_read = true;
    }
    // number of 256-K passes this launch runs per tile -> about half a tile of delay (8k-cycle sleeps)
    int passes = 1 + a.n_mid;
    for (int i = 0; i < 3; ++i) passes += (a.seg_k[i] > 0 && !a.seg_proj[i]) ? 1 : 0;
    a.stagger = g_stagger_override >= 0 ? g_stagger_override * passes : 2 * passes + 2;
    if ((a.n_cols + kColsPerWG - 1) / kColsPerWG <= 256) a.stagger = 0;
  }
  static DeviceOnce once;  // per template instantiation and device
  if (once.first()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  const int grid = (a.n_cols + kColsPerWG - 1) / kColsPerWG;
  hipLaunchKernelGGL(kernel, dim3(grid, grid_y), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
  return check_launch("chain_kernel launch");
}

bool bad_layers(const gw_mlp_weights* w) { return w->n_mid < 1 || !w->w_mid || !w->b_mid; }

void fill_weights(ChainArgs& a, const gw_mlp_weights* w) {
  for (int i = 0; i < 3; ++i) a.w1[i] = w->w1[i];
  a.b1 = w->b1;
  a.w_mid = w->w_mid;
  a.b_mid = w->b_mid;
  a.w_out = w->w_out;
  a.b_out = w->b_out;
  a.gamma = w->ln_gamma;
  a.beta = w->ln_beta;
  a.n_mid = w->n_mid;
  a.ln_width = (w->ln_width > 0 && w->ln_width < w->n_out) ? w->ln_width : w->n_out;
}

void fill_operand(ChainArgs& a, int i, const gw_operand* op) {
  a.seg_ptr[i] = op->ptr;
  a.seg_idx[i] = op->index;
  a.seg_rows_pb[i] = op->rows_per_batch;
  a.seg_ld[i] = op->ld;
  a.seg_k[i] = op->k;
  a.seg_proj[i] = op->projected;
  a.seg_half[i] = op->layout == GW_LAYOUT_ROWS_F16;
  a.seg_bf16k[i] = op->layout == GW_LAYOUT_ROWS_BF16K;
}
// the aggregate operand of the node updates: fp32 rows, or (bf16 weights) bf16 rows in K order
bool bad_agg(const gw_operand* op, bool bf16_weights) {
  if (op->k != 256 || !op->ptr || op->projected) return true;
  if (op->layout == GW_LAYOUT_ROWS_F32) return op->ld % 4 != 0;
  return !(bf16_weights && op->layout == GW_LAYOUT_ROWS_BF16K && op->ld % 8 == 0 && !op->index);
}

void fill_residual(ChainArgs& a, const gw_operand* op) {
  a.res_ptr = op->ptr;
  a.res_idx = op->index;
  a.res_rows_pb = op->rows_per_batch;
  a.res_ld = op->ld;
}

// a 256-wide row table in fp32 rows - or, with allow_half (projected operands of the bf16 node-side kernels), fp16 product rows
bool bad256(const gw_operand* op, bool allow_half = false) {
  if (op->k != 256 || op->ld % 4 != 0 || !op->ptr) return true;
  if (op->layout == GW_LAYOUT_ROWS_F32) return false;
  return !(allow_half && op->layout == GW_LAYOUT_ROWS_F16 && op->projected);
}

}  // namespace

extern "C" {

int gw_version(void) { return GW_ABI_VERSION; }

int gw_debug_timestamps(void* buffer, int capacity_workgroups, int kind) {
  g_dbg = (unsigned long long*)buffer;
  g_dbg_cap = capacity_workgroups;
  g_dbg_kind = kind;
  return GW_OK;
}
const char* gw_last_error(void) { return g_err; }

// K-steps (4 input features each) of a packed slice: the layer-1 kernels exist for 4, 28 and 64+ steps (k <= 16, k <= 112, wider)
// and stream exactly that many from the pack, so a narrower slice is zero-padded to its variant's step count (a 78-wide
// input - GraphCast, graphcast/model.py:21 - used to get 20 steps and the 28-step kernel read 32 KB past its end).
static int packed_steps_f32(int kseg) { return kseg <= 16 ? 4 : (kseg <= 112 ? 28 : ((kseg + 15) / 16) * 4); }

size_t gw_packed_floats(int n_out, int k_lo, int k_hi) {
  const int kseg = k_hi - k_lo;
  const int nsteps = packed_steps_f32(kseg);
  const int nt = (n_out + 15) / 16;
  const int nt4 = (nt + 3) / 4;
  return (size_t)nsteps * nt4 * 256;
}

int gw_pack_linear(const float* w, int n_out, int k_total, int k_lo, int k_hi, float* out, void* stream) {
  if (!w || !out || n_out <= 0 || k_lo < 0 || k_hi <= k_lo || k_hi > k_total) return fail(GW_E_BADARG, "gw_pack_linear: bad arguments");
  const int kseg = k_hi - k_lo;
  const int nsteps = packed_steps_f32(kseg);
  const int nt4 = (((n_out + 15) / 16) + 3) / 4;
  const size_t total = (size_t)nsteps * nt4 * 256;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_linear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, n_out, k_total, k_lo, kseg, nt4, nsteps, out);
  return check_launch("pack_linear_kernel launch");
}

int gw_padded_n(int n) { return ((n + 31) / 32) * 32; }

int gw_pack_many(int32_t weight_dtype, int32_t n_mats, const gw_pack_item* mats, int32_t n_vecs, const gw_pad_item* vecs,
                 void* stream) {
  if (n_mats < 0 || n_vecs < 0 || n_mats > GW_PACK_MAX_ITEMS || n_vecs > GW_PACK_MAX_ITEMS || (n_mats > 0 && !mats) ||
      (n_vecs > 0 && !vecs) || (weight_dtype != GW_DTYPE_F32 && !is16(weight_dtype)))
    return fail(GW_E_BADARG, "gw_pack_many: bad arguments");
  if (n_mats == 0 && n_vecs == 0) return GW_OK;
  PackManyArgs a;
  memset(&a, 0, sizeof(a));
  a.n_mats = n_mats;
  a.n_vecs = n_vecs;
  a.bf16 = weight_dtype == GW_DTYPE_BF16X3 ? 2 : (weight_dtype == GW_DTYPE_BF16 ? 1 : 0);
  for (int i = 0; i < n_mats; ++i) {
    const gw_pack_item& m = mats[i];
    if (!m.w || !m.out || m.n_out <= 0 || m.kseg <= 0) return fail(GW_E_BADARG, "gw_pack_many: bad matrix item");
    a.m[i] = m;
    if (m.rows != 0 && m.rows < m.n_out) return fail(GW_E_BADARG, "gw_pack_many: rows < n_out");
    const int nt = ((m.rows > m.n_out ? m.rows : m.n_out) + 15) / 16;
    if (a.bf16) {  // (gw_packed_bytes_bf16)
      a.nsteps[i] = m.kseg <= 32 ? 1 : (m.kseg <= 128 ? 4 : (m.kseg + 31) / 32);
      a.ntq[i] = (nt + 3) / 4 * 4;
    } else {       // (gw_packed_floats)
      a.nsteps[i] = packed_steps_f32(m.kseg);
      a.ntq[i] = (nt + 3) / 4;
    }
  }
  for (int i = 0; i < n_vecs; ++i) {
    if (!vecs[i].v || !vecs[i].out || vecs[i].n <= 0) return fail(GW_E_BADARG, "gw_pack_many: bad vector item");
    a.v[i] = vecs[i];
  }
  hipLaunchKernelGGL(pack_many_kernel, dim3(64, n_mats + (n_vecs > 0 ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("pack_many_kernel launch");
}

namespace {
int fill_bwd_chain(BwdChainArgs& a, int64_t n_rows, const float* d, int32_t d_ld, int32_t n_chain, const void* const* chain_w,
                   const float* const* chain_mask, float* const* chain_out, int32_t n_fan, const void* const* fan_w, float* const* fan_out) {
  if (n_rows < 0 || !d || d_ld < 256 || d_ld % 4 != 0 || n_chain < 0 || n_chain > 2 || n_fan < 0 || n_fan > 3 || n_chain + n_fan < 1 ||
      (n_chain > 0 && (!chain_w || !chain_mask || !chain_out)) || (n_fan > 0 && (!fan_w || !fan_out)))
    return fail(GW_E_BADARG, "gw_mlp_chain_backward: bad arguments");
  memset(&a, 0, sizeof(a));
  a.d = d;
  a.d_ld = d_ld;
  a.n_rows = n_rows;
  a.n_chain = n_chain;
  a.n_fan = n_fan;
  for (int i = 0; i < n_chain; ++i) {
    if (!chain_w[i] || !chain_mask[i] || !chain_out[i]) return fail(GW_E_BADARG, "gw_mlp_chain_backward: NULL chain item");
    a.w[i] = chain_w[i];
    a.mask[i] = chain_mask[i];
    a.out[i] = chain_out[i];
  }
  for (int i = 0; i < n_fan; ++i) {
    if (!fan_w[i] || !fan_out[i]) return fail(GW_E_BADARG, "gw_mlp_chain_backward: NULL fan item");
    a.w[n_chain + i] = fan_w[i];
    a.out[n_chain + i] = fan_out[i];
  }
  if ((n_rows + kColsPerWG - 1) / kColsPerWG > 0x7fffffffLL) return fail(GW_E_BADARG, "gw_mlp_chain_backward: too many rows");
  return GW_OK;
}
}  // namespace

int gw_mlp_chain_backward(int64_t n_rows, const float* d, int32_t d_ld, int32_t n_chain, const float* const* chain_w,
                          const float* const* chain_mask, float* const* chain_out, int32_t n_fan, const float* const* fan_w,
                          float* const* fan_out, void* stream) {
  BwdChainArgs a;
  if (int rc = fill_bwd_chain(a, n_rows, d, d_ld, n_chain, (const void* const*)chain_w, chain_mask, chain_out, n_fan, (const void* const*)fan_w,
                              fan_out))
    return rc;
  if (n_rows == 0) return GW_OK;
  return bwd_chain_launch(a, stream);
}

int gw_mlp_ln_chain_backward(int32_t weight_dtype, int64_t n_rows, const float* dn, int32_t dn_ld, const int32_t* dn_idx,
                             int32_t dn_idx_n, int32_t dn_table_rows_pb, const float* dn_add, int32_t dn_add_ld, const float* y,
                             const float* gamma, float* dgamma, float* dbeta, float* dy, int32_t n_chain, const void* const* chain_w,
                             const float* const* chain_mask, float* const* chain_out, float* dz_colsum, int32_t n_fan,
                             const void* const* fan_w, float* const* fan_out, const float* const* fan_add, int32_t fan_add_ld,
                             uint32_t fan_add_dn_mask, void* stream) {
  const bool ln = y != nullptr;
  if (ln && (!gamma || !dgamma || !dbeta || !dy)) return fail(GW_E_BADARG, "gw_mlp_ln_chain_backward: NULL LayerNorm argument");
  if (weight_dtype != GW_DTYPE_BF16X3 && weight_dtype != GW_DTYPE_F32)
    return fail(GW_E_UNSUPPORTED, "gw_mlp_ln_chain_backward: fp32 or split (GW_DTYPE_BF16X3) streams");
  if (dz_colsum && n_chain < 1) return fail(GW_E_BADARG, "gw_mlp_ln_chain_backward: dz_colsum needs a chain product");
  if (fan_add && (fan_add_ld < 256 || fan_add_ld % 4 != 0)) return fail(GW_E_BADARG, "gw_mlp_ln_chain_backward: bad fan_add_ld");
  if ((dn_idx || dn_add) && !ln) return fail(GW_E_UNSUPPORTED, "gw_mlp_ln_chain_backward: a gathered input gradient needs the LayerNorm in front");
  if (dn_add && !dn_idx) return fail(GW_E_BADARG, "gw_mlp_ln_chain_backward: dn_add goes with dn_idx");
  if (dn_idx && (dn_idx_n <= 0 || dn_table_rows_pb <= 0 || n_rows % dn_idx_n != 0 || (dn_add && (dn_add_ld < 256 || dn_add_ld % 4 != 0))))
    return fail(GW_E_BADARG, "gw_mlp_ln_chain_backward: bad gather arguments");
  if (fan_add_dn_mask >> (n_fan > 0 ? n_fan : 0)) return fail(GW_E_BADARG, "gw_mlp_ln_chain_backward: fan_add_dn_mask names a fan item that does not exist");
  BwdChainArgs a;
  if (int rc = fill_bwd_chain(a, n_rows, dn, dn_ld, n_chain, chain_w, chain_mask, chain_out, n_fan, fan_w, fan_out)) return rc;
  if (n_rows == 0) return GW_OK;
  if (ln) {
    a.ln_y = y;
    a.ln_gamma = gamma;
    a.ln_dgamma = dgamma;
    a.ln_dbeta = dbeta;
    a.ln_dy = dy;
  }
  a.d_idx = dn_idx;
  a.d_idx_n = dn_idx_n;
  a.d_tab_rows_pb = dn_table_rows_pb;
  a.d_add = dn_add;
  a.d_add_ld = dn_add_ld;
  a.colsum = dz_colsum;
  a.add_ld = fan_add_ld;
  for (int i = 0; fan_add && i < n_fan; ++i) a.add[n_chain + i] = fan_add[i];
  a.add_d_mask = fan_add_dn_mask << n_chain;
  return weight_dtype == GW_DTYPE_BF16X3 ? bwd_chainx3_launch(a, stream) : bwd_chain_launch(a, stream);
}

int gw_mlp_chain_backward_bf16x3(int64_t n_rows, const float* d, int32_t d_ld, int32_t n_chain, const void* const* chain_w,
                                 const float* const* chain_mask, float* const* chain_out, int32_t n_fan, const void* const* fan_w,
                                 float* const* fan_out, void* stream) {
  BwdChainArgs a;
  if (int rc = fill_bwd_chain(a, n_rows, d, d_ld, n_chain, chain_w, chain_mask, chain_out, n_fan, fan_w, fan_out)) return rc;
  if (n_rows == 0) return GW_OK;
  return bwd_chainx3_launch(a, stream);
}

int gw_pad_vector(const float* v, int n, float* out, void* stream) {
  if (!v || !out || n <= 0) return fail(GW_E_BADARG, "gw_pad_vector: bad arguments");
  const int npad = gw_padded_n(n);
  hipLaunchKernelGGL(pad_vector_kernel, dim3((npad + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, n, npad, out);
  return check_launch("pad_vector_kernel launch");
}

namespace {
int fill_save(ChainArgs& a, const gw_activation_save* save, const gw_mlp_weights* w, const char* who) {
  if (!save) return GW_OK;
  if (w->weight_dtype == GW_DTYPE_BF16)
    return fail(GW_E_UNSUPPORTED, "activation saving (training) is implemented for fp32 and bf16x3 weights (the saves are fp32 rows)");
  if (!save->hidden || save->hidden_ld < w->hidden || save->hidden_ld % 4 != 0 || (w->ln_gamma && !save->pre_norm)) {
    snprintf(g_err, sizeof(g_err), "%s: bad gw_activation_save", who);
    return GW_E_BADARG;
  }
  a.save_h = save->hidden;
  a.save_stride = save->hidden_stride;
  a.save_ld = save->hidden_ld;
  a.save_y = save->pre_norm;
  return GW_OK;
}
}  // namespace

int gw_mlp_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_mlp_weights* w,
                   const gw_operand* residual, float* out, int32_t out_ld, const gw_activation_save* save, void* stream) {
  if (!x || !w || !out || n_rows < 0 || rows_per_batch <= 0) return fail(GW_E_BADARG, "gw_mlp_forward: bad arguments");
  if (n_rows == 0) return GW_OK;
  if (n_rows >= (int64_t)1 << 31) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: more than 2^31-1 rows");
  if (x->k <= 0 || !w->w1[0] || !w->w_out || !w->b1 || !w->b_out)
    return fail(GW_E_BADARG, "gw_mlp_forward: missing weights / empty operand");
  if (x->layout != GW_LAYOUT_ROWS_F32 || (residual && residual->layout != GW_LAYOUT_ROWS_F32))
    return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: operands are fp32 rows");
  if (bad_layers(w)) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: at least 2 hidden layers (n_mid >= 1) are required");
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  a.n_cols = (int)n_rows;
  a.cols_per_batch = rows_per_batch;
  fill_operand(a, 0, x);
  fill_weights(a, w);
  if (residual) {
    a.res_ptr = residual->ptr;
    a.res_idx = residual->index;
    a.res_rows_pb = residual->rows_per_batch;
    a.res_ld = residual->ld;
  }
  a.out = out;
  a.out_ld = out_ld;
  a.out_cols = w->n_out;
  if (int rc = fill_save(a, save, w, "gw_mlp_forward")) return rc;
  if (is16(w->weight_dtype)) {
    if (residual && w->n_out == 256 && (residual->ld % 4 != 0)) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: residual ld must be a multiple of 4");
    if (w->n_out == 256 && out_ld % 4 != 0) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: out_ld must be a multiple of 4");
    if (x->k == 256 && x->ld % 4 != 0) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: input ld must be a multiple of 4");
    if (x->k > 128 && x->k != 256) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: input width must be <=128 or ==256");
    if (w->weight_dtype == GW_DTYPE_BF16 && w->ln_gamma && (w->n_out != 256 || (w->ln_width > 0 && w->ln_width != w->n_out)))
      return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: LayerNorm over fewer than 256 features is implemented for float32 and bf16x3 weights");
    return launch16(w->weight_dtype, 0, a, x->k, w->hidden, w->n_out, 1, stream);
  }
  if (w->hidden == 256 && w->n_out == 256) {
    if (residual && (residual->ld % 4 != 0)) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: residual ld must be a multiple of 4");
    if (out_ld % 4 != 0) return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: out_ld must be a multiple of 4");
    if (x->k <= 16) return launch_chain(chain_kernel<4, false, 1, 16, 16, EPI_ROWS>, a, stream);
    if (x->k <= 112) return launch_chain(chain_kernel<28, false, 1, 16, 16, EPI_ROWS>, a, stream);
    if (x->k == 256 && x->ld % 4 == 0) return launch_chain(chain_kernel<64, true, 1, 16, 16, EPI_ROWS>, a, stream);
    return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: input width must be <=112 or ==256 for hidden 256");
  }
  if (w->hidden == 256 && w->n_out <= 80 && x->k == 256 && x->ld % 4 == 0) {
    // head with 256 hidden units (GraphCast wrapper: hidden_dim_decoder = hidden_dim, graphcast/model.py:100-114)
    return launch_chain(chain_kernel<64, true, 1, 16, 5, EPI_DEC>, a, stream);
  }
  if (w->hidden == 128 && w->n_out <= 80 && x->k == 256 && x->ld % 4 == 0) {
    return launch_chain(chain_kernel<64, true, 1, 8, 5, EPI_DEC>, a, stream);
  }
  return fail(GW_E_UNSUPPORTED, "gw_mlp_forward: unsupported (hidden, n_out, k) combination");
}

int gw_mlp_post_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_mlp_weights* w, float* out,
                        int32_t out_ld, int32_t n_post, const float* const* post_w, void* const* post_out, int32_t post_layout,
                        void* stream) {
  if (!x || !w || n_rows < 0 || rows_per_batch <= 0 || n_post <= 0 || n_post > 4 || !post_w || !post_out)
    return fail(GW_E_BADARG, "gw_mlp_post_forward: bad arguments");
  if (n_rows == 0) return GW_OK;
  if (n_rows >= (int64_t)1 << 31) return fail(GW_E_UNSUPPORTED, "gw_mlp_post_forward: more than 2^31-1 rows");
  if (x->k <= 0 || !w->w1[0] || !w->w_out || !w->b1 || !w->b_out) return fail(GW_E_BADARG, "gw_mlp_post_forward: missing weights / empty operand");
  if (x->layout != GW_LAYOUT_ROWS_F32) return fail(GW_E_UNSUPPORTED, "gw_mlp_post_forward: x is fp32 rows");
  if (bad_layers(w)) return fail(GW_E_UNSUPPORTED, "gw_mlp_post_forward: at least 2 hidden layers (n_mid >= 1) are required");
  if (!is16(w->weight_dtype) || w->hidden != 256 || w->n_out != 256 || !w->ln_gamma || (w->ln_width > 0 && w->ln_width != 256))
    return fail(GW_E_UNSUPPORTED, "gw_mlp_post_forward: bf16 / bf16x3 weights, hidden 256, 256 outputs with LayerNorm");
  if (post_layout != GW_LAYOUT_ROWS_F32 && post_layout != GW_LAYOUT_ROWS_F16) return fail(GW_E_BADARG, "gw_mlp_post_forward: bad post_layout");
  if (post_layout == GW_LAYOUT_ROWS_F16 && w->weight_dtype != GW_DTYPE_BF16)
    return fail(GW_E_UNSUPPORTED, "gw_mlp_post_forward: fp16 products come with bf16 weights");
  if (out && out_ld % 4 != 0) return fail(GW_E_UNSUPPORTED, "gw_mlp_post_forward: out_ld must be a multiple of 4");
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  a.n_cols = (int)n_rows;
  a.cols_per_batch = rows_per_batch;
  fill_operand(a, 0, x);
  fill_weights(a, w);
  a.out = out;
  a.out_ld = out ? out_ld : 256;
  a.out_cols = 256;
  for (int i = 0; i < n_post; ++i) {
    if (!post_w[i] || !post_out[i]) return fail(GW_E_BADARG, "gw_mlp_post_forward: null post slice / output");
    a.proj_w[i] = post_w[i];
    a.proj_out[i] = (float*)post_out[i];
  }
  a.n_post = n_post;
  a.proj_half = post_layout == GW_LAYOUT_ROWS_F16;
  return launch16(w->weight_dtype, 5, a, x->k, w->hidden, w->n_out, 1, stream);
}

size_t gw_edge_update_workspace_bytes(int32_t batch, int32_t n_edges, const gw_operand* x_src, const gw_operand* x_dst,
                                      const gw_operand* e_in, const gw_mlp_weights* w, int32_t flags) {
  if (batch <= 0 || n_edges <= 0 || !x_src || !x_dst || !e_in || !w) return 0;
  const bool det = (flags & GW_EDGE_DETERMINISTIC) != 0;
  if (gw::edge16_eligible(x_src, x_dst, e_in, w)) return gw::edge16_workspace_needed(batch, n_edges, e_in, det);
  if (det && w->weight_dtype == GW_DTYPE_F32 && gw::edge_fast_eligible(x_src, x_dst, e_in, w)) return gw::edge_fast_carry_bytes(batch, n_edges);
  if (det && is16(w->weight_dtype)) return gw::edge_fast_carry_bytes(batch, n_edges);  // the general bf16 / bf16x3 kernel (e.g. the encoder's raw node operand)
  return 0;
}

size_t gw_edge_tiles_bytes(int32_t batch, int32_t n_edges) {
  if (batch <= 0 || n_edges <= 0) return 0;
  return gw::edge16_workspace_bytes(batch, n_edges);  // a tile buffer and the layer-1 workspace have the same shape
}

int gw_edge_rows_to_tiles(int32_t batch, int32_t n_edges, const float* rows, int32_t rows_per_batch, int32_t ld, void* tiles,
                          void* stream) {
  if (batch <= 0 || n_edges <= 0 || !rows || !tiles || ld < 256 || ld % 4 != 0 || rows_per_batch < 0)
    return fail(GW_E_BADARG, "gw_edge_rows_to_tiles: bad arguments");
  return gw::edge16_rows_to_tiles(batch, n_edges, rows, rows_per_batch, ld, tiles, stream);
}

int gw_edge_update_forward(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst,
                           const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in,
                           const gw_operand* e_res, const gw_mlp_weights* w, void* e_out_any, int32_t e_out_layout, float* agg,
                           int32_t n_dst, const gw_activation_save* save, void* workspace, size_t workspace_bytes, int32_t flags,
                           void* stream) {
  if (batch <= 0 || n_edges < 0 || n_dst <= 0) return fail(GW_E_BADARG, "gw_edge_update_forward: bad arguments");
  if (n_edges == 0) return GW_OK;  // nothing to add: agg stays as the caller zeroed it
  if (!src || !dst || !x_src || !x_dst || !e_in || !e_res || !w || !agg) return fail(GW_E_BADARG, "gw_edge_update_forward: bad arguments");
  if ((int64_t)batch * n_edges >= (int64_t)1 << 31 || (int64_t)batch * n_dst >= (int64_t)1 << 31)
    return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: batch*edges exceeds int32");
  if (w->hidden != 256 || w->n_out != 256 || (w->ln_gamma == nullptr) != (w->ln_beta == nullptr) || !w->b1 || !w->w_out || !w->b_out)
    return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: only hidden=256, out=256 is implemented");
  // norm_type=None (graph_net_block.py:50-59 allows it): the general kernels run without the LayerNorm epilogue
  if (bad_layers(w)) return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: at least 2 hidden layers (n_mid >= 1) are required");
  const gw_operand* ops[3] = {x_src, x_dst, e_in};
  for (int i = 0; i < 3; ++i) {
    if (ops[i]->k == 0) continue;
    if (ops[i]->k != 256 || ops[i]->ld % 4 != 0 || !ops[i]->ptr || (!ops[i]->projected && !w->w1[i]))  // (layouts: checked below)
      return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: operands must be 256 wide (or k=0 for zeros)");
  }
  // e_res->k == 0: no residual - for callers that want only the aggregate and have added the segment sums of their (batch-
  // shared) e into agg beforehand: sum(LN(.) + e) = sum(LN(.)) + sum(e).  bf16 path with resident weights only (edge16_launch).
  const bool no_res = e_res->k == 0;
  const bool x3 = w->weight_dtype == GW_DTYPE_BF16X3;
  if (no_res && (e_out_any != nullptr || save || !(x3 || (gw::edge16_eligible(x_src, x_dst, e_in, w) && !(flags & GW_EDGE_DETERMINISTIC)))))
    return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: an edge update without residual (e_res.k == 0) is implemented for the bf16 "
                                  "path with resident weights (atomics mode) and for bf16x3 weights, without e_out or activation saving");
  if (!no_res && (e_res->k != 256 || e_res->ld % 4 != 0 || !e_res->ptr))
    return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: e_res (residual edge features) must be 256 wide");
  // edge tiles (bf16) are a format of the bf16 path with resident weights only
  const bool tiles_in = (e_in->k > 0 && e_in->layout == GW_LAYOUT_EDGE_TILES_BF16) || e_res->layout == GW_LAYOUT_EDGE_TILES_BF16;
  const bool tiles_out = e_out_any != nullptr && e_out_layout == GW_LAYOUT_EDGE_TILES_BF16;
  if (e_out_any != nullptr && e_out_layout != GW_LAYOUT_ROWS_F32 && e_out_layout != GW_LAYOUT_EDGE_TILES_BF16)
    return fail(GW_E_BADARG, "gw_edge_update_forward: bad e_out_layout");
  if ((e_in->k > 0 && e_in->layout != GW_LAYOUT_ROWS_F32 && e_in->layout != GW_LAYOUT_EDGE_TILES_BF16) ||
      (!no_res && e_res->layout != GW_LAYOUT_ROWS_F32 && e_res->layout != GW_LAYOUT_EDGE_TILES_BF16))
    return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: edge operands are fp32 rows or bf16 edge tiles");
  const bool half_nodes = x_src->layout == GW_LAYOUT_ROWS_F16 || x_dst->layout == GW_LAYOUT_ROWS_F16;
  if ((x_src->layout != GW_LAYOUT_ROWS_F32 && x_src->layout != GW_LAYOUT_ROWS_F16) ||
      (x_dst->layout != GW_LAYOUT_ROWS_F32 && x_dst->layout != GW_LAYOUT_ROWS_F16) ||
      (half_nodes && (save || !gw::edge16_eligible(x_src, x_dst, e_in, w))))
    return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: node operands must be fp32 rows (or, projected, for the bf16 path with "
                                  "resident weights: fp16 product rows)");
  float* e_out = tiles_out ? nullptr : (float*)e_out_any;
  const bool det = (flags & GW_EDGE_DETERMINISTIC) != 0;
  const bool seg = (flags & GW_EDGE_SEGMENT_TILES) != 0;
  if ((flags & (GW_EDGE_AGG_BF16K | GW_EDGE_SEGMENT_SPLIT)) && !seg)
    return fail(GW_E_BADARG, "gw_edge_update_forward: GW_EDGE_AGG_BF16K / GW_EDGE_SEGMENT_SPLIT come with GW_EDGE_SEGMENT_TILES");
  if (seg) {
    const size_t ws_seg = gw::edge16_eligible(x_src, x_dst, e_in, w) ? gw::edge16_workspace_needed(batch, n_edges, e_in, false) : 0;
    if (det || save || n_edges % 64 != 0 || !gw::edge16_eligible(x_src, x_dst, e_in, w) || (e_out_any && !tiles_out) ||
        (!no_res && e_res->layout != GW_LAYOUT_EDGE_TILES_BF16) || (ws_seg > 0 && (!workspace || workspace_bytes < ws_seg)))
      return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: segment-aligned tiles (GW_EDGE_SEGMENT_TILES) are implemented for the bf16 "
                                    "path with resident weights: projected node operands, residual none or bf16 edge tiles, e' none "
                                    "or bf16 edge tiles, atomics mode, n_edges a multiple of 64, the workspace of "
                                    "gw_edge_update_workspace_bytes");
    return gw::edge16_launch(batch, n_edges, src, dst, x_src, x_dst, e_in, e_res, w, nullptr, tiles_out ? e_out_any : nullptr, agg, n_dst,
                             workspace, flags, stream);
  }
  const size_t ws16 = gw::edge16_workspace_needed(batch, n_edges, e_in, det);
  if (det && save) return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: deterministic segment sums are an inference option");
  if (tiles_in || tiles_out || (no_res && !x3) || half_nodes) {
    if (save || (ws16 > 0 && (!workspace || workspace_bytes < ws16)) || !gw::edge16_eligible(x_src, x_dst, e_in, w))
      return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: bf16 edge tiles need bf16 weights, one middle layer, projected node "
                                    "operands, no activation saving and the workspace of gw_edge_update_workspace_bytes");
    return gw::edge16_launch(batch, n_edges, src, dst, x_src, x_dst, e_in, e_res, w, e_out, tiles_out ? e_out_any : nullptr, agg,
                             n_dst, workspace, det ? GW_EDGE_DETERMINISTIC : 0, stream);
  }
  if (w->weight_dtype == GW_DTYPE_F32 && (!save || w->n_mid == 1) && gw::edge_fast_eligible(x_src, x_dst, e_in, w)) {
    if (save && (!save->hidden || !save->pre_norm || save->hidden_ld < 256 || save->hidden_ld % 4 != 0))
      return fail(GW_E_BADARG, "gw_edge_update_forward: bad gw_activation_save");
    if (det && (!workspace || workspace_bytes < gw::edge_fast_carry_bytes(batch, n_edges)))
      return fail(GW_E_BADARG, "gw_edge_update_forward: deterministic mode needs the workspace of gw_edge_update_workspace_bytes");
    return gw::edge_fast_launch(batch, n_edges, src, dst, x_src, x_dst, e_in, e_res, w, e_out, agg, n_dst, save,
                                det ? (float*)workspace : nullptr, stream);
  }
  if (!save && gw::edge16_eligible(x_src, x_dst, e_in, w) && (ws16 == 0 || (workspace && workspace_bytes >= ws16)))
    return gw::edge16_launch(batch, n_edges, src, dst, x_src, x_dst, e_in, e_res, w, e_out, nullptr, agg, n_dst, workspace,
                             det ? GW_EDGE_DETERMINISTIC : 0, stream);
  if (det && (!is16(w->weight_dtype) || !workspace || workspace_bytes < gw::edge_fast_carry_bytes(batch, n_edges)))
    return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: deterministic segment sums exist on the fast fp32 edge kernel (at most one "
                                  "raw operand, native 256 widths) and on the bf16 kernels, and need their workspace");
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  a.n_cols = batch * n_edges;
  a.cols_per_batch = n_edges;
  fill_operand(a, 0, x_src);
  a.seg_idx[0] = src;
  fill_operand(a, 1, x_dst);
  a.seg_idx[1] = dst;
  fill_operand(a, 2, e_in);
  fill_weights(a, w);
  if (!no_res) fill_residual(a, e_res);
  a.out = e_out;
  a.out_ld = 256;
  a.out_cols = 256;
  a.agg = agg;
  a.agg_idx = dst;
  a.agg_rows_pb = n_dst;
  if (int rc = fill_save(a, save, w, "gw_edge_update_forward")) return rc;
  if (is16(w->weight_dtype)) {
    if (w->weight_dtype == GW_DTYPE_BF16 && w->ln_gamma && a.ln_width != 256)
      return fail(GW_E_UNSUPPORTED, "gw_edge_update_forward: LayerNorm over fewer than 256 features needs float32 or bf16x3 weights");
    if (det) a.carry = (float*)workspace;
    if (int rc = launch16(w->weight_dtype, 1, a, 256, 256, 256, 1, stream)) return rc;
    return det ? gw::segment_fixup_launch(((int64_t)a.n_cols + 63) / 64, a.carry, agg, stream) : GW_OK;
  }
  return launch_chain(chain_kernel<64, true, 3, 16, 16, EPI_EDGE>, a, stream, 1, 1);
}

int gw_node_update_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_operand* x_res,
                           const gw_operand* agg, const gw_mlp_weights* w, float* x_out, int32_t out_ld,
                           const gw_activation_save* save, int32_t n_post, const float* const* post_w, void* const* post_out,
                           int32_t post_layout, float* zero_rows, void* stream) {
  if (!x || !agg || !w || !x_out || n_rows < 0 || rows_per_batch <= 0) return fail(GW_E_BADARG, "gw_node_update_forward: bad arguments");
  if (n_rows == 0) return GW_OK;
  if (n_rows >= (int64_t)1 << 31) return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: more than 2^31-1 rows");
  if (w->hidden != 256 || w->n_out != 256 || !w->b1 || !w->w_out || !w->b_out)
    return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: only hidden=256, out=256");
  if (bad_layers(w)) return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: at least 2 hidden layers (n_mid >= 1) are required");
  if (bad_agg(agg, w->weight_dtype == GW_DTYPE_BF16) || !w->w1[1])
    return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: agg must be 256 wide (raw fp32 rows; bf16 rows in K order with bf16 weights)");
  if (agg->layout == GW_LAYOUT_ROWS_BF16K && save) return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: bf16 aggregates are an inference format");
  if (x->k != 0 && (bad256(x, w->weight_dtype == GW_DTYPE_BF16) || (!x->projected && !w->w1[0])))
    return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: x must be 256 wide (fp32 rows; fp16 product rows with bf16 weights) or zeros");
  if (x_res && x_res->k != 0 && bad256(x_res)) return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: x_res must be 256 wide");
  if (out_ld % 4 != 0) return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: out_ld must be a multiple of 4");
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  a.n_cols = (int)n_rows;
  a.cols_per_batch = rows_per_batch;
  fill_operand(a, 0, x);
  fill_operand(a, 1, agg);
  fill_weights(a, w);
  if (x_res && x_res->k != 0) fill_residual(a, x_res);
  a.out = x_out;
  a.out_ld = out_ld;
  a.out_cols = 256;
  if (int rc = fill_save(a, save, w, "gw_node_update_forward")) return rc;
  if (n_post < 0 || n_post > 4 || (n_post > 0 && (!post_w || !post_out))) return fail(GW_E_BADARG, "gw_node_update_forward: bad post products");
  if ((n_post > 0 || zero_rows) && (save || out_ld != 256)) return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: post products need out_ld 256, no activation saving");
  if (zero_rows && n_post == 0) return fail(GW_E_BADARG, "gw_node_update_forward: zero_rows comes with post products");
  for (int i = 0; i < n_post; ++i) {
    if (!post_w[i] || !post_out[i]) return fail(GW_E_BADARG, "gw_node_update_forward: null post slice / output");
    a.proj_w[i] = post_w[i];
    a.proj_out[i] = (float*)post_out[i];
  }
  a.n_post = n_post;
  if (post_layout != GW_LAYOUT_ROWS_F32 && post_layout != GW_LAYOUT_ROWS_F16) return fail(GW_E_BADARG, "gw_node_update_forward: bad post_layout");
  if (post_layout == GW_LAYOUT_ROWS_F16 && (n_post == 0 || w->weight_dtype != GW_DTYPE_BF16))
    return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: fp16 post products come with bf16 weights");
  a.proj_half = post_layout == GW_LAYOUT_ROWS_F16;
  a.zero_rows = zero_rows;
  // mesh-sized launches (at most one round of 48-column workgroups): row tiles split over the waves of a column group
  // (gw_noders.hip; fp32 and bf16x3 - the modes whose tables are all fp32 rows)
  if (w->weight_dtype != GW_DTYPE_BF16 && GW_TUNE("GW_NODE_RS", 1) != 0 && gw::node_rs_eligible(a))
    return gw::node_rs_launch(a, w->weight_dtype == GW_DTYPE_BF16X3, stream);
  if (is16(w->weight_dtype)) {
    if (w->weight_dtype == GW_DTYPE_BF16 && w->ln_gamma && a.ln_width != 256)
      return fail(GW_E_UNSUPPORTED, "gw_node_update_forward: LayerNorm over fewer than 256 features needs float32 or bf16x3 weights");
    return launch16(w->weight_dtype, n_post > 0 ? 4 : 2, a, 256, 256, 256, 1, stream);
  }
  if (n_post > 0) return launch_chain(chain_kernel<64, true, 2, 16, 16, EPI_ROWS, false, true>, a, stream, 1, 2);
  return launch_chain(chain_kernel<64, true, 2, 16, 16, EPI_ROWS>, a, stream, 1, 2);
}

int gw_node_update_row_split_groups(int64_t n_rows) { return gw::node_rs_groups(n_rows); }

int gw_node_update_head_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_operand* agg,
                                const gw_mlp_weights* w, const gw_mlp_weights* head, const gw_operand* residual, float* out,
                                int32_t out_ld, void* stream) {
  if (!x || !agg || !w || !head || !out || n_rows < 0 || rows_per_batch <= 0) return fail(GW_E_BADARG, "gw_node_update_head_forward: bad arguments");
  if (n_rows == 0) return GW_OK;
  if (n_rows >= (int64_t)1 << 31) return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: more than 2^31-1 rows");
  if (head->weight_dtype != w->weight_dtype)
    return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: the same weight dtype for both MLPs");
  const bool b16 = w->weight_dtype == GW_DTYPE_BF16;  // (the 16-bit table formats belong to the bf16 mode)
  if (w->hidden != 256 || w->n_out != 256 || !w->b1 || !w->w_out || !w->b_out || bad_layers(w) || w->n_mid != 1 || !w->ln_gamma ||
      (w->ln_width > 0 && w->ln_width != 256))
    return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: node MLP must be 512 -> 256 -> 256 -> 256 with LayerNorm");
  if (bad_agg(agg, b16) || !w->w1[1]) return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: agg must be 256 wide (raw)");
  if (x->k != 0 && (bad256(x, b16) || (!x->projected && !w->w1[0]))) return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: x must be 256 wide or zeros");
  if (head->hidden != 128 || head->n_mid != 1 || head->n_out > 80 || head->n_out <= 0 || head->ln_gamma || !head->w1[0] || !head->b1 ||
      !head->w_mid || !head->b_mid || !head->w_out || !head->b_out)
    return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: the head must be 256 -> 128 -> 128 -> <= 80 features without norm");
  // the kernel streams 8 K-steps of the head's first Linear and 80 rows of its last one whatever the caller packed
  if (head->k_in != 256 || (head->out_rows != 80 && !(head->out_rows == 0 && head->n_out == 80)))
    return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: the head must be packed for 256 inputs (k_in) and 80 output rows (out_rows)");
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  a.n_cols = (int)n_rows;
  a.cols_per_batch = rows_per_batch;
  fill_operand(a, 0, x);
  fill_operand(a, 1, agg);
  fill_weights(a, w);
  a.hd_w1 = head->w1[0];
  a.hd_b1 = head->b1;
  a.hd_w2 = head->w_mid;
  a.hd_b2 = head->b_mid;
  a.hd_w3 = head->w_out;
  a.hd_b3 = head->b_out;
  if (residual && residual->k != 0) {
    if (!residual->ptr || residual->layout != GW_LAYOUT_ROWS_F32) return fail(GW_E_BADARG, "gw_node_update_head_forward: bad residual");
    fill_residual(a, residual);
  }
  a.out = out;
  a.out_ld = out_ld;
  a.out_cols = head->n_out;
  if (!is16(w->weight_dtype)) {  // fp32 (v17): chain_kernel with the head behind the node update
    if (x->k != 0 && x->layout != GW_LAYOUT_ROWS_F32) return fail(GW_E_UNSUPPORTED, "gw_node_update_head_forward: fp32 weights take fp32 rows");
    return launch_chain(chain_kernel<64, true, 2, 16, 16, EPI_ROWS, false, false, true>, a, stream, 1, 2);
  }
  return launch16(w->weight_dtype, 6, a, 256, 256, 256, 1, stream);
}

int gw_project_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, int32_t n_slices,
                       const float* const* w_slices, void* const* outs, int32_t out_ld, int32_t out_layout, int32_t weight_dtype,
                       const float* relu_mask, float* zero_rows, void* stream) {
  if (!x || !w_slices || !outs || n_rows < 0 || rows_per_batch <= 0 || n_slices <= 0 || n_slices > 4)
    return fail(GW_E_BADARG, "gw_project_forward: bad arguments (1..4 slices)");
  if (n_rows == 0) return GW_OK;
  if (n_rows >= (int64_t)1 << 31) return fail(GW_E_UNSUPPORTED, "gw_project_forward: more than 2^31-1 rows");
  if (bad256(x) || x->projected || out_ld % 4 != 0) return fail(GW_E_UNSUPPORTED, "gw_project_forward: x must be a raw 256-wide table");
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  a.n_cols = (int)n_rows;
  a.cols_per_batch = rows_per_batch;
  fill_operand(a, 0, x);
  for (int i = 0; i < n_slices; ++i) {
    if (!w_slices[i] || !outs[i]) return fail(GW_E_BADARG, "gw_project_forward: null slice / output");
    a.proj_w[i] = w_slices[i];
    a.proj_out[i] = (float*)outs[i];
  }
  if (out_layout != GW_LAYOUT_ROWS_F32 && out_layout != GW_LAYOUT_ROWS_F16) return fail(GW_E_BADARG, "gw_project_forward: bad out_layout");
  if (out_layout == GW_LAYOUT_ROWS_F16 && weight_dtype != GW_DTYPE_BF16)
    return fail(GW_E_UNSUPPORTED, "gw_project_forward: fp16 products come with bf16 slices");
  a.proj_half = out_layout == GW_LAYOUT_ROWS_F16;
  a.out_ld = out_ld;
  a.out_cols = 256;
  if (relu_mask && (weight_dtype == GW_DTYPE_BF16 || n_slices != 1))
    return fail(GW_E_UNSUPPORTED, "gw_project_forward: relu_mask needs fp32 or bf16x3 weights and a single slice");
  a.relu_mask = relu_mask;
  if (zero_rows && weight_dtype != GW_DTYPE_F32) return fail(GW_E_UNSUPPORTED, "gw_project_forward: zero_rows needs fp32 weights");
  a.zero_rows = zero_rows;
  if (is16(weight_dtype)) return launch16(weight_dtype, 3, a, 256, 256, 256, n_slices, stream);
  return launch_chain(chain_kernel<64, true, 1, 16, 16, EPI_ROWS, true>, a, stream, n_slices, 3);
}

int gw_normalized_mse_forward(const float* pred, const float* target, const float* inv_var, int32_t inv_var_full,
                              const float* lat_weights, int32_t num_unique_lat, int32_t batch, int32_t nodes, int32_t channels,
                              float* loss_out, void* stream) {
  if (!pred || !target || !lat_weights || !loss_out || num_unique_lat <= 0 || batch <= 0 || nodes <= 0 || channels <= 0)
    return fail(GW_E_BADARG, "gw_normalized_mse_forward: bad arguments");
  const int num_lon = nodes / num_unique_lat;
  if (num_lon <= 0 || (nodes + num_lon - 1) / num_lon > num_unique_lat)
    return fail(GW_E_BADARG, "gw_normalized_mse_forward: nodes must equal num_unique_lat * num_lon");
  const size_t total = (size_t)batch * nodes * channels;
  const float scale = 1.0f / ((float)channels * (float)batch * (float)nodes);
  int grid = (int)((total + 1023) / 1024);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(nmse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, target, inv_var, inv_var_full, lat_weights, num_lon, nodes,
                     channels, total, scale, loss_out);
  return check_launch("nmse_kernel launch");
}

}  // extern "C"
