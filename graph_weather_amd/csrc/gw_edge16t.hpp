// gw_edge16t.hpp - device helpers shared by the team-pipelined bf16 edge kernels (gw_edge16t.hip: gather / DMA forms;
// gw_edge16p.hip: the segment-aligned processor form).  gfx950 only.
#ifndef GW_EDGE16T_HPP
#define GW_EDGE16T_HPP

#include "gw_edge16.hpp"

namespace gw16t {
using namespace gw;
using namespace gw16;

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// LDS traffic of this wave is complete, then the workgroup barrier.  Not __syncthreads(): its release fence also drains the
// vector-memory counter, i.e. the residual / index loads and the LDS-DMA that are meant to stay in flight across the barrier.
__device__ __forceinline__ void team_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// The value recomputed from here on: keeps cheap lane-derived offsets (lane * 16, ...) from being shared kernel-wide as one
// long-lived (and then spilled) register - a scratch reload queues behind the wave's stores like any other load.
__device__ __forceinline__ int fresh(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ float relu1(float x) {  // one v_max_f32 (fmaxf adds a canonicalising v_max in front)
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// LayerNorm sums of the transposed output layer, pieces mm = 16 .. 27 of a group: the reduce-scatter of (s1[0..3], s2[0..3])
// over the 16 feature lanes (gw_edge16.hpp: 16 DPP instructions) spread over the 12 filler slots.  Inside the layer an MFMA
// lies between a producer and its DPP use; in the tail behind the last MFMA (group 3) nothing does: wait states there.
template <int OP, bool WAIT>
__device__ __forceinline__ void reduce_op_range(int lo, int hi, const float (&s1)[4], const float (&s2)[4], float (&ra)[4],
                                                float (&rb)[2], float (&rc)[2], float (&rd)[2]) {
  if constexpr (OP < 16) {
    if (OP >= lo && OP < hi) row_reduce_scatter_op<OP, WAIT>(s1, s2, ra, rb, rc, rd);
    reduce_op_range<OP + 1, WAIT>(lo, hi, s1, s2, ra, rb, rc, rd);
  }
}
__device__ __forceinline__ void reduce_ops(int g, int mm, const float (&s1)[4], const float (&s2)[4], float (&ra)[4], float (&rb)[2],
                                           float (&rc)[2], float (&rd)[2]) {
  const int lo = ((mm - 16) * 16) / 12, hi = ((mm - 15) * 16) / 12;  // (compile-time after unrolling)
  if (g == 3) reduce_op_range<0, true>(lo, hi, s1, s2, ra, rb, rc, rd);
  else reduce_op_range<0, false>(lo, hi, s1, s2, ra, rb, rc, rd);
}

// One resident layer of a team wave on the 4 groups of a tile (4 row tiles x 8 K-steps x 4 groups = 128 MFMAs), software
// pipelined inside the wave.  The team has ONE wave per SIMD, and a wave issues in order: whatever it does between two MFMAs
// beyond the ~2 issue slots the 16-cycle matrix instruction covers is time the matrix pipe idles (measured: 128 MFMAs with their
// ~370 other instructions in blocks took 4.3 k cycles, = not overlapped at all).  So every MFMA is followed by at most one
// small filler:
//  * slots 0..3 of each half-group (16 MFMAs): one B fragment (ds_read_b128) of the NEXT half-group into the other buffer;
//  * slots 2..29 of group g: piece m = slot - 2 of the epilogue of group g - 1 (what the caller does with its accumulators:
//    relu + bf16 pack + LDS store, or LayerNorm partial sums), one or two instructions each;
//  * slots 28..31: the bias of group g + 1 into its accumulator set (LDS reads straight into the accumulators).
// Two accumulator sets alternate (a caller that keeps all groups passes 4).  The pieces of the last group run at the end.
// bias(dst, t): accumulator of row tile t <- bias;  piece(g, m, acc): m = 0 .. 27.
// BIAS_SLOT: first of the 4 consecutive filler slots of group g in which the bias of group g + 1 is read from LDS into its
// accumulator set.  The default 28 leaves 4 MFMAs (~70 cycles) to the first use - less than an LDS round trip under load, i.e.
// a stall per group (measured with the partner team parked: 128 MFMAs took 3.1 k cycles instead of 2.2 k); callers whose
// epilogue pieces are done with that set earlier (or that keep all 4 sets) name an earlier slot.
template <int NSETS, bool TR = false, bool F16 = false, int BIAS_SLOT = 28, bool BIAS_C = false, class Bias, class Piece>
__device__ __forceinline__ void team_layer(f32x4 (&acc)[NSETS][4], const bf16x8 (&w)[4][8], const char* __restrict__ hin, int lane,
                                           Bias bias, Piece piece) {
#ifdef GW_LAYER_ABL  // tuning builds only (timing experiments, wrong results; -DGW_LAYER_ABL=n through GW_HIPCC_EXTRA): 1 = no
  constexpr int abl = GW_LAYER_ABL;  // pieces, 2 = no bias reads (a COMPILE-time switch: a scalar branch per MFMA slot doubles the layer)
#else
  constexpr int abl = 0;
#endif
  // TR: ONE ring of 4 fragment registers instead of two alternating sets - the fragment of K-step ks is requested again (for
  // the next half-group) right behind the 4th MFMA that reads it (issued in order, its operands are read long before the LDS data
  // returns) and is needed 13 MFMAs later; the transposed layer's filler state (8 running sums) takes the 16 registers
  constexpr int NFR = TR ? 1 : 2;
  bf16x8 fr[NFR][4];
  const char* const p0 = hin + fresh(lane) * 16;
#pragma unroll
  for (int s = 0; s < 4; ++s) fr[0][s] = *(const bf16x8*)(p0 + s * 1024);
  // BIAS_C: the bias of row tile t sits in 4 registers for the whole layer and enters as the C operand of each group's first
  // product (the same for every group) - no LDS read, no address arithmetic per group, and the accumulator set of the NEXT group
  // is not alive before its first MFMA
  f32x4 bc[BIAS_C ? 4 : 1];
  if constexpr (BIAS_C) {
#pragma unroll
    for (int t = 0; t < 4; ++t) bias(bc[t], t);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) bias(acc[0][t], t);  // (defines the first set)
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const int g = h >> 1, hh = h & 1;
    f32x4 (&ac)[4] = acc[g % NSETS];
#pragma unroll
    for (int m16 = 0; m16 < 16; ++m16) {
      const int ks = m16 >> 2, t = m16 & 3, slot = 16 * hh + m16;
      const bool first = BIAS_C && hh == 0 && ks == 0;  // (the first product of this group's accumulator t)
      if constexpr (TR) {
        if (first) mfma_t_c(ac[t], fr[0][ks], w[t][4 * hh + ks], bc[t]);
        else mfma_t(ac[t], fr[0][ks], w[t][4 * hh + ks]);  // (the transposed product, see mfma_t)
        if (t == 3 && h + 1 < 8) fr[0][ks] = *(const bf16x8*)(p0 + ((h + 1) >> 1) * 8192 + ((h + 1) & 1) * 4096 + ks * 1024);
      } else {
        if constexpr (F16) {
          if (first) mfma_a_f16_c(ac[t], w[t][4 * hh + ks], fr[h & 1][ks], bc[t]);
          else mfma_a_f16(ac[t], w[t][4 * hh + ks], fr[h & 1][ks]);
        } else {
          if (first) mfma_a_c(ac[t], w[t][4 * hh + ks], fr[h & 1][ks], bc[t]);
          else mfma_a(ac[t], w[t][4 * hh + ks], fr[h & 1][ks]);
        }
        if (m16 < 4 && h + 1 < 8)
          fr[(h + 1) & 1][m16] = *(const bf16x8*)(p0 + ((h + 1) >> 1) * 8192 + ((h + 1) & 1) * 4096 + m16 * 1024);
      }
      if (g > 0 && slot >= 2 && slot < 30 && !(abl & 1)) piece(g - 1, slot - 2, acc[(g - 1) % NSETS]);
      if (!BIAS_C && slot >= BIAS_SLOT && slot < BIAS_SLOT + 4 && g + 1 < 4 && !(abl & 2)) bias(acc[(g + 1) % NSETS][slot - BIAS_SLOT], slot - BIAS_SLOT);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    f32x4 (&ac)[4] = acc[3 % NSETS];
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]));  // MFMA results -> VALU readers
#pragma unroll
    for (int m = 0; m < 28; ++m)
      if (!(abl & 1)) piece(3, m, ac);
  }
}

struct TileId {
  int eb, b;
};


}  // namespace gw16t

#endif  // GW_EDGE16T_HPP
