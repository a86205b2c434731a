// gw_edge.hip - the edge-update kernel of the hot path, specialised for what the forecaster actually launches.
//
// EdgeProcessor.forward + scatter_sum (reference graph_net_block.py:131-137 and :188) for destination-sorted edges:
//     e' = LN(W_out . relu(W_mid . relu(b1 + [W_raw . raw] + sum_p P_p[row_p]) + b_mid) + b_out) + e_res
//     agg[dst] += e'
// After the layer-1 split every launch of the forecaster has AT MOST ONE operand that still
// needs a matrix pass in layer 1 (the per-sample edge features in processor blocks 1..8, the grid-node features in
// the encoder); the others are per-node / per-edge products that are only gathered and added.  Compared with the
// general chain_kernel this kernel
//   * keeps the gathered rows of the projected operands in a two-chunk-deep register ring, requested two weight
//     chunks ahead from asm statements hipcc does not count, so no gather is waited for where it is issued,
//   * streams the weights with asm-issued LDS-DMA (hipcc's waits stay counted instead of full drains), two 1 KiB
//     pieces per K-step interleaved with the MFMAs, and does the chunk hand-over (wait, barrier, first fragments of the
//     next chunk) in front of the last 16 MFMAs of a chunk instead of behind them,
//   * needs no layer-1 accumulator at all when nothing is raw (decoder, first processor block),
//   * stages e' through LDS (the weight buffers are free by then) so that the rows of e_out are written with one
//     fully coalesced 1 KiB store per row and the segment sum runs with one thread per feature: interior
//     destination runs of a tile become plain coalesced stores, only the first and last run of a tile use
//     (row-coalesced) atomics.
// Same register-resident transposed-MLP scheme and weight stream as chain_kernel (gw_kernels.hip).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

constexpr int kStageLd = 260;                                  // floats per staged row: 65 x 16 B -> conflict-free b128 writes
constexpr int kStageFloats = kColsPerWG * kStageLd;            // 64 rows
constexpr int kEdgeLdsBytes = (kStageFloats + kColsPerWG) * 4;  // staging (overlays the 64 KiB weight buffers) + 64 dst ids
static_assert(kStageFloats * 4 >= kLdsBytes, "staging area must cover the weight double buffer");
constexpr int kChunkFloats = kChunkSteps * 1024;               // one weight chunk: 8 K-steps x 256 rows x 4 k = 32 KiB
constexpr int kChunksPerLayer = 64 / kChunkSteps;              // K = 256

struct EdgeArgs {
  int n_cols;  // batch * n_edges
  int n_edges;
  int n_dst;
  int stagger;
  int skip;  // tuning aid (GW_EDGE_SKIP): 1 = no segment sum, 2 = no staging either (results are then wrong)
  int dma6;  // tuning aid (GW_EDGE_DMA6): DMA pieces of the next chunk over six K-steps instead of four
  // XCD-aware tile order: workgroup i is dispatched to XCD i % 8 (each XCD has its own L2); XCD x then walks the contiguous
  // tile range [x * xcd_base + min(x, xcd_rem), ...) so that neighbouring destination-sorted tiles - which gather the
  // same few mesh rows - share an L2.  xcd_base == 0: identity order.
  int xcd_base;
  int xcd_rem;
  unsigned long long* dbg;
  int dbg_cap;
  const int* src;
  const int* dst;
  // raw operand (RAW kernels): full 256-float rows, multiplied by w_raw on the matrix cores
  const float* raw_ptr;
  int raw_rows_pb;
  int raw_ld;
  int raw_kind;  // 0: row = src[k], 1: dst[k], 2: k
  const float* w_raw;
  // projected operands: rows already hold X . W1_slice^T, gathered and added
  const float* p_ptr[3];
  int p_rows_pb[3];
  int p_ld[3];
  int p_kind[3];
  // weights
  const float* b1;
  const float* w_mid;
  const float* b_mid;
  int n_mid;
  const float* w_out;
  const float* b_out;
  const float* gamma;
  const float* beta;
  // residual e rows (indexed by edge), outputs
  const float* res_ptr;
  int res_rows_pb;
  int res_ld;
  float* e_out;
  float* agg;
  float* carry;  // deterministic segment sums: per-tile carry records (gw_internal.hpp), NULL = atomics
  // training: activations saved for the backward (gw_activation_save; NULL in inference); n_mid == 1 only
  float* save_h;
  long long save_stride;
  int save_ld;
  float* save_y;
};

#ifdef GW_TUNING
#define GW_DMA6(a) ((a).dma6)
#else
#define GW_DMA6(a) 0
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- loads hipcc must not count -------------------------------------------------------------------------------
// global_load_lds is a FLAT-class instruction: while one is pending in hipcc's model, every s_waitcnt it generates
// for a VMEM result is vmcnt(0) - which would drain the weight DMA of the NEXT chunk each time a gathered register is
// first used.  The loads whose results are consumed inside the chunk loops are therefore issued from asm statements
// (invisible to that bookkeeping) and completed by explicit counted waits that name their destination registers
// ("+v"), so no consumer can be scheduled above the wait (cdna_hip_programming.md 5.7, form ii).
template <int OFF>
__device__ __forceinline__ f32x4 hld4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "i"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ int hldi(const int* p) {
  int v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void wait_regs(int& a, int& b) {
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(a), "+v"(b) : [n] "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_regs(f32x4 (&r)[1][2]) {
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r[0][0]), "+v"(r[0][1]) : [n] "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_regs(f32x4 (&r)[2][2]) {
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]) : [n] "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_regs(f32x4 (&r)[3][2]) {
  asm volatile("s_waitcnt vmcnt(%[n])"
               : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[2][0]), "+v"(r[2][1])
               : [n] "n"(N)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_regs(f32x4 (&r)[4][2]) {
  asm volatile("s_waitcnt vmcnt(%[n])"
               : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[2][0]), "+v"(r[2][1]), "+v"(r[3][0]),
                 "+v"(r[3][1])
               : [n] "n"(N)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_regs(f32x4 (&r)[16]) {
  asm volatile("s_waitcnt vmcnt(%[n])"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                 "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
               : [n] "n"(N)
               : "memory");
}
// 16 x 16 B of one 256-float row (accumulator layout: tile t at +16t floats; p already includes the 4q lane offset)
__device__ __forceinline__ void hld_row(f32x4 (&r)[16], const float* p) {
  r[0] = hld4<0>(p);     r[1] = hld4<64>(p);    r[2] = hld4<128>(p);   r[3] = hld4<192>(p);
  r[4] = hld4<256>(p);   r[5] = hld4<320>(p);   r[6] = hld4<384>(p);   r[7] = hld4<448>(p);
  r[8] = hld4<512>(p);   r[9] = hld4<576>(p);   r[10] = hld4<640>(p);  r[11] = hld4<704>(p);
  r[12] = hld4<768>(p);  r[13] = hld4<832>(p);  r[14] = hld4<896>(p);  r[15] = hld4<960>(p);
}

// one half (tiles 8h .. 8h+7) of such a row
template <int H>
__device__ __forceinline__ void hld_half_row(f32x4 (&r)[16], const float* p) {
  r[8 * H + 0] = hld4<512 * H + 0>(p);    r[8 * H + 1] = hld4<512 * H + 64>(p);
  r[8 * H + 2] = hld4<512 * H + 128>(p);  r[8 * H + 3] = hld4<512 * H + 192>(p);
  r[8 * H + 4] = hld4<512 * H + 256>(p);  r[8 * H + 5] = hld4<512 * H + 320>(p);
  r[8 * H + 6] = hld4<512 * H + 384>(p);  r[8 * H + 7] = hld4<512 * H + 448>(p);
}

// Workgroup barrier for the LDS weight ring.  __syncthreads() carries a workgroup-scope release fence, which hipcc
// lowers to s_waitcnt vmcnt(0): that would drain the gathers this kernel deliberately keeps in flight.  Here only
// LDS traffic has to be ordered: this wave's LDS reads are complete (lgkmcnt(0)) and its share of the weight DMA
// has landed (the caller's counted wait) before it arrives.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

extern __shared__ __attribute__((aligned(16))) float gw_edge_lds[];
__device__ __forceinline__ float* lds_base() { return gw_edge_lds; }
// LDS byte address of the dynamic segment (0 unless the kernel also had static LDS; taken from the pointer, not assumed)
__device__ __forceinline__ unsigned lds_base_bytes() {
  return (unsigned)(size_t)(__attribute__((address_space(3))) float*)gw_edge_lds;
}

// Each wave DMAs its 8 KiB share of one 32 KiB weight chunk into an LDS buffer: exactly 8 x global_load_lds (1 KiB
// each), no branches - the vmcnt(N) bookkeeping of the kernel counts on that.
__device__ __forceinline__ void issue_chunk32k(const float* __restrict__ g, float* ldsbuf, int lane, int wave) {
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_base_bytes() + (unsigned)(ldsbuf - lds_base()) * 4u + (unsigned)wave * 1024u);
#pragma unroll
  for (int i = 0; i < 8; ++i) glds16_asm_s(g + (size_t)(wave + 4 * i) * 256, (unsigned)lane * 16u, base + (unsigned)i * 4096u);
}

// N of this wave's 8 pieces (first..first+N-1) of a chunk; buf_floats = float offset of the destination LDS buffer.
constexpr int kDmaSteps = 4;  // the 8 pieces of the next chunk are issued 2 per K-step during the first 4 steps of a chunk
template <int N>
__device__ __forceinline__ void issue_pieces(const float* __restrict__ g, int buf_floats, int first, int lane, int wave) {
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_base_bytes() + (unsigned)buf_floats * 4u + (unsigned)wave * 1024u);
#pragma unroll
  for (int i = 0; i < N; ++i)
    glds16_asm_s(g + (size_t)(wave + 4 * (first + i)) * 256, (unsigned)lane * 16u, base + (unsigned)(first + i) * 4096u);
}

// Source of weight chunk i of a tile: [w_raw (8 chunks)] w_mid (8 x n_mid, contiguous) w_out (8).
template <bool RAW>
__device__ __forceinline__ const float* chunk_src(const EdgeArgs& a, int i) {
  if (RAW) {
    if (i < kChunksPerLayer) return a.w_raw + (size_t)i * kChunkFloats;
    i -= kChunksPerLayer;
  }
  const int nm = a.n_mid * kChunksPerLayer;
  if (i < nm) return a.w_mid + (size_t)i * kChunkFloats;
  return a.w_out + (size_t)(i - nm) * kChunkFloats;
}

// One weight chunk (8 K-steps x 16 row tiles): ACC[t] += W[16t.., k(s)] * IN8[s].
// The A fragments of a step are read from LDS one step ahead (a_cur / a_nxt), ACROSS chunk boundaries: during the
// last step of chunk ci the boundary work for chunk ci+1 is done - counted wait WAIT_STMT (this wave's share of chunk
// ci+1 has landed), workgroup barrier (everybody's share has, and everybody is done reading chunk ci), DMA of chunk
// ci+2 into the buffer chunk ci vacates, first fragments of chunk ci+1 - and only then the step's 16 MFMAs are
// issued, so barrier skew and LDS latency sit underneath 16 MFMAs instead of draining the matrix pipe.
#define GW_CHUNK(ACC, IN8, NEXT_EXISTS, WAIT_STMT)                                                             \
  {                                                                                                            \
    const float* bl_ = lds + (ci & 1) * kLdsBufFloats + lane * 4;                                              \
    const float* nsrc_ = chunk_src<RAW>(a, ci + 1);                                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < kChunkSteps; ++s_) {                                               \
      f32x4 a_nxt_[4];                                                                                         \
      if ((NEXT_EXISTS) && GW_DMA6(a)) { /* tuning: the 8 pieces over SIX K-steps (2, 1, 1, 2, 1, 1) */                    \
        if (s_ == 0) issue_pieces<2>(nsrc_, ((ci + 1) & 1) * kLdsBufFloats, 0, lane, wave);                    \
        if (s_ == 1) issue_pieces<1>(nsrc_, ((ci + 1) & 1) * kLdsBufFloats, 2, lane, wave);                    \
        if (s_ == 2) issue_pieces<1>(nsrc_, ((ci + 1) & 1) * kLdsBufFloats, 3, lane, wave);                    \
        if (s_ == 3) issue_pieces<2>(nsrc_, ((ci + 1) & 1) * kLdsBufFloats, 4, lane, wave);                    \
        if (s_ == 4) issue_pieces<1>(nsrc_, ((ci + 1) & 1) * kLdsBufFloats, 6, lane, wave);                    \
        if (s_ == 5) issue_pieces<1>(nsrc_, ((ci + 1) & 1) * kLdsBufFloats, 7, lane, wave);                    \
      } else if ((NEXT_EXISTS) && s_ < kDmaSteps && GW_SKIP(a) != 4) {                                              \
        issue_pieces<8 / kDmaSteps>(nsrc_, ((ci + 1) & 1) * kLdsBufFloats, s_ * (8 / kDmaSteps), lane, wave);  \
      }                                                                                                        \
      if (s_ + 1 < kChunkSteps) {                                                                              \
        _Pragma("unroll") for (int b4 = 0; b4 < 4; ++b4) a_nxt_[b4] = *(const f32x4*)(bl_ + (s_ + 1) * 1024 + b4 * 256); \
      } else if (NEXT_EXISTS) {                                                                                \
        WAIT_STMT;                                                                                             \
        if (GW_SKIP(a) != 5) lds_barrier();                                                                        \
        const float* bn_ = lds + ((ci + 1) & 1) * kLdsBufFloats + lane * 4;                                    \
        _Pragma("unroll") for (int b4 = 0; b4 < 4; ++b4) a_nxt_[b4] = *(const f32x4*)(bn_ + b4 * 256);          \
      }                                                                                                        \
      const float b_ = IN8[s_];                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int t = 0; t < 16; ++t)                                                           \
          ACC[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t >> 2][t & 3], b_, ACC[t], 0, 0, 0);             \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      if (s_ + 1 < kChunkSteps || (NEXT_EXISTS)) {                                                             \
        _Pragma("unroll") for (int b4 = 0; b4 < 4; ++b4) a_cur[b4] = a_nxt_[b4];                                \
      }                                                                                                        \
    }                                                                                                          \
    ++ci;                                                                                                      \
  }

template <bool RAW, int NPROJ>
__global__ __launch_bounds__(kThreads, 2) void edge_kernel(const EdgeArgs a) {
  float* const lds = lds_base();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  int tile = blockIdx.x;
  if (a.xcd_base > 0) {
    const int xcd = tile & 7, idx = tile >> 3;
    tile = xcd * a.xcd_base + (xcd < a.xcd_rem ? xcd : a.xcd_rem) + idx;
  }
  const int tile_c0 = tile * kColsPerWG;
  const int c_raw = tile_c0 + wave * kColsPerWave + j;
  const bool valid = c_raw < a.n_cols;
  const int c = valid ? c_raw : a.n_cols - 1;
  const int b = c / a.n_edges;
  const int k = c - b * a.n_edges;
  const int total = kChunksPerLayer * ((RAW ? 1 : 0) + a.n_mid + 1);  // weight chunks per tile

  // anti-phase start of the second batch of workgroups (see chain_kernel)
  if (a.stagger > 0 && (blockIdx.x >> 8) == 1) {
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  GW_STAMP(0)

  // ---- prologue.  Issue order matters for the counted waits (vmcnt retires in order) ----
  int s_idx = hldi(a.src + k);
  int d_idx = hldi(a.dst + k);
  issue_chunk32k(chunk_src<RAW>(a, 0), lds, lane, wave);
  int ci = 0;  // chunk counter of this tile (wave uniform); chunk i lives in LDS buffer i & 1
  wait_regs<8>(s_idx, d_idx);  // the 8 DMA pieces stay in flight

  const float* prow[NPROJ];
#pragma unroll
  for (int p = 0; p < NPROJ; ++p) {
    const int r = a.p_kind[p] == 0 ? s_idx : (a.p_kind[p] == 1 ? d_idx : k);
    prow[p] = a.p_ptr[p] + ((size_t)b * (size_t)a.p_rows_pb[p] + (size_t)r) * (size_t)a.p_ld[p] + 4 * q;
  }
  const int gd_id = valid ? b * a.n_dst + d_idx : -1;  // global destination row of this column (segment-sum key)

  // ring[s & 1][p][h]: features 32s + 16h + 4q .. +3 of projected operand p = its part of the B operand of produce
  // chunk s.  Slice s is requested when slice s-2 has been consumed (end of chunk s-3) and consumed at the end of
  // chunk s-1.  When nothing is raw the layer-1 bias is one more ring member (same address for all columns).
  constexpr int NR = RAW ? NPROJ : NPROJ + 1;
  const float* brow = a.b1 + 4 * q;
  f32x4 ring[2][NR][2];
#define GW_REQUEST_SLICE(slot, slice)                                          \
  {                                                                            \
    _Pragma("unroll") for (int p = 0; p < NPROJ; ++p) {                        \
      ring[slot][p][0] = hld4<128 * (slice)>(prow[p]);                         \
      ring[slot][p][1] = hld4<128 * (slice) + 64>(prow[p]);                    \
    }                                                                          \
    if (!RAW) {                                                                \
      ring[slot][NR - 1][0] = hld4<128 * (slice)>(brow);                       \
      ring[slot][NR - 1][1] = hld4<128 * (slice) + 64>(brow);                  \
    }                                                                          \
  }
  // B operand of produce chunk `slice` from ring slot `slot` (+ the layer-1 accumulator tiles when RAW)
#define GW_CONSUME_SLICE(slot, slice)                                          \
  {                                                                            \
    f32x4 v0_, v1_;                                                            \
    if (RAW) {                                                                 \
      v0_ = acc[2 * (slice)] + ring[slot][0][0];                               \
      v1_ = acc[2 * (slice) + 1] + ring[slot][0][1];                           \
    } else {                                                                   \
      v0_ = ring[slot][0][0];                                                  \
      v1_ = ring[slot][0][1];                                                  \
    }                                                                          \
    _Pragma("unroll") for (int p = 1; p < NR; ++p) {                           \
      v0_ += ring[slot][p][0];                                                 \
      v1_ += ring[slot][p][1];                                                 \
    }                                                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                            \
      in8[r] = fmaxf(v0_[r], 0.f);                                             \
      in8[4 + r] = fmaxf(v1_[r], 0.f);                                         \
    }                                                                          \
    if (a.save_h != nullptr && valid) { /* relu output of layer 1, features 32 slice + 4q.. and + 16 */ \
      float* sr_ = a.save_h + (size_t)c * (size_t)a.save_ld + 32 * (slice) + 4 * q;                     \
      stg4(sr_, f32x4{in8[0], in8[1], in8[2], in8[3]});                        \
      stg4(sr_ + 16, f32x4{in8[4], in8[5], in8[6], in8[7]});                   \
    }                                                                          \
  }

  f32x4 a_cur[4];  // A fragments of the next K-step to run
  f32x4 acc[16];   // layer-1 accumulator (RAW only)
  f32x4 acc2[16];  // first hidden layer accumulator
  float in8[8];    // B operand values of the next chunk
  if (RAW) {
    const int r = a.raw_kind == 0 ? s_idx : (a.raw_kind == 1 ? d_idx : k);
    const float* xrow = a.raw_ptr + ((size_t)b * (size_t)a.raw_rows_pb + (size_t)r) * (size_t)a.raw_ld + 4 * q;
    f32x4 xv[16];
    hld_row(xv, xrow);
    hld_row(acc, brow);
    wait_regs<0>(xv);  // chunk 0, x and the bias rows have landed
    wait_regs<0>(acc);
    lds_barrier();
    GW_STAMP(1)
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) a_cur[b4] = *(const f32x4*)(lds + lane * 4 + b4 * 256);
    float x[64];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[4 * i + 0] = xv[i].x;
      x[4 * i + 1] = xv[i].y;
      x[4 * i + 2] = xv[i].z;
      x[4 * i + 3] = xv[i].w;
    }
#pragma unroll
    for (int cc = 0; cc < kChunksPerLayer; ++cc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) in8[i] = x[8 * cc + i];
      if (cc == kChunksPerLayer - 3) {  // the first two ring slices, once most of x is dead
        GW_REQUEST_SLICE(0, 0)
        GW_REQUEST_SLICE(1, 1)
      }
      if (cc == kChunksPerLayer - 1) hld_row(acc2, a.b_mid + 4 * q);  // next layer's bias, under the last chunk (x is dead)
      if (cc == kChunksPerLayer - 1) {
        GW_CHUNK(acc, in8, true, wait_regs<0>(acc2))
      } else {
        GW_CHUNK(acc, in8, true, wait_vm<0>())
      }
    }
    wait_regs<0>(ring[0]);  // (landed long ago: everything was drained by the vmcnt(0) boundaries above)
    wait_regs<0>(ring[1]);
    GW_CONSUME_SLICE(0, 0)
    GW_REQUEST_SLICE(0, 2)
  } else {
    hld_row(acc2, a.b_mid + 4 * q);
    GW_REQUEST_SLICE(0, 0)
    GW_REQUEST_SLICE(1, 1)
    wait_regs<2 * NR>(acc2);     // chunk 0, the bias rows and slice 0 have landed;
    wait_regs<2 * NR>(ring[0]);  // slice 1 stays in flight
    lds_barrier();
    GW_STAMP(1)
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) a_cur[b4] = *(const f32x4*)(lds + lane * 4 + b4 * 256);
    GW_CONSUME_SLICE(0, 0)
    GW_REQUEST_SLICE(0, 2)
  }
  GW_STAMP(2)

  // ---- first hidden layer: B operand produced slice by slice = relu(layer-1 accumulator + gathered rows) ----
  // boundary into produce chunk cc+1: the pieces of chunk cc+1 were issued during this chunk's first K-steps, i.e.
  // AFTER slice cc+2 was requested, so the wait is a full drain; the slice has had a whole chunk to land.
#define GW_PRODUCE_CHUNK(cc)                                                              \
  {                                                                                       \
    if ((cc) <= 6) {                                                                      \
      GW_CHUNK(acc2, in8, true, wait_regs<0>(ring[((cc) + 1) & 1]))                        \
    } else {                                                                              \
      GW_CHUNK(acc2, in8, true, wait_vm<0>())                                              \
    }                                                                                     \
    if ((cc) + 1 < kChunksPerLayer) GW_CONSUME_SLICE(((cc) + 1) & 1, (cc) + 1)             \
  }
  GW_PRODUCE_CHUNK(0)
  GW_REQUEST_SLICE(1, 3)
  GW_PRODUCE_CHUNK(1)
  GW_REQUEST_SLICE(0, 4)
  GW_PRODUCE_CHUNK(2)
  GW_REQUEST_SLICE(1, 5)
  GW_PRODUCE_CHUNK(3)
  GW_REQUEST_SLICE(0, 6)
  GW_PRODUCE_CHUNK(4)
  GW_REQUEST_SLICE(1, 7)
  GW_PRODUCE_CHUNK(5)
  GW_PRODUCE_CHUNK(6)
  GW_PRODUCE_CHUNK(7)

  // ---- further hidden layers (hidden_layers > 2; not used by the forecaster defaults) ----
  float hin[64];
#pragma unroll 1
  for (int l = 1; l < a.n_mid; ++l) {
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) hin[4 * t + r] = fmaxf(acc2[t][r], 0.f);
    hld_row(acc2, a.b_mid + l * 256 + 4 * q);
    wait_regs<0>(acc2);
#pragma unroll
    for (int cc = 0; cc < kChunksPerLayer; ++cc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) in8[i] = hin[8 * cc + i];
      GW_CHUNK(acc2, in8, true, wait_vm<0>())
    }
  }
  GW_STAMP(3)

  // ---- output layer; the residual rows are requested underneath it ----
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) hin[4 * t + r] = fmaxf(acc2[t][r], 0.f);
  if (a.save_h != nullptr && valid) {
    float* sr = a.save_h + (size_t)a.n_mid * (size_t)a.save_stride + (size_t)c * (size_t)a.save_ld + 4 * q;
#pragma unroll
    for (int t = 0; t < 16; ++t) stg4(sr + 16 * t, f32x4{hin[4 * t], hin[4 * t + 1], hin[4 * t + 2], hin[4 * t + 3]});
  }
  f32x4 o[16];
  f32x4 rres[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cc = 0; cc < kChunksPerLayer; ++cc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) in8[i] = hin[8 * cc + i];
    if (cc + 1 < kChunksPerLayer) {
      GW_CHUNK(o, in8, true, wait_vm<0>())
    } else {
      GW_CHUNK(o, in8, false, wait_vm<0>())
    }
    // residual rows, requested late and in halves: by now 32 / 48 of the 64 B-operand registers are dead
    if (cc == 3 || cc == 5) {  // (pointer recomputed here rather than kept live since the prologue)
      const float* rrow = a.res_ptr + ((size_t)b * (size_t)a.res_rows_pb + (size_t)k) * (size_t)a.res_ld + 4 * q;
      if (cc == 3) hld_half_row<0>(rres, rrow); else hld_half_row<1>(rres, rrow);
    }
  }
  wait_regs<0>(rres);
  GW_STAMP(4)

  // ---- bias, LayerNorm over the 256 features of each column (eps 1e-5, biased variance), residual ----
  {
#pragma unroll
    for (int t = 0; t < 16; ++t) o[t] += ldg4(a.b_out + 16 * t + 4 * q);
    if (a.save_y != nullptr && valid) {
      float* sr = a.save_y + (size_t)c * 256 + 4 * q;
#pragma unroll
      for (int t = 0; t < 16; ++t) stg4(sr + 16 * t, o[t]);
    }
    constexpr float inv_n = 1.0f / 256.0f;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += (o[t].x + o[t].y) + (o[t].z + o[t].w);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * inv_n;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = o[t][r] - mean;
        v += d * d;
      }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const float rstd = 1.0f / sqrtf(v * inv_n + 1e-5f);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const f32x4 gm = ldg4(a.gamma + 16 * t + 4 * q);
      const f32x4 bt = ldg4(a.beta + 16 * t + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] = (o[t][r] - mean) * rstd * gm[r] + bt[r] + rres[t][r];
    }
  }

  if (GW_SKIP(a) >= 2) return;
  // ---- stage e' through LDS: [64 columns][260] + 64 global destination ids ----
  __syncthreads();  // every wave is done reading the weight buffers
  {
    float* srow = lds + (wave * kColsPerWave + j) * kStageLd + 4 * q;
#pragma unroll
    for (int t = 0; t < 16; ++t) *(f32x4*)(srow + 16 * t) = o[t];
    if (q == 0) ((int*)(lds + kStageFloats))[wave * kColsPerWave + j] = gd_id;
  }
  __syncthreads();
  GW_STAMP(5)
  const int* gdl = (const int*)(lds + kStageFloats);

  // e_out rows: one 1 KiB coalesced store per row (wave w writes the rows of its own 16 columns)
  if (a.e_out != nullptr) {
#pragma unroll 4
    for (int i = 0; i < kColsPerWave; ++i) {
      const int col = wave * kColsPerWave + i;
      if (tile_c0 + col < a.n_cols) {
        const f32x4 vv = *(const f32x4*)(lds + col * kStageLd + 4 * lane);
        stg4(a.e_out + (size_t)(tile_c0 + col) * 256 + 4 * lane, vv);
      }
    }
  }

  if (GW_SKIP(a) >= 1) return;
  // segment sum: thread f owns feature f; columns are sorted by global destination id, so equal ids form runs.
  // Interior runs belong to this tile alone -> plain stores; the first and the last run may continue in the
  // neighbouring tiles -> atomics (agg is zero-filled by the caller).
  // All 64 LDS reads are issued first; segment ends are the same for every thread, so lane i compares column i's
  // destination with column i + 1's and the ballot gives a 64-bit scalar mask: the walk is straight-line code that tests
  // one bit per column (a rarely taken scalar branch) instead of a data-dependent branch per column around an LDS read.
  {
    const int f = threadIdx.x;
    float vv[kColsPerWG];
#pragma unroll
    for (int i = 0; i < kColsPerWG; ++i) vv[i] = lds[i * kStageLd + f];
    const int gdv = gdl[lane];
    const int gdn = gdl[lane < kColsPerWG - 1 ? lane + 1 : lane];
    const unsigned long long ends = __ballot(lane == kColsPerWG - 1 || gdn != gdv);  // bit i: a run ends with column i
    // deterministic mode: is the first / last run of this tile open towards the neighbouring tile (same destination row)?
    bool open_lo = true, open_hi = true;  // atomics mode: assume so
    float* rec = nullptr;
    if (a.carry != nullptr) {
      rec = a.carry + (size_t)tile * kCarryFloats;
      const int c_prev = tile_c0 - 1, c_next = tile_c0 + kColsPerWG;
      int gd_prev = -2, gd_next = -2;
      if (c_prev >= 0) {
        const int bp = c_prev / a.n_edges;
        gd_prev = bp * a.n_dst + ldgi(a.dst + (c_prev - bp * a.n_edges));
      }
      if (c_next < a.n_cols) {
        const int bn = c_next / a.n_edges;
        gd_next = bn * a.n_dst + ldgi(a.dst + (c_next - bn * a.n_edges));
      }
      open_lo = gd_prev == __builtin_amdgcn_readlane(gdv, 0);
      open_hi = gd_next == __builtin_amdgcn_readlane(gdv, kColsPerWG - 1);
      if (f == 0) {  // header: no slot used yet (the same thread fills it in below)
        rec[512] = __int_as_float(-1);
        rec[513] = __int_as_float(-1);
        rec[514] = __int_as_float(0);
      }
    }
    float run = 0.f;
    bool first = true;
#pragma unroll
    for (int i = 0; i < kColsPerWG; ++i) {
      run += vv[i];
      if (__builtin_expect((ends >> i) & 1ull, 0)) {
        const int cur = __builtin_amdgcn_readlane(gdv, i);
        if (cur >= 0) {
          float* dstp = a.agg + (size_t)cur * 256 + f;
          const bool lo = first && open_lo, hi = i == kColsPerWG - 1 && open_hi;
          if (rec != nullptr) {
            if (lo) {  // continues from the previous tile: slot 0 (and the chain runs on through this tile if it is open high too)
              rec[f] = run;
              if (f == 0) {
                rec[512] = __int_as_float(cur);
                if (hi) rec[514] = __int_as_float(1);
              }
            } else if (hi) {  // a segment that starts here and continues in the next tile: slot 1
              rec[256 + f] = run;
              if (f == 0) rec[513] = __int_as_float(cur);
            } else {
              stg1(dstp, run);
            }
          } else if (first || i == kColsPerWG - 1) {
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
#undef GW_REQUEST_SLICE
#undef GW_CONSUME_SLICE
#undef GW_PRODUCE_CHUNK

  if (a.dbg != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = gw_clock();
    if (threadIdx.x == 0 && (int)blockIdx.x < a.dbg_cap) {
      unsigned long long* rec = a.dbg + (size_t)blockIdx.x * 16;
      for (int i = 0; i < 7; ++i) rec[i] = ts[i];
      rec[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      rec[9] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID
      rec[10] = blockIdx.x;
    }
  }
}

template <typename K>
int launch(K kernel, const EdgeArgs& a, void* stream) {
  static DeviceOnce once;  // per template instantiation and device
  static const int lds_bytes = kEdgeLdsBytes + GW_TUNE("GW_EDGE_LDS_PAD", 0);  // tuning aid: > 15 KiB of padding forces one workgroup per CU
  if (once.first()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = (a.n_cols + kColsPerWG - 1) / kColsPerWG;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds_bytes, (hipStream_t)stream, a);
  return check_launch("edge_kernel launch");
}

inline bool is_raw(const gw_operand* op) { return op->k > 0 && !op->projected; }
inline bool is_proj(const gw_operand* op) { return op->k > 0 && op->projected; }

}  // namespace

namespace gw {

bool edge_fast_eligible(const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in, const gw_mlp_weights* w) {
  const gw_operand* ops[3] = {x_src, x_dst, e_in};
  int n_raw = 0, n_proj = 0;
  for (int i = 0; i < 3; ++i) {
    n_raw += is_raw(ops[i]) ? 1 : 0;
    n_proj += is_proj(ops[i]) ? 1 : 0;
  }
  if (n_raw > 1 || n_proj < 1) return false;
  if (w->n_mid < 1 || !w->ln_gamma) return false;  // (no LayerNorm: the general kernel)
  if (w->ln_width > 0 && w->ln_width != 256) return false;  // zero-padded narrow models: masked statistics live in the general kernel
  static int impl = -1;  // GW_EDGE_IMPL=0 forces the general chain kernel (A/B measurements, tests of both paths)
  if (impl < 0) impl = GW_TUNE("GW_EDGE_IMPL", 1);
  return impl != 0;
}

// ---- deterministic segment sums: carry records -> totals, one workgroup per chain start, in tile order ----
__global__ __launch_bounds__(256) void segment_fixup_kernel(long long n_tiles, const float* __restrict__ carry, float* __restrict__ agg) {
  const long long t = blockIdx.x;
  const float* rec = carry + (size_t)t * kCarryFloats;
  const int row = __float_as_int(rec[513]);
  if (row < 0) return;  // no segment starts in this tile and leaves it
  const int f = threadIdx.x;
  float acc = rec[256 + f];
  for (long long u = t + 1; u < n_tiles; ++u) {
    const float* r = carry + (size_t)u * kCarryFloats;
    if (__float_as_int(r[512]) != row) break;  // (cannot happen for a well-formed chain; keeps a malformed one finite)
    acc += r[f];
    if (__float_as_int(r[514]) != 1) break;  // the segment ends in tile u
  }
  agg[(size_t)row * 256 + f] = acc;
}

size_t segment_carry_bytes(int64_t n_tiles) { return (size_t)n_tiles * kCarryFloats * sizeof(float); }

int segment_fixup_launch(int64_t n_tiles, const float* carry, float* agg, void* stream) {
  if (n_tiles <= 0) return GW_OK;
  hipLaunchKernelGGL(segment_fixup_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, (long long)n_tiles, carry, agg);
  return check_launch("segment_fixup_kernel launch");
}

size_t edge_fast_carry_bytes(int32_t batch, int32_t n_edges) {
  return segment_carry_bytes(((int64_t)batch * n_edges + kColsPerWG - 1) / kColsPerWG);
}

int edge_fast_launch(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst, const gw_operand* x_src,
                     const gw_operand* x_dst, const gw_operand* e_in, const gw_operand* e_res, const gw_mlp_weights* w,
                     float* e_out, float* agg, int32_t n_dst, const gw_activation_save* save, float* carry, void* stream) {
  EdgeArgs a;
  memset(&a, 0, sizeof(a));
  a.n_cols = batch * n_edges;
  a.n_edges = n_edges;
  a.n_dst = n_dst;
  a.src = src;
  a.dst = dst;
  const gw_operand* ops[3] = {x_src, x_dst, e_in};
  int n_proj = 0;
  bool raw = false;
  for (int i = 0; i < 3; ++i) {
    if (is_raw(ops[i])) {
      raw = true;
      a.raw_ptr = ops[i]->ptr;
      a.raw_rows_pb = ops[i]->rows_per_batch;
      a.raw_ld = ops[i]->ld;
      a.raw_kind = i;
      a.w_raw = w->w1[i];
    } else if (is_proj(ops[i])) {
      a.p_ptr[n_proj] = ops[i]->ptr;
      a.p_rows_pb[n_proj] = ops[i]->rows_per_batch;
      a.p_ld[n_proj] = ops[i]->ld;
      a.p_kind[n_proj] = i;
      ++n_proj;
    }
  }
  a.b1 = w->b1;
  a.w_mid = w->w_mid;
  a.b_mid = w->b_mid;
  a.n_mid = w->n_mid;
  a.w_out = w->w_out;
  a.b_out = w->b_out;
  a.gamma = w->ln_gamma;
  a.beta = w->ln_beta;
  a.res_ptr = e_res->ptr;
  a.res_rows_pb = e_res->rows_per_batch;
  a.res_ld = e_res->ld;
  a.e_out = e_out;
  a.agg = agg;
  a.carry = carry;
  if (save) {
    a.save_h = save->hidden;
    a.save_stride = save->hidden_stride;
    a.save_ld = save->hidden_ld;
    a.save_y = save->pre_norm;
  }
  if (g_dbg != nullptr && g_dbg_kind == 1) {
    a.dbg = g_dbg;
    a.dbg_cap = g_dbg_cap;
  }
  {
    static int skip = -1;
    if (skip < 0) skip = GW_TUNE("GW_EDGE_SKIP", 0);
    a.skip = skip;
    static const int dma6 = GW_TUNE("GW_EDGE_DMA6", 0);
    a.dma6 = dma6;
  }
  {
    static int stagger_override = -2;
    if (stagger_override == -2) stagger_override = GW_TUNE("GW_STAGGER", -1);
    const int passes = 1 + a.n_mid + (raw ? 1 : 0);
    a.stagger = stagger_override >= 0 ? stagger_override * passes : 2 * passes + 2;
    if ((a.n_cols + kColsPerWG - 1) / kColsPerWG <= 256) a.stagger = 0;
  }
  {
    static int xcd_map = -1;  // GW_XCD_MAP=0: workgroup i takes tile i (A/B measurements)
    if (xcd_map < 0) xcd_map = GW_TUNE("GW_XCD_MAP", 1);
    const int tiles = (a.n_cols + kColsPerWG - 1) / kColsPerWG;
    a.xcd_base = (xcd_map != 0 && tiles >= 64) ? tiles / 8 : 0;
    a.xcd_rem = tiles % 8;
  }
  int rc;
  if (raw) rc = n_proj == 1 ? launch(edge_kernel<true, 1>, a, stream) : launch(edge_kernel<true, 2>, a, stream);
  else if (n_proj == 1) rc = launch(edge_kernel<false, 1>, a, stream);
  else if (n_proj == 2) rc = launch(edge_kernel<false, 2>, a, stream);
  else rc = launch(edge_kernel<false, 3>, a, stream);
  if (rc != GW_OK || carry == nullptr) return rc;
  return segment_fixup_launch((a.n_cols + kColsPerWG - 1) / kColsPerWG, carry, agg, stream);
}

}  // namespace gw
