// gw_noders.hip - row-split form of the node update (graph_net_block.py:184-193, the NodeProcessor MLP behind scatter_sum)
// for MESH-SIZED launches: a few thousand rows, i.e. fewer 64-column workgroups than the chip has CUs.
//
// Why: chain_kernel / chainx3_kernel give one wave the whole 256 x 256 pass of its 16 columns.  A node update on the 5 882 mesh
// nodes of one sample is 92 workgroups x 4 waves: 184 of 256 CUs at batch 2, ONE wave per SIMD, each walking six passes alone
// (fp32: 6 144 MFMAs x 32 cycles = 82 us of a 131 us launch; bf16x3: 6.1 k cycles of MFMA in a 23 k-cycle pass - latency bound).
// Cutting the launch into narrower column tiles does not shorten a wave's chain, it multiplies the weight stream.
//
// Here the ROW TILES of a column group are split over four waves instead: wave (g, r) of a workgroup owns column group g
// (16 columns) and row tiles 4r .. 4r+3 of every layer.  It keeps the layer's whole B operand (all 256 input features of its 16
// columns) in registers, reads only its quarter of each weight K-step from LDS (the packed streams are already grouped by four
// row tiles: one ds_read_b128 per K-step in fp32) and issues a quarter of the MFMAs.  Between layers the four waves of a group
// exchange their quarter of the new activations through 16 KiB of LDS - in the transposed scheme a lane's accumulator registers
// ARE its B-operand registers of the next layer, so the exchange is lane-to-lane (conflict-free 16-byte accesses, no shuffles).
// Before LayerNorm the whole row is exchanged once more: every wave computes the statistics in the order of the 64-column
// kernels (bitwise their rows).  A workgroup is CG column groups (CG = 1, 2, 3 -> 4, 8, 12
// waves sharing ONE weight stream): 736 column groups of a batch-2 mesh become 246 workgroups of 48 columns - every CU of the chip
// carries three waves per SIMD and streams the 1.5 MB of weights once, instead of 184 CUs carrying one wave per SIMD.
// Products and LayerNorm sums are added in the order of chain_kernel / chainx3_kernel: bitwise the same rows.
//
// Launch kinds: node update (+ POST products of the next block's layer-1 slices, + zero fill of the next aggregate); the
// aggregate a raw 256-wide fp32 table, the node operand raw, already projected (a cached product: gather-add) or absent; one
// middle layer, LayerNorm over 256 features or none, inference (no activation saves); fp32 (node_rs_kernel) and split-operand
// bf16x3 weights (node_rs3_kernel, arithmetic of gw_split.hip).  Anything else stays on chain_kernel / chainx3_kernel
// (gw_node_update_forward decides).
//
// LDS: two 64 KiB weight buffers; the exchange buffer (16 KiB per column group) lies OVER the one that is free between two passes.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

// phase clocks and A/B switches: tuning builds only (gw_debug_timestamps, scripts/gpu_timeline_rs.py; GW_RS_TUNE bit 0 = no L2 prefetch)
#ifdef GW_TUNING
#define RS_STAMP(i) \
  if (a.dbg != nullptr) { ts[i] = gw::gw_clock(); }
#define RS_TUNE(a) ((a).tune16)
#define RS_CLOCKS                                           \
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};      \
  unsigned long long waited_[2] = {0, 0};
#define RS_WAITED (a.dbg != nullptr ? waited_ : nullptr)
#define RS_RECORD                                                                   \
  if (a.dbg != nullptr) {                                                           \
    ts[6] = gw::gw_clock();                                                         \
    if (threadIdx.x == 0 && (int)blockIdx.x < a.dbg_cap) {                          \
      unsigned long long* rec = a.dbg + (size_t)blockIdx.x * 16;                    \
      for (int i = 0; i < 7; ++i) rec[i] = ts[i];                                   \
      rec[10] = blockIdx.x;                                                         \
      rec[11] = waited_[0]; /* middle + output passes: own DMA pieces landing */    \
      rec[12] = waited_[1]; /* ... and the workgroup barrier */                     \
    }                                                                               \
  }
#else
#define RS_STAMP(i)
#define RS_TUNE(a) 0
#define RS_CLOCKS
#define RS_WAITED nullptr
#define RS_RECORD
#endif

// the chunk hand-over: own DMA pieces landed, then the workgroup barrier (tuning builds clock the two waits per wave)
#ifdef GW_TUNING
#define RS_WAIT_BARRIER()                                      \
  {                                                            \
    unsigned long long t0w = 0, t1w = 0;                       \
    if (waited != nullptr) t0w = gw::gw_clock();               \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           \
    if (waited != nullptr) t1w = gw::gw_clock();               \
    __syncthreads();                                           \
    if (waited != nullptr) {                                   \
      waited[0] += t1w - t0w;                                  \
      waited[1] += gw::gw_clock() - t1w;                       \
    }                                                          \
  }
#else
#define RS_WAIT_BARRIER()                            \
  {                                                  \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    __syncthreads();                                 \
  }
#endif

constexpr int kXbufFloats = 16 * 64 * 4;  // exchange buffer of one column group: 16 row tiles x 64 lanes x 4 floats = 16 KiB
constexpr int kStepFloats = 4 * 256;      // one fp32 K-step of 16 row tiles in the packed stream (gw_pack_linear)

// Weight chunks: 64 KiB (16 fp32 K-steps / 2 bf16x3 K-steps of all 16 row tiles), double buffered - half the hand-overs of the
// 32 KiB chunks of the 64-column kernels: with 12 waves behind one barrier every hand-over idles the matrix pipes for ~1.5 k
// cycles (barrier, LDS latency of the first fragments, the oldest wave of a SIMD finishing first), and a bf16x3 chunk of one
// K-step is only 576 cycles of MFMAs per SIMD.  The exchange buffer always lies over the buffer that is free between two passes.
constexpr int kRsBufFloats = 2 * kLdsBufFloats;  // 64 KiB
constexpr int kRsSteps = 2 * kChunkSteps;        // fp32 K-steps per chunk
constexpr int rs_lds_bytes(int) { return 2 * kRsBufFloats * 4; }

// round i of a wave's DMA pieces of a 64 KiB chunk (piece wave + i NW of 64), issued from between the MFMA groups of a pass;
// src == nullptr: nothing follows.  Only a last, partial round tests the wave index (12 waves: pieces 60 .. 63).
constexpr int rs_rounds(int nw) { return (64 + nw - 1) / nw; }
template <int NW>
__device__ __forceinline__ void issue_piece64(const char* __restrict__ src, unsigned lds0, int i, int lane, int wave) {
  const int pc = wave + i * NW;
  if ((i + 1) * NW <= 64 || pc < 64)
    glds16_asm_s((const float*)(src + (size_t)pc * 1024), (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)pc * 1024u));
}
// the rounds due behind unit u of a chunk whose first `span` units carry the DMA (compile-time after unrolling)
template <int NW>
__device__ __forceinline__ void issue_due(const char* __restrict__ src, bool go, unsigned lds0, int u, int span, int lane, int wave) {
  constexpr int R = rs_rounds(NW);
  if (u >= span) return;
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (i >= u * R / span && i < (u + 1) * R / span)
      if (go) issue_piece64<NW>(src, lds0, i, lane, wave);
}

// Every wave DMAs its share of one 64 KiB chunk (64 pieces of 1 KiB) of a packed weight stream into an LDS buffer.  The trip
// count is a compile-time constant and only a last, partial round tests the wave index (12 waves: pieces 60 .. 63) - no loop
// branch inside the matrix passes.
template <int NW>
__device__ __forceinline__ void issue_chunk64(const char* __restrict__ g, const void* ldsbuf, int lane, int wave) {
  constexpr int NP = 64;
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)ldsbuf;
#pragma unroll
  for (int i = 0; i < (NP + NW - 1) / NW; ++i) {
    const int pc = wave + i * NW;
    if ((i + 1) * NW <= NP || pc < NP)
      glds16_asm_s((const float*)(g + (size_t)pc * 1024), (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)pc * 1024u));
  }
}

__device__ __forceinline__ const float* operand_row(const float* ptr, const int* idx, int rows_pb, int ld, int b, int k) {
  const int rr = idx ? ldgi(idx + k) : k;
  return ptr + ((size_t)b * (size_t)rows_pb + (size_t)rr) * (size_t)ld;
}

// what the two kernels share: who this wave is, where its column lives
struct RsWave {
  int lane, wave, r, g, j, q, c, b, k;
  bool valid;
};
template <int CG>
__device__ __forceinline__ RsWave rs_wave(const ChainArgs& a) {
  RsWave w;
  w.lane = threadIdx.x & 63;
  w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  w.r = w.wave & 3;   // row quarter: row tiles 4r .. 4r+3
  w.g = w.wave >> 2;  // column group
  w.j = w.lane & 15;
  w.q = w.lane >> 4;
  const int c_raw = blockIdx.x * (16 * CG) + w.g * 16 + w.j;
  w.valid = c_raw < a.n_cols;
  w.c = w.valid ? c_raw : a.n_cols - 1;
  w.b = w.c / a.cols_per_batch;
  w.k = w.c - w.b * a.cols_per_batch;
  return w;
}

// LayerNorm (eps 1e-5, biased variance) + residual + store of this wave's quarter (+ zero fill of the next aggregate).
// `full` holds the pre-LayerNorm row of ALL sixteen row tiles (exchanged through LDS): every wave of a column group computes
// the statistics itself, in the order of the 64-column kernels (X3: chainx3_kernel's expressions, else chain_kernel's) - the new
// rows are bitwise those of the other kernel form, so a checkpointed segment replayed on it sees the values of its first run.
// Returns mean / rstd for rs_full_tile (the B operand of the POST products).
template <bool X3>
__device__ __forceinline__ void rs_epilogue(const ChainArgs& a, const RsWave& w, f32x4 (&o)[4], const f32x4 (&rres)[4], const float (&full)[64],
                                            float& mean_out, float& rstd_out) {
  const int r = w.r, q = w.q;
  float mean = 0.f, rstd = 1.f;
  if (a.gamma != nullptr) {
    const int nfeat = a.ln_width;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += (full[4 * t] + full[4 * t + 1]) + (full[4 * t + 2] + full[4 * t + 3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    float v = 0.f, inv_n;
    if (X3) {
      inv_n = 1.0f / (float)nfeat;
      const bool all = nfeat == 256;
      mean = s * inv_n;
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = full[4 * t + i] - mean;
          v += (all || 16 * t + 4 * q + i < nfeat) ? d * d : 0.f;
        }
    } else {
      const bool narrow = nfeat != 256;
      inv_n = narrow ? 1.0f / (float)nfeat : 1.0f / 256;
      mean = s * inv_n;
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = (narrow && 16 * t + 4 * q + i >= nfeat) ? 0.f : full[4 * t + i] - mean;
          v += d * d;
        }
    }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    rstd = 1.0f / sqrtf(v * inv_n + 1e-5f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 gm = ldg4(a.gamma + 16 * (4 * r + t) + 4 * q);
      const f32x4 bt = ldg4(a.beta + 16 * (4 * r + t) + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[t][i] = (o[t][i] - mean) * rstd * gm[i] + bt[i];
    }
  }
  if (a.res_ptr != nullptr) {
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] += rres[t];
  }
  if (w.valid) {
    float* orow = a.out + (size_t)w.c * (size_t)a.out_ld;
#pragma unroll
    for (int t = 0; t < 4; ++t) stg4(orow + 16 * (4 * r + t) + 4 * q, o[t]);
    if (a.zero_rows != nullptr) {  // the aggregate buffer of the next block's edge update, zero-filled on the side
      float* zrow = a.zero_rows + (size_t)w.c * 256;
#pragma unroll
      for (int t = 0; t < 4; ++t) stg4(zrow + 16 * (4 * r + t) + 4 * q, f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
  mean_out = mean;
  rstd_out = rstd;
}

// Row tile t of the whole new row (LayerNorm + residual of the pre-LayerNorm values every wave holds): the B operand of the POST
// products, recomputed by each wave from the same values with the same expressions as the quarter's owner - bitwise its row.
__device__ __forceinline__ f32x4 rs_full_tile(const ChainArgs& a, const RsWave& w, f32x4 v, int t, float mean, float rstd, const float* rrow) {
  if (a.gamma != nullptr) {
    const f32x4 gm = ldg4(a.gamma + 16 * t + 4 * w.q);
    const f32x4 bt = ldg4(a.beta + 16 * t + 4 * w.q);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (v[i] - mean) * rstd * gm[i] + bt[i];
  }
  if (rrow != nullptr) v += ldg4(rrow + 16 * t + 4 * w.q);
  return v;
}

// ================================================ fp32 (arithmetic of chain_kernel) ================================================

// (k(s, q) = 16 (s >> 2) + 4 q + (s & 3))
// in[16c .. 16c+15] <- row[k(s, q)] for the K-steps of chunk c
__device__ __forceinline__ void load_slice16(float (&in)[64], const float* __restrict__ row, int c, int q) {
#pragma unroll
  for (int i = 4 * c; i < 4 * c + 4; ++i) {
    const f32x4 v = ldg4(row + 16 * i + 4 * q);
    in[4 * i + 0] = v.x;
    in[4 * i + 1] = v.y;
    in[4 * i + 2] = v.z;
    in[4 * i + 3] = v.w;
  }
}

// One 256-deep pass of this wave's four row tiles: acc[t] += W[16 (4r + t) .., k] . in[k], four chunks of 16 K-steps.
// Protocol of chain_kernel's mma_pass: the first chunk of the pass is already on its way into buffer `parity`; the first chunk
// of the next pass is issued while the last one computes.  Behind the barrier a wave first requests its first fragment, THEN
// issues its DMA pieces (each blocks the wave for 60 - 180 cycles: the fragment's LDS latency hides under them).
// RELOAD: the 16 operand registers a chunk has consumed are refilled, one chunk later, with the same K slice of the next
// layer-1 operand (`next_row`); the slice the previous pass consumed last is refilled during chunk 0 (`tail_row`).
template <int NW, bool RELOAD>
__device__ __forceinline__ void rs_pass(f32x4 (&acc)[4], float (&in)[64], const float* __restrict__ gw, const float* __restrict__ next_gw,
                                        float* lds, int& parity, int lane, int wave, int r, const float* __restrict__ tail_row, bool do_tail,
                                        const float* __restrict__ next_row, bool do_next, int q, unsigned long long* waited = nullptr) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    RS_WAIT_BARRIER();  // chunk c has landed for every wave; nobody still reads the other buffer (weights or exchanged rows)
    const float* buf = lds + parity * kRsBufFloats + r * 256 + lane * 4;
    f32x4 a_cur = *(const f32x4*)buf;
    const char* nsrc = c + 1 < 4 ? (const char*)(gw + (size_t)(c + 1) * kRsSteps * kStepFloats) : (const char*)next_gw;
    const bool go = c + 1 < 4 || next_gw != nullptr;  // (inside a pass: a compile-time constant, no branch)
    const unsigned nlds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + (parity ^ 1) * kRsBufFloats);
    if (RELOAD) {
      if (c == 0) {
        if (do_tail) load_slice16(in, tail_row, 3, q);
      } else if (do_next) {
        load_slice16(in, next_row, c - 1, q);
      }
    }
#pragma unroll
    for (int s = 0; s < kRsSteps; ++s) {
      f32x4 a_nxt = a_cur;
      if (s + 1 < kRsSteps) a_nxt = *(const f32x4*)(buf + (s + 1) * kStepFloats);
      const float b = in[c * kRsSteps + s];
      __builtin_amdgcn_sched_barrier(0);  // the LDS read of step s + 1 stays ahead of the MFMAs of step s
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t], b, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // this wave's DMA pieces of the next chunk, spread over the first three quarters of the chunk's K-steps: an issue blocks
      // the wave for 60 - 180 cycles - spread out, the other waves of the SIMD own the matrix pipe meanwhile (all pieces right
      // behind the barrier: every wave of the SIMD blocked at once, ~800 idle cycles per chunk)
      issue_due<NW>(nsrc, go, nlds, s, 12, lane, wave);
      a_cur = a_nxt;
    }
    parity ^= 1;
  }
}

// The four waves of a column group publish their quarter of a layer's output (RELU: its relu) and every wave reads the
// whole 256-feature B operand of the next layer back: in[4 T + i] = value of row tile T, register i of this lane.
// xg lies in the weight buffer the pass has just finished with - slower waves may still be reading its last chunk.
template <bool RELU>
__device__ __forceinline__ void exchange(float (&in)[64], const f32x4 (&acc)[4], float* xg, int r, int lane) {
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 v = acc[t];
    if (RELU) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
    *(f32x4*)(xg + ((4 * r + t) * 64 + lane) * 4) = v;
  }
  __syncthreads();
#pragma unroll
  for (int T = 0; T < 16; ++T) {
    const f32x4 v = *(const f32x4*)(xg + (T * 64 + lane) * 4);
    in[4 * T + 0] = v.x;
    in[4 * T + 1] = v.y;
    in[4 * T + 2] = v.z;
    in[4 * T + 3] = v.w;
  }
}

template <int CG>
__global__ __launch_bounds__(CG * 256, CG) void node_rs_kernel(const ChainArgs a) {
  constexpr int NW = 4 * CG;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const RsWave w = rs_wave<CG>(a);
  const int lane = w.lane, wave = w.wave, r = w.r, q = w.q;
  int parity = 0;
  RS_CLOCKS
  RS_STAMP(0)
  // exchange buffer of this wave's column group: over the weight buffer the finished pass consumed last
  auto xg = [&]() -> float* { return lds + (parity ^ 1) * kRsBufFloats + w.g * kXbufFloats; };

  const bool raw0 = a.seg_k[0] > 0 && a.seg_proj[0] == 0;  // x . Wx^T is a matrix pass
  const bool prj0 = a.seg_k[0] > 0 && a.seg_proj[0] != 0;  // ... or a cached product row (gather-add); neither: x == 0
  issue_chunk64<NW>((const char*)(raw0 ? a.w1[0] : a.w1[1]), lds, lane, wave);
  if (!(RS_TUNE(a) & 1)) {  // the rest of the stream into this XCD's L2 (gw_device.hpp); scratch: the second weight buffer, free until the first barrier
    const char* mats[8] = {(const char*)(raw0 ? a.w1[0] : a.w1[1]), (const char*)(raw0 ? a.w1[1] : nullptr), (const char*)a.w_mid,
                           (const char*)a.w_out, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < a.n_post && i < 4; ++i) mats[4 + i] = (const char*)a.proj_w[i];
    l2_prefetch_stream(mats, 8, lane, wave, (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + kRsBufFloats));
  }

  const float* arow = operand_row(a.seg_ptr[1], a.seg_idx[1], a.seg_rows_pb[1], a.seg_ld[1], w.b, w.k);
  float in[64];
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b1 + 16 * (4 * r + t) + 4 * q);

  // ---- layer 1: cat[x, agg] . W1^T = x . Wx^T + agg . Wa^T ----
  if (raw0) {  // the aggregate rows stream in under the pass of x
    const float* xrow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) load_slice16(in, xrow, cc, q);
    rs_pass<NW, true>(acc, in, a.w1[0], a.w1[1], lds, parity, lane, wave, r, nullptr, false, arow, true, q);
    rs_pass<NW, true>(acc, in, a.w1[1], a.w_mid, lds, parity, lane, wave, r, arow, true, nullptr, false, q);
  } else {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) load_slice16(in, arow, cc, q);
    f32x4 pr[4];
    if (prj0) {  // the cached product rows: requested under the pass, added behind it as chain_kernel does ((b1 + agg . Wa^T) + P)
      const float* prow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
      for (int t = 0; t < 4; ++t) pr[t] = ldg4(prow + 16 * (4 * r + t) + 4 * q);
    }
    rs_pass<NW, false>(acc, in, a.w1[1], a.w_mid, lds, parity, lane, wave, r, nullptr, false, nullptr, false, q);
    if (prj0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] += pr[t];
    }
  }

  RS_STAMP(1)
  // ---- middle layer ----
  exchange<true>(in, acc, xg(), r, lane);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b_mid + 16 * (4 * r + t) + 4 * q);
  rs_pass<NW, false>(acc, in, a.w_mid, a.w_out, lds, parity, lane, wave, r, nullptr, false, nullptr, false, q, RS_WAITED);
  RS_STAMP(2)

  // ---- output layer (the residual rows are requested underneath it) ----
  exchange<true>(in, acc, xg(), r, lane);
  f32x4 o[4], rres[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = ldg4(a.b_out + 16 * (4 * r + t) + 4 * q);
  if (a.res_ptr != nullptr) {
    const float* rrow = operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, w.b, w.k);
#pragma unroll
    for (int t = 0; t < 4; ++t) rres[t] = ldg4(rrow + 16 * (4 * r + t) + 4 * q);
  }
  rs_pass<NW, false>(o, in, a.w_out, a.n_post > 0 ? a.proj_w[0] : nullptr, lds, parity, lane, wave, r, nullptr, false, nullptr, false, q,
                     RS_WAITED);
  RS_STAMP(3)

  // ---- LayerNorm + residual + store; the whole pre-LayerNorm row travels through LDS once (statistics, POST operand) ----
  if (a.gamma != nullptr || a.n_post > 0) exchange<false>(in, o, xg(), r, lane);
  float mean, rstd;
  rs_epilogue<false>(a, w, o, rres, in, mean, rstd);
  if (a.n_post > 0) {  // the whole new row in registers
    const float* rrow = a.res_ptr != nullptr ? operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, w.b, w.k) : nullptr;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const f32x4 v = rs_full_tile(a, w, f32x4{in[4 * t], in[4 * t + 1], in[4 * t + 2], in[4 * t + 3]}, t, mean, rstd, rrow);
      in[4 * t + 0] = v.x;
      in[4 * t + 1] = v.y;
      in[4 * t + 2] = v.z;
      in[4 * t + 3] = v.w;
    }
  }
  RS_STAMP(4)

  // ---- POST: the next block's layer-1 products of the new rows (ChainArgs) ----
  if (a.n_post > 0) {
#pragma unroll 1
    for (int sl = 0; sl < a.n_post; ++sl) {
      f32x4 pacc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) pacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* nx = sl + 1 < a.n_post ? a.proj_w[sl + 1] : nullptr;
      rs_pass<NW, false>(pacc, in, a.proj_w[sl], nx, lds, parity, lane, wave, r, nullptr, false, nullptr, false, q);
      if (w.valid) {
        float* prow = a.proj_out[sl] + (size_t)w.c * 256;
#pragma unroll
        for (int t = 0; t < 4; ++t) stg4(prow + 16 * (4 * r + t) + 4 * q, pacc[t]);
      }
    }
  }
  RS_STAMP(5)
  RS_RECORD
}

// ============================== bf16x3: split operands on v_mfma_f32_16x16x32_bf16 (arithmetic of gw_split.hip) ==============================
// Packed stream (gw_pack_linear_bf16x3): per 32-wide K-step the hi fragments of the 16 row tiles, then their lo fragments,
// 1 KiB each = 32 KiB = one chunk.  K order k(s, q, i) = 32 s + 16 (i >> 2) + 4 q + (i & 3): K-step s of a layer's B operand is
// split(acc[2s], acc[2s + 1]) of the layer before, so row quarter r owns K-steps 2r, 2r + 1 of the next operand.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kStepBytes3 = 32768;       // one K-step: hi + lo fragments of the 16 row tiles
constexpr int kChunkBytes3 = 2 * kStepBytes3;  // one chunk buffer

__device__ __forceinline__ void split8(f32x4 a, f32x4 b, bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 ha = (__bf16)a[i];
    h[i] = ha;
    l[i] = (__bf16)(a[i] - (float)ha);
    const __bf16 hb = (__bf16)b[i];
    h[4 + i] = hb;
    l[4 + i] = (__bf16)(b[i] - (float)hb);
  }
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }

// (bh, bl)[s] <- split(row[k(s, q, i)]) of a full fp32 row
__device__ __forceinline__ void load_slice3(bf16x8& h, bf16x8& l, const float* __restrict__ row, int s, int q) {
  split8(ldg4(row + 32 * s + 4 * q), ldg4(row + 32 * s + 16 + 4 * q), h, l);
}

// One 256-deep pass: 4 chunks of two K-steps (64 KiB); per K-step this wave reads the hi / lo fragments of its four row tiles
// (8 x 1 KiB) and issues 12 MFMAs, term-major (hi.hi, hi.lo, lo.hi: the other tiles' MFMAs between two on one accumulator).
// Behind the barrier the first K-step's fragments are requested BEFORE the wave's DMA pieces of the next chunk are issued.
template <int NW>
__device__ __forceinline__ void rs3_pass(f32x4 (&acc)[4], const bf16x8 (&bh)[8], const bf16x8 (&bl)[8], const char* __restrict__ gw,
                                         const char* __restrict__ next_gw, char* lds, int& parity, int lane, int wave, int r,
                                         unsigned long long* waited = nullptr) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    RS_WAIT_BARRIER();
    const char* buf = lds + parity * kChunkBytes3 + (4 * r) * 1024 + lane * 16;
    // fragments in flight: hi + lo of this K-step and hi of the next (48 registers); lo of the next follows once hi is spent.
    // The wave's DMA pieces of the next chunk go out behind the first four of the six MFMA groups (see rs_pass).
    const char* nsrc = c + 1 < 4 ? gw + (size_t)(c + 1) * kChunkBytes3 : next_gw;
    const bool go = c + 1 < 4 || next_gw != nullptr;
    const unsigned nlds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + (parity ^ 1) * kChunkBytes3);
    bf16x8 h0[4], l0[4], h1[4], l1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) h0[t] = *(const bf16x8*)(buf + t * 1024);
#pragma unroll
    for (int t = 0; t < 4; ++t) l0[t] = *(const bf16x8*)(buf + 16384 + t * 1024);
#pragma unroll
    for (int t = 0; t < 4; ++t) h1[t] = *(const bf16x8*)(buf + kStepBytes3 + t * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0[t], bh[2 * c], acc[t], 0, 0, 0);
    issue_due<NW>(nsrc, go, nlds, 0, 4, lane, wave);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0[t], bl[2 * c], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) l1[t] = *(const bf16x8*)(buf + kStepBytes3 + 16384 + t * 1024);
    issue_due<NW>(nsrc, go, nlds, 1, 4, lane, wave);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0[t], bh[2 * c], acc[t], 0, 0, 0);
    issue_due<NW>(nsrc, go, nlds, 2, 4, lane, wave);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1[t], bh[2 * c + 1], acc[t], 0, 0, 0);
    issue_due<NW>(nsrc, go, nlds, 3, 4, lane, wave);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1[t], bl[2 * c + 1], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1[t], bh[2 * c + 1], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    parity ^= 1;
  }
}

// exchange of the split activations: row quarter r publishes K-steps 2r, 2r + 1 (hi, lo), every wave reads all eight back
// (xg lies over the weight buffer the finished pass consumed last: slower waves may still be reading it)
template <bool RELU>
__device__ __forceinline__ void exchange3(bf16x8 (&bh)[8], bf16x8 (&bl)[8], const f32x4 (&acc)[4], char* xg, int r, int lane) {
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    bf16x8 h, l;
    if (RELU)
      split8(relu4(acc[2 * u]), relu4(acc[2 * u + 1]), h, l);
    else
      split8(acc[2 * u], acc[2 * u + 1], h, l);
    const int s = 2 * r + u;
    *(bf16x8*)(xg + ((2 * s) * 64 + lane) * 16) = h;
    *(bf16x8*)(xg + ((2 * s + 1) * 64 + lane) * 16) = l;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    bh[s] = *(const bf16x8*)(xg + ((2 * s) * 64 + lane) * 16);
    bl[s] = *(const bf16x8*)(xg + ((2 * s + 1) * 64 + lane) * 16);
  }
}

template <int CG>
__global__ __launch_bounds__(CG * 256, CG) void node_rs3_kernel(const ChainArgs a) {
  constexpr int NW = 4 * CG;
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  const RsWave w = rs_wave<CG>(a);
  const int lane = w.lane, wave = w.wave, r = w.r, q = w.q;
  int parity = 0;
  RS_CLOCKS
  RS_STAMP(0)
  auto xg = [&]() -> char* { return lds3 + (parity ^ 1) * kChunkBytes3 + w.g * (kXbufFloats * 4); };

  const bool raw0 = a.seg_k[0] > 0 && a.seg_proj[0] == 0;
  const bool prj0 = a.seg_k[0] > 0 && a.seg_proj[0] != 0;
  const char* w1x = (const char*)a.w1[0];
  const char* w1a = (const char*)a.w1[1];
  const char* w_mid = (const char*)a.w_mid;
  const char* w_out = (const char*)a.w_out;
  issue_chunk64<NW>(raw0 ? w1x : w1a, lds3, lane, wave);
  if (!(RS_TUNE(a) & 1)) {  // the rest of the stream into this XCD's L2 (gw_device.hpp); scratch: the second weight buffer, free until the first barrier
    const char* mats[8] = {raw0 ? w1x : w1a, raw0 ? w1a : nullptr, w_mid, w_out, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < a.n_post && i < 4; ++i) mats[4 + i] = (const char*)a.proj_w[i];
    l2_prefetch_stream(mats, 8, lane, wave, (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds3 + kChunkBytes3));
  }

  const float* arow = operand_row(a.seg_ptr[1], a.seg_idx[1], a.seg_rows_pb[1], a.seg_ld[1], w.b, w.k);
  bf16x8 bh[8], bl[8];
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b1 + 16 * (4 * r + t) + 4 * q);

  if (raw0) {
    const float* xrow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
    for (int s = 0; s < 8; ++s) load_slice3(bh[s], bl[s], xrow, s, q);
    rs3_pass<NW>(acc, bh, bl, w1x, w1a, lds3, parity, lane, wave, r);
  } else if (prj0) {  // chainx3_kernel adds the product rows in front of the matrix passes: (b1 + P) + agg . Wa^T
    const float* prow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += ldg4(prow + 16 * (4 * r + t) + 4 * q);
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) load_slice3(bh[s], bl[s], arow, s, q);
  rs3_pass<NW>(acc, bh, bl, w1a, w_mid, lds3, parity, lane, wave, r);

  RS_STAMP(1)
  exchange3<true>(bh, bl, acc, xg(), r, lane);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b_mid + 16 * (4 * r + t) + 4 * q);
  rs3_pass<NW>(acc, bh, bl, w_mid, w_out, lds3, parity, lane, wave, r, RS_WAITED);
  RS_STAMP(2)

  exchange3<true>(bh, bl, acc, xg(), r, lane);
  f32x4 o[4], rres[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = ldg4(a.b_out + 16 * (4 * r + t) + 4 * q);
  rs3_pass<NW>(o, bh, bl, w_out, a.n_post > 0 ? (const char*)a.proj_w[0] : nullptr, lds3, parity, lane, wave, r, RS_WAITED);
  RS_STAMP(3)
  if (a.res_ptr != nullptr) {  // (requested here, not under the pass: 16 registers the pass does not have at three waves per SIMD)
    const float* rrow = operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, w.b, w.k);
#pragma unroll
    for (int t = 0; t < 4; ++t) rres[t] = ldg4(rrow + 16 * (4 * r + t) + 4 * q);
  }

  {
    float full[64];
    if (a.gamma != nullptr || a.n_post > 0) exchange<false>(full, o, (float*)xg(), r, lane);
    float mean, rstd;
    rs_epilogue<true>(a, w, o, rres, full, mean, rstd);
    if (a.n_post > 0) {  // the whole new row as the split B operand of the POST products, one K-step (two row tiles) at a time
      const float* rrow = a.res_ptr != nullptr ? operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, w.b, w.k) : nullptr;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const f32x4 v0 = rs_full_tile(a, w, f32x4{full[8 * s8], full[8 * s8 + 1], full[8 * s8 + 2], full[8 * s8 + 3]}, 2 * s8, mean, rstd, rrow);
        const f32x4 v1 =
            rs_full_tile(a, w, f32x4{full[8 * s8 + 4], full[8 * s8 + 5], full[8 * s8 + 6], full[8 * s8 + 7]}, 2 * s8 + 1, mean, rstd, rrow);
        split8(v0, v1, bh[s8], bl[s8]);
      }
    }
  }
  RS_STAMP(4)

  if (a.n_post > 0) {
#pragma unroll 1
    for (int sl = 0; sl < a.n_post; ++sl) {
      f32x4 pacc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) pacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* nx = sl + 1 < a.n_post ? (const char*)a.proj_w[sl + 1] : nullptr;
      rs3_pass<NW>(pacc, bh, bl, (const char*)a.proj_w[sl], nx, lds3, parity, lane, wave, r);
      if (w.valid) {
        float* prow = a.proj_out[sl] + (size_t)w.c * 256;
#pragma unroll
        for (int t = 0; t < 4; ++t) stg4(prow + 16 * (4 * r + t) + 4 * q, pacc[t]);
      }
    }
  }
  RS_STAMP(5)
  RS_RECORD
}

template <int CG, bool X3>
int launch_rs(ChainArgs& a, void* stream) {
  static DeviceOnce once;  // per instantiation and device
  const void* fn = X3 ? (const void*)node_rs3_kernel<CG> : (const void*)node_rs_kernel<CG>;
  if (once.first()) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, rs_lds_bytes(CG));
  const int grid = (a.n_cols + 16 * CG - 1) / (16 * CG);
  if (X3)
    hipLaunchKernelGGL(node_rs3_kernel<CG>, dim3(grid), dim3(CG * 256), rs_lds_bytes(CG), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(node_rs_kernel<CG>, dim3(grid), dim3(CG * 256), rs_lds_bytes(CG), (hipStream_t)stream, a);
  return check_launch("node_rs_kernel launch");
}

}  // namespace

namespace gw {

// Column groups per workgroup for a launch of n_cols columns, 0 = not a mesh-sized launch (more than one round of 48-column
// workgroups: the 64-column kernels with two workgroups per CU are the better shape there).
int node_rs_groups(int64_t n_cols) {
  const int64_t groups = (n_cols + 15) / 16;
  if (groups <= 0 || groups > 3 * 256) return 0;
  const int cg = (int)((groups + 255) / 256);
  return cg < 1 ? 1 : cg;
}

// fp32 rows everywhere (the formats of the fp32 and bf16x3 modes); the caller has checked widths and pointers of the MLP
bool node_rs_eligible(const ChainArgs& a) {
  if (node_rs_groups(a.n_cols) == 0) return false;
  if (a.seg_k[1] != 256 || a.seg_proj[1] || a.seg_half[1] || a.seg_bf16k[1] || a.seg_ptr[1] == nullptr || a.seg_ld[1] % 4 != 0 ||
      a.w1[1] == nullptr)
    return false;
  if (a.seg_k[0] != 0) {
    if (a.seg_k[0] != 256 || a.seg_half[0] || a.seg_bf16k[0] || a.seg_ptr[0] == nullptr || a.seg_ld[0] % 4 != 0) return false;
    if (!a.seg_proj[0] && a.w1[0] == nullptr) return false;
  }
  if (a.seg_k[2] != 0) return false;
  if (a.n_mid != 1 || a.save_h != nullptr || a.save_y != nullptr) return false;
  if (a.gamma != nullptr && a.ln_width != 256) return false;
  if (a.res_ptr != nullptr && a.res_ld % 4 != 0) return false;
  if (a.n_post > 0 && a.out_ld != 256) return false;
  if (a.proj_half) return false;
  return a.out != nullptr && a.out_ld % 4 == 0;
}

int node_rs_launch(ChainArgs& a, bool x3, void* stream) {
  const int cg = node_rs_groups(a.n_cols);
#ifdef GW_TUNING
  if (g_dbg != nullptr && (g_dbg_kind == 2 || g_dbg_kind == 4)) {  // gw_debug_timestamps: node updates
    a.dbg = g_dbg;
    a.dbg_cap = g_dbg_cap;
  }
  static const int rs_tune = GW_TUNE("GW_RS_TUNE", 0);
  a.tune16 = rs_tune;
#endif
  if (x3) {
    switch (cg) {
      case 1: return launch_rs<1, true>(a, stream);
      case 2: return launch_rs<2, true>(a, stream);
      case 3: return launch_rs<3, true>(a, stream);
    }
  } else {
    switch (cg) {
      case 1: return launch_rs<1, false>(a, stream);
      case 2: return launch_rs<2, false>(a, stream);
      case 3: return launch_rs<3, false>(a, stream);
    }
  }
  return set_error(GW_E_UNSUPPORTED, "node_rs_launch: not a mesh-sized launch");
}

}  // namespace gw
