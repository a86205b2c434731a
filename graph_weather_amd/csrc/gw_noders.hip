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
// LayerNorm statistics are combined through LDS in a fixed order.  A workgroup is CG column groups (CG = 1, 2, 3 -> 4, 8, 12
// waves sharing ONE weight stream): 736 column groups of a batch-2 mesh become 246 workgroups of 48 columns - every CU of the chip
// carries three waves per SIMD and streams the 1.5 MB of weights once, instead of 184 CUs carrying one wave per SIMD.
// The products are summed in the order of chain_kernel (bitwise the same MLP outputs before LayerNorm).
//
// Launch kinds: node update (+ POST products of the next block's layer-1 slices, + zero fill of the next aggregate); the
// aggregate a raw 256-wide fp32 table, the node operand raw, already projected (a cached product: gather-add) or absent; one
// middle layer, LayerNorm over 256 features or none, inference (no activation saves); fp32 (node_rs_kernel) and split-operand
// bf16x3 weights (node_rs3_kernel, arithmetic of gw_split.hip).  Anything else stays on chain_kernel / chainx3_kernel
// (gw_node_update_forward decides).
//
// LDS: the two 32 KiB weight buffers; the exchange buffer (16 KiB per column group) lies OVER the weight buffer that is free
// between two passes when it fits (CG <= 2: 65 KiB per workgroup, so a 64-column edge-update workgroup of the other sample's
// stream - or a second row-split workgroup - shares the CU), behind them for CG = 3 (113 KiB).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

constexpr int kXbufFloats = 16 * 64 * 4;  // exchange buffer of one column group: 16 row tiles x 64 lanes x 4 floats = 16 KiB
constexpr int kStepFloats = 4 * 256;      // one fp32 K-step of 16 row tiles in the packed stream (gw_pack_linear)

constexpr bool rs_overlay(int cg) { return cg * kXbufFloats <= kLdsBufFloats; }
constexpr int rs_lds_bytes(int cg) { return (2 * kLdsBufFloats + (rs_overlay(cg) ? 0 : cg * kXbufFloats) + 2 * cg * 64) * 4; }

template <int NW>
__device__ __forceinline__ void issue_chunk_nw(const float* __restrict__ g, int nfloats, float* ldsbuf, int lane, int wave) {
  const int npieces = nfloats >> 8;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf;
  for (int p = wave; p < npieces; p += NW)
    glds16_asm_s(g + (size_t)p * 256, (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)p * 1024u));
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

// LayerNorm over the 256 features of each column (eps 1e-5, biased variance), residual, store of this wave's quarter, zero fill
// of the next aggregate: the quarter sums of a column meet in LDS and are added in the order r = 0 .. 3 by every wave
template <int CG>
__device__ __forceinline__ void rs_epilogue(const ChainArgs& a, const RsWave& w, f32x4 (&o)[4], const f32x4 (&rres)[4], float* stat) {
  const int r = w.r, q = w.q, j = w.j;
  if (a.gamma != nullptr) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += (o[t].x + o[t].y) + (o[t].z + o[t].w);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    float* st0 = stat + (w.g * 4) * 16;
    float* st1 = stat + (CG * 4 + w.g * 4) * 16;
    if (q == 0) st0[r * 16 + j] = s;
    __syncthreads();
    const float mean = ((st0[j] + st0[16 + j]) + (st0[32 + j] + st0[48 + j])) * (1.0f / 256.0f);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = o[t][i] - mean;
        v += d * d;
      }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (q == 0) st1[r * 16 + j] = v;
    __syncthreads();
    const float var = ((st1[j] + st1[16 + j]) + (st1[32 + j] + st1[48 + j])) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
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
}

// ================================================ fp32 (arithmetic of chain_kernel) ================================================

// in[8c .. 8c+7] <- row[k(s, q)] for the K-steps of chunk c (k(s, q) = 16 (s >> 2) + 4 q + (s & 3))
__device__ __forceinline__ void load_slice(float (&in)[64], const float* __restrict__ row, int c, int q) {
#pragma unroll
  for (int i = 2 * c; i < 2 * c + 2; ++i) {
    const f32x4 v = ldg4(row + 16 * i + 4 * q);
    in[4 * i + 0] = v.x;
    in[4 * i + 1] = v.y;
    in[4 * i + 2] = v.z;
    in[4 * i + 3] = v.w;
  }
}

// One 256-deep pass of this wave's four row tiles: acc[t] += W[16 (4r + t) .., k] . in[k].  Protocol of chain_kernel's mma_pass:
// the first chunk of the pass is already on its way into buffer `parity`; the first chunk of the next pass is issued while the
// last one computes.  RELOAD: the 8 operand registers a chunk has consumed are refilled, one chunk later, with the same K slice
// of the next layer-1 operand (`next_row`); the slice the previous pass consumed last is refilled during chunk 0 (`tail_row`).
template <int NW, bool RELOAD>
__device__ __forceinline__ void rs_pass(f32x4 (&acc)[4], float (&in)[64], const float* __restrict__ gw, const float* __restrict__ next_gw,
                                        float* lds, int& parity, int lane, int wave, int r, const float* __restrict__ tail_row, bool do_tail,
                                        const float* __restrict__ next_row, bool do_next, int q) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunk c has landed for every wave; nobody still reads the other buffer (weights or exchanged rows)
    float* other = lds + (parity ^ 1) * kLdsBufFloats;
    if (c + 1 < 8)
      issue_chunk_nw<NW>(gw + (size_t)(c + 1) * kChunkSteps * kStepFloats, kChunkSteps * kStepFloats, other, lane, wave);
    else if (next_gw != nullptr)
      issue_chunk_nw<NW>(next_gw, kChunkSteps * kStepFloats, other, lane, wave);
    if (RELOAD) {
      if (c == 0) {
        if (do_tail) load_slice(in, tail_row, 7, q);
      } else if (do_next) {
        load_slice(in, next_row, c - 1, q);
      }
    }
    const float* buf = lds + parity * kLdsBufFloats + r * 256 + lane * 4;
    f32x4 a_cur = *(const f32x4*)buf;
#pragma unroll
    for (int s = 0; s < kChunkSteps; ++s) {
      f32x4 a_nxt = a_cur;
      if (s + 1 < kChunkSteps) a_nxt = *(const f32x4*)(buf + (s + 1) * kStepFloats);
      const float b = in[c * kChunkSteps + s];
      __builtin_amdgcn_sched_barrier(0);  // the LDS read of step s + 1 stays ahead of the MFMAs of step s
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t], b, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
    }
    parity ^= 1;
  }
}

// The four waves of a column group publish their quarter of a layer's output (RELU: its relu) and every wave reads the
// whole 256-feature B operand of the next layer back: in[4 T + i] = value of row tile T, register i of this lane.
// OVERLAY: xg lies in the weight buffer the pass has just finished with - slower waves may still be reading its last chunk.
template <bool RELU, bool OVERLAY>
__device__ __forceinline__ void exchange(float (&in)[64], const f32x4 (&acc)[4], float* xg, int r, int lane) {
  if (OVERLAY) __syncthreads();
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
  constexpr bool OV = rs_overlay(CG);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xfix = lds + 2 * kLdsBufFloats;                       // the exchange buffer when it has its own LDS (CG = 3)
  float* stat = xfix + (OV ? 0 : CG * kXbufFloats);            // [2][CG][4][16]: partial LayerNorm sums of the row quarters
  const RsWave w = rs_wave<CG>(a);
  const int lane = w.lane, wave = w.wave, r = w.r, q = w.q;
  int parity = 0;
  // exchange buffer of this wave's column group: over the weight buffer the finished pass consumed last (OV), or its own
  auto xg = [&]() -> float* { return (OV ? lds + (parity ^ 1) * kLdsBufFloats : xfix) + w.g * kXbufFloats; };

  const bool raw0 = a.seg_k[0] > 0 && a.seg_proj[0] == 0;  // x . Wx^T is a matrix pass
  const bool prj0 = a.seg_k[0] > 0 && a.seg_proj[0] != 0;  // ... or a cached product row (gather-add); neither: x == 0
  issue_chunk_nw<NW>(raw0 ? a.w1[0] : a.w1[1], kChunkSteps * kStepFloats, lds, lane, wave);

  const float* arow = operand_row(a.seg_ptr[1], a.seg_idx[1], a.seg_rows_pb[1], a.seg_ld[1], w.b, w.k);
  float in[64];
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b1 + 16 * (4 * r + t) + 4 * q);

  // ---- layer 1: cat[x, agg] . W1^T = x . Wx^T + agg . Wa^T ----
  if (raw0) {  // the aggregate rows stream in under the pass of x
    const float* xrow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) load_slice(in, xrow, cc, q);
    rs_pass<NW, true>(acc, in, a.w1[0], a.w1[1], lds, parity, lane, wave, r, nullptr, false, arow, true, q);
    rs_pass<NW, true>(acc, in, a.w1[1], a.w_mid, lds, parity, lane, wave, r, arow, true, nullptr, false, q);
  } else {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) load_slice(in, arow, cc, q);
    if (prj0) {
      const float* prow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] += ldg4(prow + 16 * (4 * r + t) + 4 * q);
    }
    rs_pass<NW, false>(acc, in, a.w1[1], a.w_mid, lds, parity, lane, wave, r, nullptr, false, nullptr, false, q);
  }

  // ---- middle layer ----
  exchange<true, OV>(in, acc, xg(), r, lane);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b_mid + 16 * (4 * r + t) + 4 * q);
  rs_pass<NW, false>(acc, in, a.w_mid, a.w_out, lds, parity, lane, wave, r, nullptr, false, nullptr, false, q);

  // ---- output layer (the residual rows are requested underneath it) ----
  exchange<true, OV>(in, acc, xg(), r, lane);
  f32x4 o[4], rres[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = ldg4(a.b_out + 16 * (4 * r + t) + 4 * q);
  if (a.res_ptr != nullptr) {
    const float* rrow = operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, w.b, w.k);
#pragma unroll
    for (int t = 0; t < 4; ++t) rres[t] = ldg4(rrow + 16 * (4 * r + t) + 4 * q);
  }
  rs_pass<NW, false>(o, in, a.w_out, a.n_post > 0 ? a.proj_w[0] : nullptr, lds, parity, lane, wave, r, nullptr, false, nullptr, false, q);

  rs_epilogue<CG>(a, w, o, rres, stat);

  // ---- POST: the next block's layer-1 products of the new rows (ChainArgs) ----
  if (a.n_post > 0) {
    exchange<false, OV>(in, o, xg(), r, lane);
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
}

// ============================== bf16x3: split operands on v_mfma_f32_16x16x32_bf16 (arithmetic of gw_split.hip) ==============================
// Packed stream (gw_pack_linear_bf16x3): per 32-wide K-step the hi fragments of the 16 row tiles, then their lo fragments,
// 1 KiB each = 32 KiB = one chunk.  K order k(s, q, i) = 32 s + 16 (i >> 2) + 4 q + (i & 3): K-step s of a layer's B operand is
// split(acc[2s], acc[2s + 1]) of the layer before, so row quarter r owns K-steps 2r, 2r + 1 of the next operand.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kStepBytes3 = 32768;

template <int NW>
__device__ __forceinline__ void issue_bytes_nw(const char* __restrict__ g, char* ldsbuf, int lane, int wave) {
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsbuf;
  for (int p = wave; p < 32; p += NW)
    glds16_asm_s((const float*)(g + (size_t)p * 1024), (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)p * 1024u));
}

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

// One 256-deep pass: 8 chunks of one K-step; this wave reads the hi / lo fragments of its four row tiles (8 x 1 KiB) and issues
// 12 MFMAs per chunk, term-major (hi.hi, hi.lo, lo.hi: two other tiles' MFMAs between two on the same accumulator).
// RELOAD as in rs_pass: K-step c - 1 of the next raw operand replaces the registers chunk c - 1 consumed.
template <int NW, bool RELOAD>
__device__ __forceinline__ void rs3_pass(f32x4 (&acc)[4], bf16x8 (&bh)[8], bf16x8 (&bl)[8], const char* __restrict__ gw,
                                         const char* __restrict__ next_gw, char* lds, int& parity, int lane, int wave, int r,
                                         const float* __restrict__ tail_row, bool do_tail, const float* __restrict__ next_row, bool do_next,
                                         int q) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* other = lds + (parity ^ 1) * kStepBytes3;
    if (c + 1 < 8)
      issue_bytes_nw<NW>(gw + (size_t)(c + 1) * kStepBytes3, other, lane, wave);
    else if (next_gw != nullptr)
      issue_bytes_nw<NW>(next_gw, other, lane, wave);
    f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = t0;
    const float* rrow = nullptr;
    int slot = 0;
    if (RELOAD) {
      if (c == 0) {
        if (do_tail) { rrow = tail_row; slot = 7; }
      } else if (do_next) {
        rrow = next_row;
        slot = c - 1;
      }
      if (rrow != nullptr) {
        t0 = ldg4(rrow + 32 * slot + 4 * q);
        t1 = ldg4(rrow + 32 * slot + 16 + 4 * q);
      }
    }
    const char* buf = lds + parity * kStepBytes3 + (4 * r) * 1024 + lane * 16;
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ah[t] = *(const bf16x8*)(buf + t * 1024);
#pragma unroll
    for (int t = 0; t < 4; ++t) al[t] = *(const bf16x8*)(buf + 16384 + t * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[c], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl[c], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh[c], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (RELOAD) {
      if (rrow != nullptr) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s == slot) split8(t0, t1, bh[s], bl[s]);
      }
    }
    parity ^= 1;
  }
}

// exchange of the split activations: row quarter r publishes K-steps 2r, 2r + 1 (hi, lo), every wave reads all eight back
template <bool RELU, bool OVERLAY>
__device__ __forceinline__ void exchange3(bf16x8 (&bh)[8], bf16x8 (&bl)[8], const f32x4 (&acc)[4], char* xg, int r, int lane) {
  if (OVERLAY) __syncthreads();
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
  constexpr bool OV = rs_overlay(CG);
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  char* xfix = lds3 + 2 * kStepBytes3;
  float* stat = (float*)(xfix + (OV ? 0 : CG * kXbufFloats * 4));
  const RsWave w = rs_wave<CG>(a);
  const int lane = w.lane, wave = w.wave, r = w.r, q = w.q;
  int parity = 0;
  auto xg = [&]() -> char* { return (OV ? lds3 + (parity ^ 1) * kStepBytes3 : xfix) + w.g * (kXbufFloats * 4); };

  const bool raw0 = a.seg_k[0] > 0 && a.seg_proj[0] == 0;
  const bool prj0 = a.seg_k[0] > 0 && a.seg_proj[0] != 0;
  const char* w1x = (const char*)a.w1[0];
  const char* w1a = (const char*)a.w1[1];
  const char* w_mid = (const char*)a.w_mid;
  const char* w_out = (const char*)a.w_out;
  issue_bytes_nw<NW>(raw0 ? w1x : w1a, lds3, lane, wave);

  const float* arow = operand_row(a.seg_ptr[1], a.seg_idx[1], a.seg_rows_pb[1], a.seg_ld[1], w.b, w.k);
  bf16x8 bh[8], bl[8];
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b1 + 16 * (4 * r + t) + 4 * q);

  if (raw0) {
    const float* xrow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
    for (int s = 0; s < 8; ++s) load_slice3(bh[s], bl[s], xrow, s, q);
    rs3_pass<NW, true>(acc, bh, bl, w1x, w1a, lds3, parity, lane, wave, r, nullptr, false, arow, true, q);
    rs3_pass<NW, true>(acc, bh, bl, w1a, w_mid, lds3, parity, lane, wave, r, arow, true, nullptr, false, q);
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) load_slice3(bh[s], bl[s], arow, s, q);
    if (prj0) {
      const float* prow = operand_row(a.seg_ptr[0], a.seg_idx[0], a.seg_rows_pb[0], a.seg_ld[0], w.b, w.k);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] += ldg4(prow + 16 * (4 * r + t) + 4 * q);
    }
    rs3_pass<NW, false>(acc, bh, bl, w1a, w_mid, lds3, parity, lane, wave, r, nullptr, false, nullptr, false, q);
  }

  exchange3<true, OV>(bh, bl, acc, xg(), r, lane);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = ldg4(a.b_mid + 16 * (4 * r + t) + 4 * q);
  rs3_pass<NW, false>(acc, bh, bl, w_mid, w_out, lds3, parity, lane, wave, r, nullptr, false, nullptr, false, q);

  exchange3<true, OV>(bh, bl, acc, xg(), r, lane);
  f32x4 o[4], rres[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = ldg4(a.b_out + 16 * (4 * r + t) + 4 * q);
  if (a.res_ptr != nullptr) {
    const float* rrow = operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, w.b, w.k);
#pragma unroll
    for (int t = 0; t < 4; ++t) rres[t] = ldg4(rrow + 16 * (4 * r + t) + 4 * q);
  }
  rs3_pass<NW, false>(o, bh, bl, w_out, a.n_post > 0 ? (const char*)a.proj_w[0] : nullptr, lds3, parity, lane, wave, r, nullptr, false,
                      nullptr, false, q);

  rs_epilogue<CG>(a, w, o, rres, stat);

  if (a.n_post > 0) {
    exchange3<false, OV>(bh, bl, o, xg(), r, lane);
#pragma unroll 1
    for (int sl = 0; sl < a.n_post; ++sl) {
      f32x4 pacc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) pacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* nx = sl + 1 < a.n_post ? (const char*)a.proj_w[sl + 1] : nullptr;
      rs3_pass<NW, false>(pacc, bh, bl, (const char*)a.proj_w[sl], nx, lds3, parity, lane, wave, r, nullptr, false, nullptr, false, q);
      if (w.valid) {
        float* prow = a.proj_out[sl] + (size_t)w.c * 256;
#pragma unroll
        for (int t = 0; t < 4; ++t) stg4(prow + 16 * (4 * r + t) + 4 * q, pacc[t]);
      }
    }
  }
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
