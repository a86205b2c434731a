// gw_split.hip - split-operand ("bf16x3") variant of the fused MLP kernels: GW_DTYPE_BF16X3.
//
// The reference computes every Linear in fp32 (graph_net_block.py:45-61); BASELINE.json asks for its outputs within 1e-3 and for
// matrix products on the bf16 matrix cores.  One bf16 operand carries 8 significant bits (measured 9.5e-3 end to end), so this
// mode keeps BOTH operands of every product as a pair of bf16 values
//       x = x_hi + x_lo,   x_hi = bf16(x) (RNE),  x_lo = bf16(x - x_hi)          (16 significant bits, fp32 exponent range)
// and evaluates  x . w  ~=  x_hi . w_hi + x_hi . w_lo + x_lo . w_hi   on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: three
// MFMAs per product (the dropped x_lo . w_lo term is 2^-16 relative to a product).  Three bf16 MFMAs cost 48 cycles per
// 16x16x32 block against 256 cycles for the eight v_mfma_f32_16x16x4_f32 of the exact-fp32 kernels: 5.3 x the matrix rate at
// ~1e-5 per product.  Everything that is not a matrix product (bias, LayerNorm statistics, residual adds, segment sums, every
// tensor in HBM) stays fp32, exactly as in the fp32 kernels.
//
// Structure: the transposed, register-resident scheme of gw_bf16.hip (weights = A operand streamed through LDS by
// global_load_lds DMA, activations = B operand, K order k(s, q, i) = 32 s + 16 (i >> 2) + 4 q + (i & 3), so a layer's accumulator
// is the next layer's B operand).  The packed stream carries, per 32-wide K-step, the hi fragments of all row tiles followed by
// their lo fragments (gw_pack_linear_bf16x3); activations are split in registers when a layer's output becomes the next layer's
// input.  A wave reads two fragments (hi, lo) from LDS per three MFMAs; the 4-byte-per-weight stream is the fp32 kernels' volume.
//
// Measured (1 degree, profiles/r05_*): 5.3 x the matrix rate buys 2.0 x the fp32 forward (555 vs 278 forecasts/s at batch 2): a
// 256 x 256 layer pass of a 64-column workgroup is 6.4 k cycles of MFMAs per wave but takes ~21 k - the same with the weight
// chunks moved by LDS-DMA (all pieces up front / spread between the MFMA units / over the first half of a chunk) or through
// registers (global_load_dwordx4 + ds_write_b128), with 64 or 128 columns per weight stream, 32 or 64 KiB chunks, a 3- or 5-deep
// fragment ring.  What does not move it either way is what a pass is made of: 8 chunk hand-overs (DMA landing + barrier: ~25 % of
// the pass), 256 KiB of fragment reads per wave (LDS array 25-40 % busy) and an MFMA pipe shared by two waves (33-43 % busy over the
// kernel, SQ_VALU_MFMA_BUSY_CYCLES).  Nor does the order of the instructions inside a unit or the register file of the accumulators.
// Nor is it the weight stream out of L2 (9.0 GB of L2 requests per decoder launch = 10.8 TB/s, a third of what the same
// double-buffered LDS-DMA stream reaches with nothing else going on: profiles/r05_pmc_l2_c2x3.json, r05_l2_stream_probe.log): every
// resource of a pass is about half busy and the pass is the serial skeleton of its 8 one-K-step chunks (DMA landing, barrier,
// restart of the fragment ring behind the barrier, 48 MFMAs, 8 DMA issues per wave).
// DESIGN.md "bf16x3" lists the experiments.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// phase clocks (tuning builds only: gw_debug_timestamps + scripts/gpu_timeline_x3.py); nothing in the product build
#ifdef GW_TUNING
#define X3_STAMP(i) \
  if (a.dbg != nullptr) { ts[i] = gw::gw_clock(); }
#else
#define X3_STAMP(i)
#endif

// One weight chunk buffer (double buffered): 32 KiB = one K-step of 16 row tiles (hi + lo) for the 4-wave workgroups (two per
// CU), 64 KiB = two K-steps for the 8-wave workgroup (one per CU: half the chunk hand-overs per layer)
constexpr int buf_bytes(int nw) { return nw == 8 ? 65536 : 32768; }
constexpr int kStageLd = 260;
constexpr int kStageFloats = 64 * kStageLd;
// LDS of a launch: the two weight buffers; the segment-sum stage of the edge epilogue (64 columns x 260 floats + 64 destination
// ids per 4 waves) lies OVER them (nothing streams any more by then), so two 64-column workgroups per CU fit the 160 KiB
constexpr int lds_bytes(int nw, bool edge) {
  const int w = 2 * buf_bytes(nw), st = (nw / 4) * (kStageFloats + 64) * 4;
  return edge && st > w ? st : w;
}
constexpr int kLdsMax = 160 * 1024;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// DMA `bytes` (multiple of 1 KiB) of the packed weight stream into LDS at byte offset lds_off; pieces round-robin over the waves.
template <int NW>
__device__ __forceinline__ void issue_bytes(const char* __restrict__ g, int bytes, unsigned lds_off, int lane, int wave) {
  const int npieces = bytes >> 10;
  for (int p = wave; p < npieces; p += NW)
    glds16_asm_s((const float*)(g + (size_t)p * 1024), (unsigned)lane * 16u,
                 __builtin_amdgcn_readfirstlane(lds_off + (unsigned)p * 1024u));
}

// x -> (bf16(x), bf16(x - bf16(x))) for the 8 values of one B fragment
__device__ __forceinline__ void split8(f32x4 a, f32x4 b, bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const __bf16 ha = (__bf16)a[r];
    h[r] = ha;
    l[r] = (__bf16)(a[r] - (float)ha);
    const __bf16 hb = (__bf16)b[r];
    h[4 + r] = hb;
    l[4 + r] = (__bf16)(b[r] - (float)hb);
  }
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

// One layer pass: acc[g][t] += W[16t.., k] . x[g][k] on split operands, K = 32 KS, NT row tiles, NTP = tiles per K-step in the
// packed stream (NT rounded up to 4).  The stream of this pass starts at gw; its first chunk has already been issued into buffer
// `parity`; while the last chunk computes, the first chunk of the next pass (next_gw, next_bytes) is issued.
// The DMA pieces of chunk c + 1 are issued one at a time between the MFMA units of chunk c.  (Measured alternatives, same time
// per pass within 10 %: all pieces up front; chunks through registers - global_load_dwordx4 + ds_write_b128 - instead of LDS-DMA:
// DESIGN.md "bf16x3".)  RING = register sets the A fragments travel through: a unit's fragments are requested RING - 1 units
// ahead of its MFMAs.
template <int NW, int NG, int KS, int BKS, int NT, int NTP, int RING>
__device__ __forceinline__ void pass_x3(f32x4 (&acc)[NG][NT], const bf16x8 (&bh)[NG][BKS], const bf16x8 (&bl)[NG][BKS],
                                        const char* __restrict__ gw, const char* __restrict__ next_gw, int next_bytes, const char* lds,
                                        int& parity, int lane, int wave, unsigned long long* waited = nullptr) {
  constexpr int kBufBytes = buf_bytes(NW);
  constexpr int STEP_BYTES = 2 * NTP * 1024;  // hi fragments of the NTP tiles, then their lo fragments
  constexpr int CS = kBufBytes / STEP_BYTES;  // K-steps per chunk
  constexpr int NCH = (KS + CS - 1) / CS;
  constexpr int UPS = NTP / 2;  // units per K-step: a unit = 2 row tiles = 4 ds_read_b128 (hi, hi, lo, lo) -> 6 NG MFMAs
  constexpr int PPW = (CS * STEP_BYTES / 1024 + NW - 1) / NW;  // DMA pieces per wave and full chunk
  constexpr int UNITS = CS * UPS;
  constexpr int AHEAD = RING - 1;
  static_assert(PPW <= UNITS, "at most two DMA pieces per unit");
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int steps_c = (KS - c * CS) < CS ? (KS - c * CS) : CS;
#ifdef GW_TUNING
    unsigned long long t0w = 0, t1w = 0;
    if (waited != nullptr) t0w = gw::gw_clock();
#endif
    wait_vm<0>();
#ifdef GW_TUNING
    if (waited != nullptr) t1w = gw::gw_clock();
#endif
    lds_barrier();  // chunk c has landed for every wave; nobody still reads the other buffer
#ifdef GW_TUNING
    if (waited != nullptr) {
      const unsigned long long t2w = gw::gw_clock();
      waited[0] += t1w - t0w;  // own DMA pieces landing
      waited[1] += t2w - t1w;  // LDS drain + barrier
    }
#endif
    // what goes into the other buffer while this chunk computes
    const char* nsrc = nullptr;
    int npieces = 0;
    if (c + 1 < NCH) {
      const int sn = (KS - (c + 1) * CS) < CS ? (KS - (c + 1) * CS) : CS;
      nsrc = gw + (size_t)(c + 1) * CS * STEP_BYTES;
      npieces = sn * STEP_BYTES / 1024;
    } else if (next_gw != nullptr) {
      nsrc = next_gw;
      npieces = next_bytes >> 10;
    }
    const unsigned nlds = (unsigned)((parity ^ 1) * kBufBytes);
    const char* buf = lds + parity * kBufBytes + lane * 16;
    const int NU = steps_c * UPS;
    bf16x8 af[RING][4];
    auto ldu = [&](bf16x8 (&f)[4], int u) {
      const int su = u / UPS, t2 = u - su * UPS;
      const char* p = buf + su * STEP_BYTES + (2 * t2) * 1024;
      f[0] = *(const bf16x8*)(p);
      f[1] = *(const bf16x8*)(p + 1024);
      f[2] = *(const bf16x8*)(p + NTP * 1024);
      f[3] = *(const bf16x8*)(p + NTP * 1024 + 1024);
    };
#pragma unroll
    for (int u = 0; u < AHEAD; ++u)
      if (u < NU) ldu(af[u], u);
    __builtin_amdgcn_sched_barrier(0);
    int issued = 0;  // pieces of this wave issued so far (compile-time after unrolling)
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      if (u < NU) {
        if (u + AHEAD < NU) ldu(af[(u + AHEAD) % RING], u + AHEAD);
        const int su = u / UPS, t2 = u - su * UPS;
        const int ks = c * CS + su;
        // term-major: two MFMAs (x NG) lie between two that accumulate into the same registers
#pragma unroll
        for (int term = 0; term < 3; ++term) {
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            if (2 * t2 + tt < NT) {
#pragma unroll
              for (int g = 0; g < NG; ++g) {
                const bf16x8 wa = af[u % RING][term == 2 ? 2 + tt : tt];
                const bf16x8 xb = term == 1 ? bl[g][ks] : bh[g][ks];
                acc[g][2 * t2 + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[g][2 * t2 + tt], 0, 0, 0);
              }
            }
          }
        }
      }
      // this unit's share of the next chunk's DMA (pieces wave, wave + NW, ...): spread evenly over the units that exist
      {
        // (over the FIRST HALF of the chunk's units: the last piece then has half a chunk of MFMAs to land - issued over all units,
        // a wave waited ~1 k cycles per chunk for its own last pieces: profiles/r05_x3_timeline.log)
        const int span = (NU < UNITS ? NU : UNITS) / 2 > 0 ? (NU < UNITS ? NU : UNITS) / 2 : 1;
        const int due = ((u + 1) * PPW + span - 1) / span;  // pieces due after unit u
        const int due_c = u + 1 >= span ? PPW : (due < PPW ? due : PPW);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
          if (i >= issued && i < due_c) {
            const int pc = wave + i * NW;
            // inside a pass the piece count is a compile-time multiple of NW (no branch); only the hand-over to the next pass
            // (which may not exist) tests a wave-uniform condition
            const bool go = (c + 1 < NCH) ? (i * NW < npieces) : (pc < npieces);
            if (go)
              glds16_asm_s((const float*)(nsrc + (size_t)pc * 1024), (unsigned)lane * 16u,
                           __builtin_amdgcn_readfirstlane(nlds + (unsigned)pc * 1024u));
          }
        }
        issued = due_c > issued ? due_c : issued;
      }
      if (u < NU) __builtin_amdgcn_sched_barrier(0);
    }
    parity ^= 1;
  }
}

template <int NG, int NT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[NG][NT], const float* __restrict__ bias, int q) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 bv = bias ? ldg4(bias + 16 * t + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g][t] = bv;
  }
}

// (bh, bl)[s] <- split(row[k(s,q,i)]) for a raw operand row (valid features [0, kvalid))
template <int KS, bool FULL, int DEPTH>
__device__ __forceinline__ void load_raw(bf16x8 (&bh)[KS], bf16x8 (&bl)[KS], const float* __restrict__ row, int kvalid, int q) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool pairs = !FULL && (kvalid & 1) == 0 && ((size_t)row & 7) == 0;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f32x4 lo, hi;
    if (FULL) {
      lo = ldg4(row + 32 * s + 4 * q);
      hi = ldg4(row + 32 * s + 16 + 4 * q);
    } else if (pairs) {  // rows of an even number of floats at an 8-byte aligned base (102 input features): 8-byte loads
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
    split8(lo, hi, bh[s], bl[s]);
    if ((s & (DEPTH - 1)) == DEPTH - 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// the next layer's B operand from this layer's accumulators (row tiles 2s, 2s+1 = K-step s)
template <int NT, bool RELU>
__device__ __forceinline__ void acc_to_b(bf16x8 (&bh)[NT / 2], bf16x8 (&bl)[NT / 2], const f32x4 (&acc)[NT]) {
#pragma unroll
  for (int s = 0; s < NT / 2; ++s) {
    if (RELU)
      split8(relu4(acc[2 * s]), relu4(acc[2 * s + 1]), bh[s], bl[s]);
    else
      split8(acc[2 * s], acc[2 * s + 1], bh[s], bl[s]);
  }
}

__device__ __forceinline__ const float* operand_row(const float* ptr, const int* idx, int rows_pb, int ld, int b, int k) {
  const int r = idx ? ldgi(idx + k) : k;
  return ptr + ((size_t)b * (size_t)rows_pb + (size_t)r) * (size_t)ld;
}

// K1S: 32-wide K-steps of a raw layer-1 operand (8: k = 256, 4: k <= 128, 1: k <= 32); HT / OT: hidden / output row tiles.
template <int K1S, bool K1FULL, int NSEG, int HT, int OT, int EPI, bool SINGLE, bool POST, bool HEAD, int NW, int NG, int RING = 3>
__global__ __launch_bounds__(NW * 64, (NW * NG == 4 ? 2 : 1)) void chainx3_kernel(const ChainArgs a) {
  constexpr int kCols = NW * NG * 16;  // columns per workgroup
  static_assert(EPI != EPI_EDGE || NW == 4 || NW == 8, "the segment-sum epilogue walks 64 columns per round and 4 waves");
  extern __shared__ __attribute__((aligned(16))) char ldsx[];
  constexpr int HTP = (HT + 3) / 4 * 4, OTP = (OT + 3) / 4 * 4;
  constexpr int HKS = HT / 2;  // K-steps of a layer fed by the hidden activations
  constexpr int kBufBytes = buf_bytes(NW);
  constexpr int H_STEP = 2 * HTP * 1024, O_STEP = 2 * OTP * 1024;
  constexpr int H_CS = kBufBytes / H_STEP, O_CS = kBufBytes / O_STEP;  // K-steps per chunk
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  const int tile_c0 = blockIdx.x * kCols;
#ifdef GW_TUNING
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long waited[2] = {0, 0};
#define X3_WAITED (a.dbg != nullptr ? waited : nullptr)
#else
#define X3_WAITED nullptr
#endif
  X3_STAMP(0)

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

  // training (gw_activation_save): the relu output of hidden layer l of every column, as the fp32 kernels store it - the
  // backward's weight-gradient GEMMs and ReLU masks read these rows (fp32; only the products are split)
  auto save_hidden = [&](int l, const f32x4 (&h)[NG][HT]) {
    if (a.save_h == nullptr) return;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (valid[g]) {
        float* srow = a.save_h + (size_t)l * (size_t)a.save_stride + (size_t)cc[g] * (size_t)a.save_ld;
#pragma unroll
        for (int t = 0; t < HT; ++t) stg4(srow + 16 * t + 4 * q, relu4(h[g][t]));
      }
  };

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
  constexpr int MID_FIRST = (HKS < H_CS ? HKS : H_CS) * H_STEP, OUT_FIRST = (HKS < O_CS ? HKS : O_CS) * O_STEP;  // first chunks
  const int after_l1_bytes = SINGLE ? 0 : (a.n_mid > 0 ? MID_FIRST : OUT_FIRST);
  int parity = 0;
  {
    const char* first = on[0] ? w1[0] : (on[1] ? w1[1] : (on[2] ? w1[2] : after_l1));
    const int first_bytes = (on[0] || on[1] || on[2]) ? K1FIRST : after_l1_bytes;
    issue_bytes<NW>(first, first_bytes, 0u, lane, wave);
  }
  // (first-chunk sizes are whole staging groups of pass_x3: half a chunk buffer)
  static_assert(K1FIRST % (kBufBytes / 2) == 0 && MID_FIRST % (kBufBytes / 2) == 0 && OUT_FIRST % (kBufBytes / 2) == 0, "first chunks");

  // ---- layer 1 ----
  constexpr int BKS = K1S > HKS ? K1S : HKS;
  constexpr bool SHARE_ACC = (OT == HT);
  f32x4 acc[NG][HT];
  bf16x8 bh[NG][BKS], bl[NG][BKS];
  init_bias<NG, HT>(acc, a.b1, q);
  {
    // Projected operands (rows already multiplied by their layer-1 slice: gather-adds).  The first two of them are fetched with
    // ALL their row pieces in flight at once (2 x 16 loads of 16 bytes per lane and group) - summed one operand after the other
    // in batches of 8 the gathers were four serial memory round trips, 14 k of a decoder tile's 67 k cycles.
    int ia = -1, ib = -1, ic = -1;
#pragma unroll
    for (int i = 0; i < NSEG; ++i)
      if (prj[i]) {
        if (ia < 0) ia = i;
        else if (ib < 0) ib = i;
        else ic = i;
      }
    auto prow = [&](int i, int g) -> const float* {
      const float* ptr = i == 0 ? a.seg_ptr[0] : (i == 1 ? a.seg_ptr[1] : a.seg_ptr[2]);
      const int* idx = i == 0 ? a.seg_idx[0] : (i == 1 ? a.seg_idx[1] : a.seg_idx[2]);
      const int rpb = i == 0 ? a.seg_rows_pb[0] : (i == 1 ? a.seg_rows_pb[1] : a.seg_rows_pb[2]);
      const int ld = i == 0 ? a.seg_ld[0] : (i == 1 ? a.seg_ld[1] : a.seg_ld[2]);
      return operand_row(ptr, idx, rpb, ld, bb[g], kk[g]);
    };
    if (ia >= 0) {
      f32x4 pa[NG][HT], pb[NG][HT];
      const bool two = ib >= 0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float* ra = prow(ia, g);
#pragma unroll
        for (int t = 0; t < HT; ++t) pa[g][t] = ldg4(ra + 16 * t + 4 * q);
      }
      if (two) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const float* rb = prow(ib, g);
#pragma unroll
          for (int t = 0; t < HT; ++t) pb[g][t] = ldg4(rb + 16 * t + 4 * q);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[g][t] += pa[g][t];
      if (two) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int t = 0; t < HT; ++t) acc[g][t] += pb[g][t];
      }
    }
    if (ic >= 0) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float* row = prow(ic, g);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          acc[g][t] += ldg4(row + 16 * t + 4 * q);
          if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // raw operands: one matrix pass each
#pragma unroll
    for (int i = 0; i < NSEG; ++i) {
      if (on[i]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const float* row = operand_row(a.seg_ptr[i], a.seg_idx[i], a.seg_rows_pb[i], a.seg_ld[i], bb[g], kk[g]);
          load_raw<K1S, K1FULL, (NG == 1 ? 8 : 4)>(reinterpret_cast<bf16x8(&)[K1S]>(bh[g]), reinterpret_cast<bf16x8(&)[K1S]>(bl[g]), row,
                                                  a.seg_k[i], q);
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
        pass_x3<NW, NG, K1S, BKS, HT, HTP, RING>(acc, bh, bl, w1[i], nx, nb, ldsx, parity, lane, wave);
      }
    }
  }

  X3_STAMP(1)
  constexpr int HS = 2 * 8 * 1024;  // bytes of one K-step of a packed head slice (<= 8 row tiles, hi + lo)
  constexpr int HD_CS = kBufBytes / HS;
  constexpr int POST_FIRST = (8 < H_CS ? 8 : H_CS) * H_STEP;                            // products: K = 256, HT row tiles
  constexpr int HD1_FIRST = (8 < HD_CS ? 8 : HD_CS) * HS, HD2_FIRST = (4 < HD_CS ? 4 : HD_CS) * HS;  // head: K = 256, then K = 128 twice
  static_assert(POST_FIRST % (kBufBytes / 2) == 0 && HD1_FIRST % (kBufBytes / 2) == 0 && HD2_FIRST % (kBufBytes / 2) == 0, "first chunks");
  f32x4 o_sep[SHARE_ACC ? 1 : NG][SHARE_ACC ? 1 : OT];
  f32x4 (&o)[NG][OT] = *reinterpret_cast<f32x4(*)[NG][OT]>(SHARE_ACC ? (void*)acc : (void*)o_sep);
  if constexpr (!SINGLE) {
    // ---- middle layers (hidden -> hidden) ----
#pragma unroll 1
    for (int l = 0; l < a.n_mid; ++l) {
      save_hidden(l, acc);
#pragma unroll
      for (int g = 0; g < NG; ++g)
        acc_to_b<HT, true>(reinterpret_cast<bf16x8(&)[HKS]>(bh[g]), reinterpret_cast<bf16x8(&)[HKS]>(bl[g]), acc[g]);
      __builtin_amdgcn_sched_barrier(0);
      init_bias<NG, HT>(acc, a.b_mid + l * (HT * 16), q);
      const bool last = (l + 1 == a.n_mid);
      const char* nx = last ? w_out : w_mid + (size_t)(l + 1) * HKS * H_STEP;
      const int nb = last ? OUT_FIRST : MID_FIRST;
      pass_x3<NW, NG, HKS, BKS, HT, HTP, RING>(acc, bh, bl, w_mid + (size_t)l * HKS * H_STEP, nx, nb, ldsx, parity, lane, wave, X3_WAITED);
    }
    X3_STAMP(2)
    // ---- output layer ----
    save_hidden(a.n_mid, acc);
#pragma unroll
    for (int g = 0; g < NG; ++g)
      acc_to_b<HT, true>(reinterpret_cast<bf16x8(&)[HKS]>(bh[g]), reinterpret_cast<bf16x8(&)[HKS]>(bl[g]), acc[g]);
    __builtin_amdgcn_sched_barrier(0);
    init_bias<NG, OT>(o, a.b_out, q);
    pass_x3<NW, NG, HKS, BKS, OT, OTP, RING>(o, bh, bl, w_out, POST ? (const char*)a.proj_w[0] : (HEAD ? (const char*)a.hd_w1 : nullptr),
                                       POST ? POST_FIRST : (HEAD ? HD1_FIRST : 0), ldsx, parity, lane, wave, X3_WAITED);
  }

  X3_STAMP(3)
  // ---- LayerNorm over the OT*16 features of each column (eps 1e-5, biased variance), fp32 ----
  if (!SINGLE && a.gamma != nullptr) {
    if (a.save_y != nullptr) {  // training: the pre-LayerNorm rows
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (valid[g]) {
          float* srow = a.save_y + (size_t)cc[g] * (size_t)(OT * 16);
#pragma unroll
          for (int t = 0; t < OT; ++t) stg4(srow + 16 * t + 4 * q, o[g][t]);
        }
    }
    // heads and zero-padded narrow models normalise over their ln_width <= OT*16 real features (gw_mlp_weights.ln_width): the
    // padding rows of the last Linear are zero, so they drop out of the sum and are masked out of the variance
    const int nfeat = a.ln_width;
    const float inv_n = 1.0f / (float)nfeat;
    const bool full = nfeat == OT * 16;
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
          v += (full || 16 * t + 4 * q + r < nfeat) ? d * d : 0.f;
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
  if (!SINGLE && !HEAD && a.res_ptr != nullptr) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float* rrow = operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, bb[g], kk[g]);
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
  if (!HEAD && outp != nullptr) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (valid[g]) {
        float* orow = outp + (size_t)cc[g] * (size_t)a.out_ld;
        if (SINGLE && a.relu_mask != nullptr) {  // ReLU backward fused into an input-gradient product: rows [n_cols, 256]
          const float* mrow = a.relu_mask + (size_t)cc[g] * 256;
#pragma unroll
          for (int t = 0; t < OT; ++t) {
            const f32x4 mv = ldg4(mrow + 16 * t + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!(mv[r] > 0.f)) o[g][t][r] = 0.f;
          }
        }
#pragma unroll
        for (int t = 0; t < OT; ++t) {
          const int f0 = 16 * t + 4 * q;
          if (EPI == EPI_DEC) {
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

  X3_STAMP(4)
  // ---- HEAD: the output head on the new rows while they are still in registers (ChainArgs): 256 -> 128 relu -> 128 relu ->
  // <= 80 features (+ residual rows): AssimilatorDecoder.node_decoder + the Decoder residual (assimilator_decoder.py:197,
  // decoder.py:93) behind the decoder's node update; the [rows, 256] table between them is never written or read ----
  if constexpr (HEAD) {
    static_assert(!HEAD || (OT == 16 && HT == 16 && SHARE_ACC && !POST && !SINGLE), "HEAD follows a 256-wide node update");
#pragma unroll
    for (int g = 0; g < NG; ++g) acc_to_b<16, false>(reinterpret_cast<bf16x8(&)[8]>(bh[g]), reinterpret_cast<bf16x8(&)[8]>(bl[g]), o[g]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 hh[NG][8];
    init_bias<NG, 8>(hh, a.hd_b1, q);
    pass_x3<NW, NG, 8, BKS, 8, 8, RING>(hh, bh, bl, (const char*)a.hd_w1, (const char*)a.hd_w2, HD2_FIRST, ldsx, parity, lane, wave);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc_to_b<8, true>(reinterpret_cast<bf16x8(&)[4]>(bh[g]), reinterpret_cast<bf16x8(&)[4]>(bl[g]), hh[g]);
    __builtin_amdgcn_sched_barrier(0);
    init_bias<NG, 8>(hh, a.hd_b2, q);
    pass_x3<NW, NG, 4, BKS, 8, 8, RING>(hh, bh, bl, (const char*)a.hd_w2, (const char*)a.hd_w3, HD2_FIRST, ldsx, parity, lane, wave);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc_to_b<8, true>(reinterpret_cast<bf16x8(&)[4]>(bh[g]), reinterpret_cast<bf16x8(&)[4]>(bl[g]), hh[g]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 y[NG][5];
    init_bias<NG, 5>(y, a.hd_b3, q);
    pass_x3<NW, NG, 4, BKS, 5, 8, RING>(y, bh, bl, (const char*)a.hd_w3, nullptr, 0, ldsx, parity, lane, wave);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (valid[g]) {
        const float* rrow = a.res_ptr ? operand_row(a.res_ptr, a.res_idx, a.res_rows_pb, a.res_ld, bb[g], kk[g]) : nullptr;
        float* orow = a.out + (size_t)cc[g] * (size_t)a.out_ld;
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
    for (int g = 0; g < NG; ++g) acc_to_b<16, false>(reinterpret_cast<bf16x8(&)[8]>(bh[g]), reinterpret_cast<bf16x8(&)[8]>(bl[g]), o[g]);
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
      init_bias<NG, HT>(acc, nullptr, q);  // (acc aliases o: the new rows have been stored and split into bh / bl)
      const char* nx = sl + 1 < a.n_post ? (const char*)a.proj_w[sl + 1] : nullptr;
      pass_x3<NW, NG, 8, BKS, HT, HTP, RING>(acc, bh, bl, (const char*)a.proj_w[sl], nx, POST_FIRST, ldsx, parity, lane, wave);
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (valid[g]) {
          float* prow = a.proj_out[sl] + (size_t)cc[g] * 256;
#pragma unroll
          for (int t = 0; t < HT; ++t) stg4(prow + 16 * t + 4 * q, acc[g][t]);
        }
    }
  }

  X3_STAMP(5)
  // ---- segment sum over destination-sorted columns, 64 columns at a time through LDS (see gw_edge.hip) ----
  if (EPI == EPI_EDGE) {
    // every 4 waves (256 threads = 256 features) own one 64-column chunk of the round: stage + destination ids per chunk
    const int half = threadIdx.x >> 8;  // chunk of this thread's wave within the round (0 for 4-wave workgroups)
    float* stage = (float*)ldsx + half * (kStageFloats + 64);  // over the weight buffers: every pass of this workgroup is complete
    int* gdl = (int*)(stage + kStageFloats);
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      __syncthreads();  // the last pass's fragment reads / the previous round's readers are done
      {
        float* srow = stage + ((wave & 3) * 16 + j) * kStageLd + 4 * q;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2)
          if (g2 == g) {
#pragma unroll
            for (int t = 0; t < OT; ++t) *(f32x4*)(srow + 16 * t) = o[g2][t];
            if (q == 0) gdl[(wave & 3) * 16 + j] = valid[g2] ? bb[g2] * a.agg_rows_pb + ldgi(a.agg_idx + kk[g2]) : -1;
          }
      }
      __syncthreads();
      const int f = threadIdx.x & 255;
      float vv[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) vv[i] = stage[i * kStageLd + f];
      const int gdv = gdl[lane];
      const int gdn = gdl[lane < 63 ? lane + 1 : lane];
      const unsigned long long ends = __ballot(lane == 63 || gdn != gdv);
      // deterministic mode: carry records instead of atomics (gw_internal.hpp; same scheme as gw_edge.hip)
      bool open_lo = true, open_hi = true;
      float* rec = nullptr;
      if (a.carry != nullptr) {
        const int chunk_c0 = tile_c0 + g * (NW * 16) + half * 64;
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
#ifdef GW_TUNING
  if (a.dbg != nullptr) {
    ts[6] = gw::gw_clock();
    if (threadIdx.x == 0 && (int)blockIdx.x < a.dbg_cap && blockIdx.y == 0) {
      unsigned long long* rec = a.dbg + (size_t)blockIdx.x * 16;
      for (int i = 0; i < 7; ++i) rec[i] = ts[i];
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      rec[8] = hw;
      rec[10] = blockIdx.x;
      rec[11] = waited[0];  // middle + output passes: cycles waiting for own DMA pieces
      rec[12] = waited[1];  // ... and in the LDS drain + barrier
    }
  }
#endif
}

// ---- weight packing: nn.Linear [n_out, k_total] slice -> split bf16 MFMA A-operand stream ---------------------------
// out[s][half][tile][lane][i]: half 0 = bf16(w), half 1 = bf16(w - bf16(w)) of
// W[16 tile + (lane & 15)][k_lo + 32 s + 16 (i >> 2) + 4 (lane >> 4) + (i & 3)], 0 outside
__global__ void pack_linear_x3_kernel(const float* __restrict__ w, long long sf, long long sk, int n_out, int kseg, int ntp, int nsteps,
                                      __bf16* __restrict__ out) {
  const size_t total = (size_t)nsteps * ntp * 512;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const int tile = (int)((e >> 9) % ntp);
    const int s = (int)((e >> 9) / ntp);
    const int f = 16 * tile + (lane & 15);
    const int kx = 32 * s + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3);
    const float v = (f < n_out && kx < kseg) ? w[(long long)f * sf + (long long)kx * sk] : 0.f;
    const __bf16 h = (__bf16)v;
    const size_t o = ((size_t)s * 2 * ntp + tile) * 512 + (size_t)lane * 8 + i;
    out[o] = h;
    out[o + (size_t)ntp * 512] = (__bf16)(v - (float)h);
  }
}

// ---- gw_mlp_chain_backward on split operands: d_{i+1} = (d_i . W_i) * (h_i > 0) for the Linear layers above layer 1, then the
// layer-1 input gradients d_n . W1[:, block] - every product three bf16 MFMAs on (hi, lo) pairs like the forward, the gradient
// rows register-resident between the products (the accumulator of one is the B operand of the next), each d_i stored once for the
// weight-gradient GEMMs and never read back here.  Single launches of chainx3_kernel<SINGLE> read d_i again for every product
// that consumes it: 2 - 4 of the ~10 table passes of an MLP's input-gradient chain.  64 rows per workgroup, two per CU.
constexpr int kBwdLnScratch = 4 * 512 * 4;  // column sums of the four waves (LN: d gamma | d beta; bias gradient), behind the two weight buffers

// LN (gw_mlp_ln_chain_backward): a.d is the gradient at the output of the MLP's LayerNorm.  The kernel reads it together with the
// saved pre-norm row, walks back through the norm in registers (the row is spread over the 4 q lanes of its column: row sums are
// two shuffles, as in the forward's LayerNorm), stores the gradient at the norm's input once (the last Linear's weight-gradient
// GEMM reads it) and feeds it to the first product without reading it back - ln_bwd_kernel + this chain were three + one passes
// over the table, now three.  d gamma / d beta: per column over the wave's 16 rows by DPP row sums, over the four waves through
// LDS, one set of atomics per workgroup.
template <int RING, bool LN>
__global__ __launch_bounds__(256, 2) void bwd_chainx3_kernel(const gw::BwdChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ldsx[];
  constexpr int NW = 4, HT = 16, HKS = 8;
  constexpr int FIRST = buf_bytes(NW);  // first chunk of a 256 -> 256 stream: one K-step of 16 row tiles (hi + lo)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  const long long c_raw = (long long)blockIdx.x * (NW * 16) + wave * 16 + j;
  const bool valid = c_raw < a.n_rows;
  const long long c = valid ? c_raw : a.n_rows - 1;
  const int n_prod = a.n_chain + a.n_fan;
  int parity = 0;
  issue_bytes<NW>((const char*)a.w[0], FIRST, 0u, lane, wave);
  bf16x8 bh[1][HKS], bl[1][HKS];
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
    float* red_all = (float*)(ldsx + 2 * buf_bytes(NW));  // [wave][d gamma | d beta][256]
    f32x4 g[HT];  // this lane's 64 columns of its row: 16 t + 4 q + r
    ln_backward_rows16(g, a.ln_y + (size_t)c * 256, drow, drow2, a.ln_gamma, valid, q, j, red_all + wave * 512);
    float* orow = a.ln_dy + (size_t)c * 256;
#pragma unroll
    for (int s = 0; s < HKS; ++s) {
      if (valid) {
        stg4(orow + 32 * s + 4 * q, g[2 * s]);
        stg4(orow + 32 * s + 16 + 4 * q, g[2 * s + 1]);
      }
      split8(g[2 * s], g[2 * s + 1], bh[0][s], bl[0][s]);
    }
    lds_barrier();  // the four waves' column sums are in LDS
    ln_backward_flush(red_all, a.ln_dgamma, a.ln_dbeta, threadIdx.x);
  } else {
    load_raw<HKS, true, 8>(bh[0], bl[0], a.d + (size_t)c * (size_t)a.d_ld, 256, q);
  }
#pragma unroll 1
  for (int p = 0; p < n_prod; ++p) {
    const bool chain = p < a.n_chain;
    f32x4 acc[1][HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* nx = p + 1 < n_prod ? (const char*)a.w[p + 1] : nullptr;
    pass_x3<NW, 1, HKS, HKS, HT, HT, RING>(acc, bh, bl, (const char*)a.w[p], nx, nx ? FIRST : 0, ldsx, parity, lane, wave);
    if (chain) {  // the ReLU output that fed this layer gates its input gradient
      const float* mrow = a.mask[p] + (size_t)c * 256;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 mv = ldg4(mrow + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(mv[r] > 0.f)) acc[0][t][r] = 0.f;
      }
    }
    if (!chain && a.add[p] != nullptr) {  // an input gradient that joins another one of the same tensor (an edge row is operand
      // and residual of its block): added here instead of by a pass of its own over both tables
      const float* arow = a.add[p] + (size_t)c * (size_t)a.add_ld;
#pragma unroll
      for (int t = 0; t < HT; ++t) acc[0][t] += ldg4(arow + 16 * t + 4 * q);
    } else if (!chain && ((a.add_d_mask >> p) & 1u)) {  // ... the launch's own input gradient (never materialised when gathered)
#pragma unroll
      for (int t = 0; t < HT; ++t) acc[0][t] += ldg4(drow + 16 * t + 4 * q);
      if (drow2 != nullptr) {
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[0][t] += ldg4(drow2 + 16 * t + 4 * q);
      }
    }
    if (valid) {
      float* orow = a.out[p] + (size_t)c * 256;
#pragma unroll
      for (int t = 0; t < HT; ++t) stg4(orow + 16 * t + 4 * q, acc[0][t]);
    }
    if (chain && p + 1 == a.n_chain && a.colsum != nullptr)  // (uniform) Linear_0's bias gradient: column sums of this gradient
      colsum_rows64(acc[0], valid, q, j, wave, threadIdx.x, (float*)(ldsx + 2 * buf_bytes(NW)), a.colsum, [] { lds_barrier(); });
    if (chain) {
      acc_to_b<HT, false>(bh[0], bl[0], acc[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <typename K>
int launchx3(K kernel, ChainArgs& a, void* stream, int grid_y, int lds, int threads, int cols) {
  static DeviceOnce once;  // per template instantiation and device
  if (once.first()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax);
  const int grid = (a.n_cols + cols - 1) / cols;
  hipLaunchKernelGGL(kernel, dim3(grid, grid_y), dim3(threads), lds, (hipStream_t)stream, a);
  return check_launch("chainx3_kernel launch");
}

// one (kind, form) -> instantiation; form = 10 NW + NG.  Product build: 41 = 4 waves x 1 column group (64 columns, two workgroups
// per CU).  Tuning builds also carry 42 = 4 x 2 (128 columns, one wave per SIMD), 81 = 8 x 1 (128 columns share one weight stream,
// one workgroup per CU, 64 KiB chunks) and a 5-deep fragment ring (GW_X3_FORM / GW_X3_FORM_EDGE / GW_X3_RING): measured within
// +-3 % of form 41 on every launch of the 1 degree forward (profiles/r05_x3_forms.log), so the product carries one form.
constexpr int kRing = 3;  // register sets of the A-fragment ring
template <int K1S, bool K1F, int NSEG, int HT, int OT, int EPI, bool SINGLE, bool POST, bool HEAD>
int launch_kind(ChainArgs& a, void* stream, int gy, int form) {
  constexpr bool E = EPI == EPI_EDGE;
#ifdef GW_TUNING
  if (form == 42) return launchx3(chainx3_kernel<K1S, K1F, NSEG, HT, OT, EPI, SINGLE, POST, HEAD, 4, 2>, a, stream, gy, lds_bytes(4, E), 256, 128);
  static const int ring = GW_TUNE("GW_X3_RING", kRing);
  if (ring == 5) {
    if (form == 81) return launchx3(chainx3_kernel<K1S, K1F, NSEG, HT, OT, EPI, SINGLE, POST, HEAD, 8, 1, 5>, a, stream, gy, lds_bytes(8, E), 512, 128);
    return launchx3(chainx3_kernel<K1S, K1F, NSEG, HT, OT, EPI, SINGLE, POST, HEAD, 4, 1, 5>, a, stream, gy, lds_bytes(4, E), 256, 64);
  }
  if (form == 81) return launchx3(chainx3_kernel<K1S, K1F, NSEG, HT, OT, EPI, SINGLE, POST, HEAD, 8, 1, kRing>, a, stream, gy, lds_bytes(8, E), 512, 128);
#endif
  (void)form;
  return launchx3(chainx3_kernel<K1S, K1F, NSEG, HT, OT, EPI, SINGLE, POST, HEAD, 4, 1, kRing>, a, stream, gy, lds_bytes(4, E), 256, 64);
}
#define GW_X3(K1S, K1F, NSEG, HT, OT, EPI, SINGLE, POST, HEAD, GY) \
  return launch_kind<K1S, K1F, NSEG, HT, OT, EPI, SINGLE, POST, HEAD>(a, stream, GY, form)

}  // namespace

namespace gw {

int chainx3_launch(int kind, ChainArgs& a, int k_in, int hidden, int n_out, int grid_y, void* stream) {
#ifdef GW_TUNING
  if (g_dbg != nullptr && kind == g_dbg_kind) {  // gw_debug_timestamps: kinds 0 mlp, 1 edge, 2 node update (+ 4, 6), 3 project
    a.dbg = g_dbg;
    a.dbg_cap = g_dbg_cap;
  }
#endif
  static const int f_rows = GW_TUNE("GW_X3_FORM", 41), f_edge = GW_TUNE("GW_X3_FORM_EDGE", 41);
  const int form = kind == 1 ? f_edge : f_rows;
  switch (kind) {
    case 0:  // mlp rows
      if (hidden == 256 && n_out == 256) {
        if (k_in <= 32) GW_X3(1, false, 1, 16, 16, EPI_ROWS, false, false, false, 1);
        if (k_in <= 128) GW_X3(4, false, 1, 16, 16, EPI_ROWS, false, false, false, 1);
        if (k_in == 256) GW_X3(8, true, 1, 16, 16, EPI_ROWS, false, false, false, 1);
      } else if (hidden == 256 && n_out <= 80 && k_in == 256) {
        GW_X3(8, true, 1, 16, 5, EPI_DEC, false, false, false, 1);
      } else if (hidden == 128 && n_out <= 80 && k_in == 256) {
        GW_X3(8, true, 1, 8, 5, EPI_DEC, false, false, false, 1);
      }
      return set_error(GW_E_UNSUPPORTED, "bf16x3 mlp: unsupported (hidden, n_out, k) combination");
    case 1:  // edge update
      GW_X3(8, true, 3, 16, 16, EPI_EDGE, false, false, false, 1);
    case 2:  // node update
      GW_X3(8, true, 2, 16, 16, EPI_ROWS, false, false, false, 1);
    case 3:  // projections (grid_y slices)
      GW_X3(8, true, 1, 16, 16, EPI_ROWS, true, false, false, grid_y);
    case 4:  // node update + POST products
      GW_X3(8, true, 2, 16, 16, EPI_ROWS, false, true, false, 1);
    case 6:  // node update + output head (decoder)
      GW_X3(8, true, 2, 16, 16, EPI_ROWS, false, false, true, 1);
    case 5:  // mlp rows + POST products of the output rows (node encoder -> layer-1 products of the encoder's edge MLP)
      if (hidden == 256 && n_out == 256 && k_in <= 128 && k_in > 32) GW_X3(4, false, 1, 16, 16, EPI_ROWS, false, true, false, 1);
      return set_error(GW_E_UNSUPPORTED, "bf16x3 mlp + post products: hidden 256, 256 outputs, 33..128 inputs");
  }
  return set_error(GW_E_BADARG, "chainx3_launch: bad kind");
}

int bwd_chainx3_launch(const BwdChainArgs& a, void* stream) {
  const long long grid = (a.n_rows + 63) / 64;
  constexpr int lds = 2 * buf_bytes(4) + kBwdLnScratch;
  if (a.ln_y != nullptr) {
    static DeviceOnce once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)bwd_chainx3_kernel<kRing, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((bwd_chainx3_kernel<kRing, true>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
    return check_launch("bwd_chainx3_kernel launch");
  }
  static DeviceOnce once;
  if (once.first()) (void)hipFuncSetAttribute((const void*)bwd_chainx3_kernel<kRing, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((bwd_chainx3_kernel<kRing, false>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
  return check_launch("bwd_chainx3_kernel launch");
}

void pack_x3_item(const float* w, long long sf, long long sk, int n_out, int kseg, int ntp, int nsteps, void* out, void* stream) {
  const size_t total = (size_t)nsteps * ntp * 512;
  int grid = (int)((total + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(pack_linear_x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, sf, sk, n_out, kseg, ntp, nsteps, (__bf16*)out);
}

}  // namespace gw

extern "C" {

// K-steps (32 input features each) of a packed slice: as gw_packed_bytes_bf16 (1, 4 or k / 32 steps)
static int packed_steps_x3(int kseg) { return kseg <= 32 ? 1 : (kseg <= 128 ? 4 : (kseg + 31) / 32); }

size_t gw_packed_bytes_bf16x3(int n_out, int k_lo, int k_hi) {
  const int kseg = k_hi - k_lo;
  const int nsteps = packed_steps_x3(kseg);
  const int ntp = (((n_out + 15) / 16) + 3) / 4 * 4;
  return (size_t)nsteps * ntp * 2048;
}

int gw_pack_linear_bf16x3(const float* w, int n_out, int k_total, int k_lo, int k_hi, void* out, void* stream) {
  if (!w || !out || n_out <= 0 || k_lo < 0 || k_hi <= k_lo || k_hi > k_total)
    return gw::set_error(GW_E_BADARG, "gw_pack_linear_bf16x3: bad arguments");
  const int kseg = k_hi - k_lo;
  gw::pack_x3_item(w + k_lo, k_total, 1, n_out, kseg, (((n_out + 15) / 16) + 3) / 4 * 4, packed_steps_x3(kseg), out, stream);
  return gw::check_launch("pack_linear_x3_kernel launch");
}

}  // extern "C"
