// gw_edge16p.hip - processor-block form of the team-pipelined bf16 edge update on SEGMENT-ALIGNED tiles (BASELINE.json configs[2]).
//
//   e'[c] = LayerNorm(W_out . relu(W_mid . h1[c] + b_mid) + b_out) + e[c],   agg[dst(c)] += LayerNorm(.)[c]
//   (graph_net_block.py:131-137 and the scatter_sum of :188 inside GraphProcessor's loop, :293-301; h1 = the layer-1
//   activations edge16_l1_kernel left in the workspace, e / e' = bf16 edge tiles)
//
// The DMA form of gw_edge16t.hip spends most of a tile on team B: LayerNorm + residual + staging through LDS (4.7 k cycles per
// 64-edge tile), a per-column segment walk on both teams (2 x 2.8 k) and the output layer (3.6 k), against 3.3 k of middle layer
// on team A (profiles/r03_team_timeline_processor.log).  On segment-aligned tiles (GW_EDGE_SEGMENT_TILES: no destination's run
// of edges crosses a tile - the latent mesh graph, encoder.py:244-268, has 7 or 6 edges per node) this kernel runs team B as the
// decoder form does - output layer TRANSPOSED (accumulators = lane (feature, q) x 4 edges), LayerNorm statistics over the 16
// feature lanes on DPP rotations, the segment sums as one more matrix product y^T . S - and splits the rest differently:
//
//   * the aggregate is a RUNNING SUM over the blocks: e_n = e'_(n-1), so sum_seg e_n is the previous block's aggregate and
//     agg_n = agg_(n-1) + sum_seg LayerNorm_n(.).  Team B adds its product to the rows of the previous block's aggregate in place
//     (complete segments: plain load + add + store, no atomics, no zero fill) - the residual never enters the segment sums, and
//     the aggregate no longer carries the bf16 rounding of the stored e;
//   * e' needs the LayerNorm output in the lane = edge layout of the tiles: team B parks it (fp32) in a transposition scratch in
//     LDS, one ds_write_b128 per 4 edges; team A - which has no segment walk any more - reads it back in the tile layout,
//     adds the residual tile (requested before its middle layer) and stores the e' tile.
//
// Per step s of a workgroup (three tiles in flight, two barriers, as in gw_edge16t.hip):
//   half 1:  A: [residual of tile s-1 requested] middle layer of tile s           B: statistics, y, scratch, segment sums of tile s-1
//   half 2:  A: DMA of tile s+1, e' of tile s-1 from the scratch, slot tables      B: output layer of tile s (transposed)

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_edge16.hpp"
#include "gw_edge16t.hpp"

using namespace gw;
using namespace gw16;
using namespace gw16t;

namespace {

constexpr int kScrLd = 68;                                   // floats per feature row of the scratch: 64 edges + 4 (bank shift)
constexpr int kP_H1 = 0;
constexpr int kP_H2 = kHBytes;
constexpr int kP_Scr = 2 * kHBytes;                          // float[4 waves][64 features][kScrLd]
constexpr int kP_Ln = kP_Scr + 4 * 64 * kScrLd * 4;          // float[2][4 waves][64 edges]: partial sums, sums of squares
constexpr int kP_Par = kP_Ln + 2048;                         // float[256]: b_mid
constexpr int kP_Dsl = kP_Par + 1024;                        // int[4][64]: destination row of each slot of a tile
constexpr int kP_Slot = kP_Dsl + 1024;                       // uint8[4][64]: slot of each column (255 = padding)
constexpr int kP_Nsl = kP_Slot + 256;                        // int[4]
constexpr int kP_Lnc = kP_Dsl + 2048;                        // float2[4 waves][64 edges]: (rstd, -mean rstd)
constexpr int kP_ParT = kP_Lnc + 2048;                       // float4[4 waves][16 lanes][4 t]: b_out replicated x 4
constexpr int kP_ParG = kP_ParT + 4096;                      // float4[2][4 waves][16 lanes]: gamma, beta of the lane's features (t = 0..3)
constexpr int kP_Smat = kP_ParG + 2048;                      // uint4[4 ring][2 halves][64 lanes]: S (one slot group: <= 16 slots per tile)
constexpr int kP_Total = kP_Smat + 4 * 2 * 64 * 16;
static_assert(kP_Total <= 160 * 1024, "LDS budget of one CU");

// scratch row of local feature f (0..63 of a team-B wave): rows f and f + 16 would start in the same banks for the writers'
// ds_write_b128 (8 consecutive lanes = two quads of rows 16 apart) - the odd 16-blocks swap their halves of 4
__device__ __forceinline__ constexpr int scr_row(int f) { return f ^ (((f >> 4) & 1) << 2); }

template <bool EOUT>
__global__ __launch_bounds__(512, 2) void edge16p_kernel(const Edge16Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool team_b = wave >= 4;
  const int tw = wave & 3;
  const int j = lane & 15;
  const int q = lane >> 4;
  const int f0 = 64 * tw + 4 * q;  // team A: this lane's features of the middle layer: f0 + 16 t + r
  const int s0 = 2 * tw;           // K-steps the wave's rows fill / the wave's e' slots

  // team B: output tile t of the wave, column m = lane & 15 is feature 64 tw + 16 (m >> 2) + 4 t + (m & 3): a segment-sum
  // result lane (slot, q') then holds the 16 consecutive features 64 tw + 16 q' .. of its destination row
  auto feat_of = [&](int t) -> int { return 64 * tw + 16 * (j >> 2) + 4 * t + (j & 3); };
  bf16x8 wr[4][8];
  if (team_b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int f = feat_of(t);
#pragma unroll
      for (int s = 0; s < 8; ++s) wr[t][s] = *(const bf16x8*)(a.w_out + ((size_t)(s * 16 + (f >> 4)) * 64 + 16 * q + (f & 15)) * 16);
    }
    if (q == 0) {
      f32x4 pg, pb;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int f = feat_of(t);
        const float b = a.b_out[f];
        ((f32x4*)(lds + kP_ParT))[(tw * 16 + j) * 4 + t] = f32x4{b, b, b, b};
        pg[t] = a.gamma[f];
        pb[t] = a.beta[f];
      }
      ((f32x4*)(lds + kP_ParG))[tw * 16 + j] = pg;
      ((f32x4*)(lds + kP_ParG))[64 + tw * 16 + j] = pb;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 8; ++s) wr[t][s] = *(const bf16x8*)(a.w_mid + ((size_t)(s * 16 + 4 * tw + t) * 64 + lane) * 16);
    ((float*)(lds + kP_Par))[threadIdx.x] = a.b_mid[threadIdx.x];
  }
  char* const h1 = lds + kP_H1;
  char* const h2 = lds + kP_H2;
  float* const scr = (float*)(lds + kP_Scr);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  // tile list: XCD x = workgroup & 7 owns a contiguous range of edge blocks, walked batch-innermost (as gw_edge16t.hip, bc = 1)
  const int slot_wg = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const TileWalk twk = tile_walk(blockIdx.x & 7, a.neb, a.batch);
  const int n = twk.n_units > slot_wg ? (twk.n_units - slot_wg + nslot - 1) / nslot : 0;
  auto tile_at = [&](int i) -> TileId {
    const int u = slot_wg + i * nslot;
    const int ebl = u / a.batch;
    return TileId{twk.eb_start + ebl, u - ebl * a.batch};
  };
  auto tile_row = [&](TileId t) -> size_t { return (size_t)(t.b * a.neb + t.eb); };

  bool stamp = false;
  const int ts_thread = team_b ? 256 : 0, ts_base = (int)blockIdx.x * 32 + (team_b ? 16 : 0);
#ifdef GW_TUNING  // (phase clocks exist in tuning builds only: a scalar branch per stamp is not free in a loop bound by what a wave can issue)
#define GW_TS(i)                                                      \
  if (stamp) {                                                        \
    const unsigned long long c_ = gw_clock();                         \
    if ((int)threadIdx.x == ts_thread) a.dbg[ts_base + (i)] = c_;     \
  }
#else
#define GW_TS(i)
#endif

  if (n == 0) return;
  __syncthreads();  // parameter blocks visible

  if (!team_b) {
    // ================================================ team A ========================================================
    auto prep_dma_issue = [&](TileId t) {  // 32 KiB of layer-1 activations -> Hbuf1: 8 LDS-DMA pieces of 1 KiB per wave
      const char* src = a.h1g + tile_row(t) * kHBytes;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int piece = 8 * tw + i;
        glds16_asm_s((const float*)(src + piece * 1024), (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kP_H1 + piece * 1024)));
      }
    };
    auto publish = [&](TileId t, int ring) {  // (wave 0) slot tables + the S operand of the tile's segment sums
      const int kr = t.eb * kTileCols + lane;
      const int d = ldgi(a.dst + kr);
      const int dp = lane > 0 ? ldgi(a.dst + kr - 1) : -2;
      const bool valid = d >= 0;
      const bool start = valid && d != dp;
      const unsigned long long sm = __ballot(start);
      const int slot = __popcll(sm & ((2ull << lane) - 1ull)) - 1;
      // (the host guarantees at most 16 slots per tile for this form; a column beyond that would be dropped, never mis-added)
      ((unsigned char*)(lds + kP_Slot))[ring * kTileCols + lane] = (unsigned char)((valid && slot < 16) ? slot : 255);
      if (start && slot < 16) ((int*)(lds + kP_Dsl))[ring * kTileCols + slot] = t.b * a.n_dst + d;
      const int nsl_ = __popcll(sm) < 16 ? __popcll(sm) : 16;
      if (lane == 0) ((int*)(lds + kP_Nsl))[ring] = nsl_;
      const unsigned* const sw = (const unsigned*)(lds + kP_Slot) + ring * 16;
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4* const sm_out = (u32x4*)(lds + kP_Smat) + ring * (2 * 64);
      const unsigned me = (unsigned)(lane & 15);
      const int qq = lane >> 4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned w0 = sw[8 * h + qq], w1 = sw[8 * h + 4 + qq];
        u32x4 pk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          pk[i] = ((((w0 >> (16 * i)) & 255u) == me) ? 0x3F80u : 0u) | ((((w0 >> (16 * i + 8)) & 255u) == me) ? 0x3F800000u : 0u);
          pk[2 + i] = ((((w1 >> (16 * i)) & 255u) == me) ? 0x3F80u : 0u) | ((((w1 >> (16 * i + 8)) & 255u) == me) ? 0x3F800000u : 0u);
        }
        sm_out[h * 64 + lane] = pk;
      }
    };
    const float* const par_l = (const float*)(lds + kP_Par);
    TileId t_next = tile_at(0);
    prep_dma_issue(t_next);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tw == 0) publish(t_next, 0);
    TileId t_cur = t_next, t_prev = t_next;
#pragma unroll 1
    for (int s = 0; s <= n; ++s) {
      t_prev = t_cur;
      t_cur = t_next;
      if (s + 1 < n) t_next = tile_at(s + 1);
      stamp = a.dbg != nullptr && s == 3 && (int)blockIdx.x < a.dbg_cap;
      const bool has_next = s + 1 < n;
      GW_TS(0)
      team_barrier();  // (alpha) Hbuf1 of tile s complete; the scratch of tile s - 2 has been read
      GW_TS(1)
      // the residual of tile s - 1 (this wave's 16-byte slots of K-steps s0, s0 + 1 in all 4 groups): requested here, used in
      // half 2 - the middle layer covers the round trip
      bf16x8 rest[kGroups][2];
      if (EOUT && s >= 1) {
        const char* rb = a.res_tiles + tile_row(t_prev) * kHBytes + (size_t)s0 * 1024 + (size_t)(unsigned)(fresh(lane) * 16);
#pragma unroll
        for (int g = 0; g < kGroups; ++g)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) rest[g][ks] = *(const GW_AS1 bf16x8*)(rb + (size_t)g * 8192 + ks * 1024);
      }
      if (s < n) {
        // ---- middle layer of tile s: Hbuf1 -> Hbuf2 ----
        f32x4 acc[2][4];
        unsigned pk[8];
        __builtin_amdgcn_s_setprio(1);
        team_layer<2, false, false, 18, true>(
            acc, wr, h1, lane, [&](f32x4& dst, int t) { dst = *(const f32x4*)(par_l + fresh(f0) + 16 * t); },  // b_mid
            [&](int g, int m, f32x4 (&ac)[4]) {
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              if (m < 16) {  // bf16 pack of a pair, then relu on the packed pair (a negative bf16 is a negative 16-bit integer)
                const int pi = m >> 1;
                if ((m & 1) == 0) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[pi]) : "v"(ac[pi >> 1][2 * (pi & 1)]), "v"(ac[pi >> 1][2 * (pi & 1) + 1]));
                else asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[pi]) : "v"(pk[pi]));
              } else if (m == 24) {
                *(u32x4*)(h2 + ((g * 8 + s0) * 64 + fresh(lane)) * 16) = u32x4{pk[0], pk[1], pk[2], pk[3]};
              } else if (m == 25) {
                *(u32x4*)(h2 + ((g * 8 + s0 + 1) * 64 + fresh(lane)) * 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
              }
            });
        __builtin_amdgcn_s_setprio(0);
        GW_TS(2)
      }
      team_barrier();  // (beta) Hbuf2 of tile s and the scratch of tile s - 1 complete; Hbuf1 free
      GW_TS(7)
      if (has_next) prep_dma_issue(t_next);  // in flight under the e' work
      if (EOUT && s >= 1) {
        // ---- e' of tile s - 1: LayerNorm output from the scratch (tile layout: this lane = edge 16 g + j, features k(s0 + ks, q, i))
        // + residual -> bf16 -> the lane's own 16-byte slots of the e' tile ----
        const float* const sb0 = scr + (tw * 64 + 4 * fresh(q)) * kScrLd + fresh(j);         // rows of even 16-blocks (i < 4)
        const float* const sb1 = scr + (tw * 64 + 4 * (fresh(q) ^ 1)) * kScrLd + fresh(j);   // rows of odd 16-blocks (i >= 4): halves swapped
        char* const ob = a.e_out_tiles + tile_row(t_prev) * kHBytes + (size_t)s0 * 1024 + (size_t)(unsigned)(fresh(lane) * 16);
#pragma unroll
        for (int g = 0; g < kGroups; ++g)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            // (scalar adds spelled out: left to the SLP vectoriser they become v_pk_add_f32 on register PAIRS it first has to
            //  assemble with two moves each - three instructions for two adds in a phase that is bound by instruction issue)
            f32x4 lo, hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              asm("v_add_f32 %0, %1, %2" : "=v"(lo[r]) : "v"(sb0[(32 * ks + r) * kScrLd + 16 * g]), "v"((float)rest[g][ks][r]));
              asm("v_add_f32 %0, %1, %2" : "=v"(hi[r]) : "v"(sb1[(32 * ks + 16 + r) * kScrLd + 16 * g]), "v"((float)rest[g][ks][4 + r]));
            }
            *(GW_AS1 bf16x8*)(ob + (size_t)g * 8192 + ks * 1024) = to_bf16x8(lo, hi);
          }
        GW_TS(9)
      }
      if (has_next) {
        if (tw == 0) publish(t_next, (s + 1) & 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA pieces have landed (and the e' stores have left)
      }
      GW_TS(11)
    }
  } else {
    // ================================================ team B ========================================================
    f32x4 o[kGroups][4];  // o[g][t][r]: feature feat_of(t), edge 16 g + 4 q + r
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) o[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* const ln1 = (float*)(lds + kP_Ln);
    float* const ln2 = ln1 + 4 * kTileCols;
    float* const lnc = (float*)(lds + kP_Lnc);
    const f32x4* const parT = (const f32x4*)(lds + kP_ParT);
    const f32x4* const parG = (const f32x4*)(lds + kP_ParG);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* const smat = (const u32x4*)(lds + kP_Smat);
    const int* const dsl = (const int*)(lds + kP_Dsl);
    const int* const nsl = (const int*)(lds + kP_Nsl);
    // the rows of the running aggregate this lane will add to (tile s: requested at the end of step s, behind the output layer -
    // the slot tables of tile s were published a step earlier - and added in half 1 of step s + 1: an HBM round trip under cover)
    f32x4 old[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) old[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int s = 0; s <= n; ++s) {
      stamp = a.dbg != nullptr && s == 3 && (int)blockIdx.x < a.dbg_cap;
      GW_TS(0)
      team_barrier();  // (alpha) LayerNorm partial sums of tile s - 1 visible; team A has read the scratch of tile s - 2
      GW_TS(1)
      if (s >= 1) {
        const int ring = (s - 1) & 3;
        {  // statistics: lane (j, q) combines the four waves' partial sums of ONE edge and parks (rstd, -mean rstd) for its wave
          const int e_mine = fresh(16 * (j >> 2) + 4 * q + (j & 3));
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            t1 += ln1[w * kTileCols + e_mine];
            t2 += ln2[w * kTileCols + e_mine];
          }
          const float mean = t1 * (1.0f / 256.0f);
          const float var = fmaxf(t2 * (1.0f / 256.0f) - mean * mean, 0.f);
          const float ga = __builtin_amdgcn_rsqf(var + 1e-5f);
          *(float2*)(lnc + (tw * kTileCols + e_mine) * 2) = float2{ga, -mean * ga};
        }
        // y = ((o - mean) rstd) gamma + beta: (fp32) into the transposition scratch for team A's e' tile, (bf16) into the A operand
        // of the segment-sum product; group by group, the next group's statistics requested a group ahead
        u32x4 ypk[4][2];
        {
          const f32x4 gm4 = parG[tw * 16 + fresh(j)];
          const f32x4 bt4 = parG[64 + tw * 16 + fresh(j)];
          f32x4 gnx0 = *(const f32x4*)(lnc + (tw * kTileCols + 4 * fresh(q)) * 2);
          f32x4 gnx1 = *(const f32x4*)(lnc + (tw * kTileCols + 4 * fresh(q) + 2) * 2);
          // scratch rows of this lane's features: local feature 16 (j >> 2) + 4 t + (j & 3) -> row scr_row(.), + 4 q + 16 g columns
          // this lane's features: local feature fl + 4 t, fl = 16 (j >> 2) + (j & 3), scratch row scr_row(.) = fl + 4 (t ^ b) with
          // b = bit 4 of fl: TWO row addresses for the tile (rows of even / odd t), everything else is an immediate offset
          const int fl = 16 * (fresh(j) >> 2) + (fresh(j) & 3), fb = (fresh(j) >> 2) & 1;
          float* const swr_e = scr + (tw * 64 + fl + 4 * fb) * kScrLd + 4 * fresh(q);
          float* const swr_o = scr + (tw * 64 + fl + 4 * (1 - fb)) * kScrLd + 4 * fresh(q);
#pragma unroll
          for (int g = 0; g < kGroups; ++g) {
            const f32x4 gab0 = gnx0, gab1 = gnx1;  // (rstd, -mean rstd) x edges 4 q + (0, 1) and + (2, 3)
            if (g + 1 < kGroups) {
              gnx0 = *(const f32x4*)(lnc + (tw * kTileCols + 16 * (g + 1) + 4 * fresh(q)) * 2);
              gnx1 = *(const f32x4*)(lnc + (tw * kTileCols + 16 * (g + 1) + 4 * fresh(q) + 2) * 2);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              f32x4 y;
              // (scalar FMAs spelled out, as the adds of the e' tile below: no v_pk_fma_f32 on assembled register pairs)
              y[0] = fma_s(fma_s(o[g][t][0], gab0[0], gab0[1]), gm4[t], bt4[t]);
              y[1] = fma_s(fma_s(o[g][t][1], gab0[2], gab0[3]), gm4[t], bt4[t]);
              y[2] = fma_s(fma_s(o[g][t][2], gab1[0], gab1[1]), gm4[t], bt4[t]);
              y[3] = fma_s(fma_s(o[g][t][3], gab1[2], gab1[3]), gm4[t], bt4[t]);
              if (EOUT) *(f32x4*)((t & 1 ? swr_o : swr_e) + (t >> 1) * 8 * kScrLd + 16 * g) = y;  // (scratch row scr_row(fl + 4 t))
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ypk[t][g >> 1][2 * (g & 1)]) : "v"(y[0]), "v"(y[1]));
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ypk[t][g >> 1][2 * (g & 1) + 1]) : "v"(y[2]), "v"(y[3]));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        GW_TS(2)
        // segment sums: D[feature][slot] = sum_edges y^T . S, added to the rows of the previous block's aggregate in place
        const int nslots = __builtin_amdgcn_readfirstlane(nsl[ring]);
        const bool mine = j < nslots && GW_SKIP(a) != 1;
        float* const dstp = a.agg + (size_t)(mine ? dsl[ring * kTileCols + fresh(j)] : 0) * 256 + fresh(64 * tw + 16 * q);
        u32x4 sb[2];
        sb[0] = smat[(ring * 2 + 0) * 64 + fresh(lane)];
        sb[1] = smat[(ring * 2 + 1) * 64 + fresh(lane)];
        f32x4 dsum[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dsum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_nop 7" : "+v"(dsum[0]), "+v"(dsum[1]), "+v"(dsum[2]), "+v"(dsum[3]), "+v"(sb[0]), "+v"(sb[1]));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(dsum[t]) : "v"(ypk[t][h]), "v"(sb[h]));
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(dsum[0]), "+v"(dsum[1]), "+v"(dsum[2]), "+v"(dsum[3]));
        if (mine) {
#pragma unroll
          for (int t = 0; t < 4; ++t) stg4(dstp + 4 * t, old[t] + dsum[t]);
        }
        GW_TS(3)
      }
      team_barrier();  // (beta) Hbuf2 of tile s complete; the scratch of tile s - 1 complete
      GW_TS(7)
      if (s < n) {
        // ---- output layer of tile s, transposed; per group the sums / sums of squares over the wave's 64 features ----
        float s1[4], s2[4], ra[4], rb[2], rc[2], rd[2];  // (running sums of 4 edges; registers of their 16-lane reduce-scatter)
        __builtin_amdgcn_s_setprio(1);
        team_layer<4, true, false, 4, true>(
            o, wr, h2, lane, [&](f32x4& dst, int t) { dst = parT[(tw * 16 + fresh(j)) * 4 + t]; },
            [&](int g, int mm, f32x4 (&ac)[4]) {
              if (mm < 16) {
                const int t = mm >> 2, r = mm & 3;
                const float x = ac[t][r];
                s1[r] = t == 0 ? x : s1[r] + x;
                s2[r] = t == 0 ? x * x : fmaf(x, x, s2[r]);
              } else {
                reduce_ops(g, mm, s1, s2, ra, rb, rc, rd);
                if (mm == 27 && (j & 3) == 0) {  // bank b = j >> 2 holds the sums (2 (b & 1), + 1) of s1 (b < 2) / s2: edges 4 q + ...
                  *(float2*)(ln1 + (fresh(j) >> 3) * (4 * kTileCols) + tw * kTileCols + 16 * g + 4 * fresh(q) + 2 * ((fresh(j) >> 2) & 1)) = float2{rd[0], rd[1]};
                }
              }
            });
        __builtin_amdgcn_s_setprio(0);
        {
          const int ring_s = s & 3;
          const bool mine_s = j < __builtin_amdgcn_readfirstlane(nsl[ring_s]) && GW_SKIP(a) != 1;
          if (mine_s) {
            const float* src = a.agg + (size_t)dsl[ring_s * kTileCols + fresh(j)] * 256 + fresh(64 * tw + 16 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) old[t] = ldg4(src + 4 * t);
          }
        }
        GW_TS(9)
      } else {
        // (last iteration: redefine the accumulators from nothing - see gw_edge16t.hip)
#pragma unroll
        for (int g = 0; g < kGroups; ++g)
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" : "=v"(o[g][t]));
      }
      GW_TS(13)
    }
  }
#undef GW_TS
}

template <typename K>
int launch_p(K kernel, int n_wg, const Edge16Args& a, void* stream) {
  static DeviceOnce once;
  if (once.first()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kP_Total);
  hipLaunchKernelGGL(kernel, dim3((unsigned)n_wg), dim3(512), kP_Total, (hipStream_t)stream, a);
  return check_launch("edge16p_kernel launch");
}

}  // namespace

namespace gw {

// Processor-block form on segment-aligned tiles: layer-1 tiles from the workspace, residual = per-sample bf16 edge tiles, e' as
// bf16 edge tiles (or dropped), aggregate accumulated onto the caller's rows (the previous block's aggregate).
int edge16p_launch(const void* edge16_args, int n_wg, void* stream) {
  const Edge16Args& a = *(const Edge16Args*)edge16_args;
  if (!a.seg_tiles || !a.h1g || !a.agg || a.agg_bf16k || a.e_out || (a.e_out_tiles && (!a.res_tiles || a.res_tiles_shared)))
    return set_error(GW_E_UNSUPPORTED, "edge16p: layer-1 tiles from the workspace, per-sample residual tiles, fp32 aggregate rows, e' as tiles or dropped");
  return a.e_out_tiles ? launch_p(edge16p_kernel<true>, n_wg, a, stream) : launch_p(edge16p_kernel<false>, n_wg, a, stream);
}

}  // namespace gw
