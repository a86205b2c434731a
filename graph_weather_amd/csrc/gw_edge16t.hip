// gw_edge16t.hip - team-pipelined form of the bf16 edge update with on-chip weights (BASELINE.json configs[2]).
//
//   e'[c] = LayerNorm(W_out . relu(W_mid . relu(z1[c]) + b_mid) + b_out) + e[c],   agg[dst(c)] += e'[c]
//   (graph_net_block.py:131-137 + the scatter_sum of :188; z1 = layer-1 pre-activations, see gw_edge16.hip)
//
// The lock-step kernel of gw_edge16.hip walks every 64-edge tile through gather | middle layer | output layer | LayerNorm |
// segment sums with all 8 waves in the same phase: its phase clocks (profiles/r02_edge16_timeline_*.log) show ~4 k of the
// 16-21 k cycles of a tile on the matrix cores - while the waves run LayerNorm or walk segments the MFMA pipe idles, and while
// they issue MFMAs the vector ALU / LDS / memory pipes idle.  Here the workgroup's 8 waves form two TEAMS with one matrix each:
//
//     team A = waves 0-3 (one per SIMD): W_mid resident in AGPRs (rows 64 w ..), produces Hbuf2 from Hbuf1, and prepares
//              Hbuf1 of the tile after (GATHER: relu(b1 + sum of projected rows); else LDS-DMA of the layer-1 tiles a
//              previous launch left in the workspace);
//     team B = waves 4-7 (the other wave of each SIMD): W_out resident, output layer, LayerNorm, residual, e' tile store,
//              staging for the segment sums.
//
// A step of the pipeline is two half-steps separated by workgroup barriers; in each half one team is on the matrix pipe and the
// other on the vector / LDS / memory pipes of the same SIMDs:
//
//     half 1 (s):   A: middle layer of tile s   (Hbuf1 -> Hbuf2)         B: LayerNorm + residual + staging of tile s - 1
//     half 2 (s):   A: segment sums of tile s - 1 (columns 0..31),        B: segment sums of tile s - 1 (columns 32..63),
//                      then Hbuf1 of tile s + 1                              then output layer of tile s (Hbuf2 -> registers)
//
// so three tiles are in flight per workgroup.  Every buffer has ONE producer half-step and ONE consumer half-step with a
// barrier between them (Hbuf1: A.h2 -> A.h1; Hbuf2: A.h1 -> B.h2; LayerNorm partial sums: B.h2 -> B.h1; staged tile: B.h1 ->
// both .h2), so none is double buffered.  Tiles are walked XCD-aware as in gw_edge16.hip, and batch-innermost PER WORKGROUP in
// chunks of `bc` batch elements of one edge block: what an edge block shares across the batch (row indices, the cached
// per-edge products We.e + b1 of encoder / decoder / first processor block) is fetched once per chunk and kept in registers
// (as fp16 pairs: 11 significant bits in front of a bf16 rounding).
// The residual e comes from bf16 edge tiles; segment sums meet in fp32 atomics where a run crosses a tile or the middle of
// one (the deterministic mode stays on the lock-step 4-wave kernel).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_edge16.hpp"
#include "gw_edge16t.hpp"  // team_barrier, fresh, relu1, team_layer, TileId

using namespace gw;
using namespace gw16;
using namespace gw16t;

namespace {

constexpr int kT_H1 = 0;
constexpr int kT_H2 = kHBytes;
constexpr int kT_Stage = 2 * kHBytes;
constexpr int kT_Gd = kT_Stage + kTileCols * kStageLd * 4;  // destination rows of 4 tiles in flight (ring)
constexpr int kT_Ln = kT_Gd + 4 * kTileCols * 4;            // [team-B wave][column] (sum, sum of squares)
constexpr int kT_Par = kT_Ln + 4 * kTileCols * 8;           // b_mid, b_out, gamma, beta, b1
constexpr int kT_Zc = kT_Par + 5 * 256 * 4;                 // GATHER: cached batch-shared layer-1 part of column pass 1 (fp16), thread private
constexpr int kT_Total = kT_Zc + 8 * 256 * 8;
static_assert(kT_Total <= 160 * 1024, "LDS budget of one CU");
// Segment-aligned form (SEGT): no staged tile - the area holds the slot tables of the 4 tiles in flight and small exchange buffers
constexpr int kS_Dsl = kT_Stage;             // int[4][64]: destination row (global) of each destination slot of a tile
constexpr int kS_Slot = kS_Dsl + 1024;       // uint8[4][64]: destination slot of each column (255 = padding column)
constexpr int kS_Nsl = kS_Slot + 256;        // int[4]: number of destination slots of the tile
constexpr int kS_Lnc = kT_Stage + 2048;      // float2[4 team-B waves][64 edges]: (rstd, -mean rstd), private to the wave
constexpr int kS_ParT = kT_Stage + 4096;     // float4[4 waves][16 lanes][4 t]: b_out of a lane's feature of tile t, replicated x 4
constexpr int kS_ParP = kT_Stage + 8192;     // float[2][256]: gamma, beta in the POSITION order of a destination row
constexpr int kS_Cnt = kT_Stage + 10240;     // float[4][64]: edges of each destination slot
constexpr int kS_Part = kT_Stage + 11264;    // int[4][64]: 1 = the slot is a piece of a run that continues in a neighbouring tile
constexpr int kS_Smat = kT_Stage + 12288;    // uint4[4 ring][4 slot groups][2 halves][64 lanes]: the B operand S of the segment sums
constexpr int kS_Zc0 = kS_Smat + 4 * 4 * 2 * 64 * 16;  // half4[8][256 threads]: the cached layer-1 part of column pass 0 (as kT_Zc holds pass 1's)
static_assert(kS_Zc0 + 8 * 256 * 8 <= kT_Stage + kTileCols * kStageLd * 4, "the slot tables fit the staging area");

// GATHER form: exactly one projected table has a row set per batch element (the decoder's P_s); PH: that table is fp16 rows
// (GW_LAYOUT_ROWS_F16).  The other projected tables are fp32 rows shared by the batch and cached per chunk.
// RES: the residual e of graph_net_block.py:135 is added from bf16 edge tiles (true), or not at all (false: callers that only
// want the aggregate - the decoder - add the segment sums of their batch-shared e into the aggregate buffer beforehand:
// sum(LN(.) + e) = sum(LN(.)) + sum(e), and sum(e) per destination is the same for every batch element).
// SEGT (GW_EDGE_SEGMENT_TILES; GATHER, no residual): the caller's edge list is padded so that no destination's run of edges
// crosses a tile (dst < 0 marks padding columns; the decoder graph has 7 or 6 edges per grid node: 9 nodes = 63 columns per
// tile).  Team B then runs its output layer TRANSPOSED (mfma_t: accumulators = lane (feature, q) x 4 edges), applies LayerNorm
// in that layout (row sums over the 16 feature lanes on DPP rotations, parameters one float4 per lane) and computes the segment
// sums as one more matrix product - y^T (bf16) . S, S[edge][slot] = 1 where the edge belongs to the tile's destination slot -
// whose result, lane (slot, q) x 16 consecutive features, is stored straight to the destination rows: no staged tile in LDS, no
// per-column walk, no atomics (every segment is complete inside its tile), and optionally as bf16 rows in the K order the node
// update's matrix product reads them in (GW_LAYOUT_ROWS_BF16K: what it would round them to anyway).
template <bool GATHER, bool PH, bool RES, bool SEGT = false>
__global__ __launch_bounds__(512, 2) void edge16t_kernel(const Edge16Args a) {
  static_assert(!SEGT || (GATHER && !RES), "the segment-aligned form gathers layer 1 and adds no residual");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool team_b = wave >= 4;
  const int tw = wave & 3;              // wave within its team
  const int j = lane & 15;
  const int q = lane >> 4;
  const int f0 = 64 * tw + 4 * q;       // this lane's features of its team's layer: f0 + 16 t + r, t < 4
  const int s0 = 2 * tw;                // K-steps of the B layout the wave's 4 row tiles fill: s0, s0 + 1

  // ---- resident weights: rows 64 tw .. of the team's matrix, all 8 K-steps (packed stream: [s][16 tiles][lane][8]) ----
  // SEGT, team B: output tile t of the wave, column m = lane & 15 (= row 4 q' + r of the segment-sum result) is the feature at
  // POSITION 64 tw + 16 (m >> 2) + 4 t + (m & 3) of a destination row, so that a result lane holds 16 consecutive positions;
  // position -> feature is the identity for fp32 rows and the K order of the packed streams for GW_LAYOUT_ROWS_BF16K
  // (position 32 s + 8 q + i holds feature 32 s + 16 (i >> 2) + 4 q + (i & 3)).
  auto feat_of = [&](int t) -> int {
    const int pos = 64 * tw + 16 * (j >> 2) + 4 * t + (j & 3);
    if (!a.agg_bf16k) return pos;
    const int i = pos & 7;
    return (pos & ~31) + 16 * (i >> 2) + 4 * ((pos >> 3) & 3) + (i & 3);
  };
  // F16MID (segment-aligned form with the per-sample products as fp16 rows): layer 1 = relu(fp16 row + cached fp16 part) stays in
  // packed fp16 arithmetic (a third of the instructions of widen / add / relu / round-to-bf16: the gather is team A's critical
  // phase) and the middle layer runs on fp16 operands - W_mid's bf16 values are exact in fp16 (|w| >= 6e-5), converted once here
  constexpr bool F16MID = SEGT && PH;
  bf16x8 wr[4][8];
  if (SEGT && team_b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int f = feat_of(t);
#pragma unroll
      for (int s = 0; s < 8; ++s) wr[t][s] = *(const bf16x8*)(a.w_out + ((size_t)(s * 16 + (f >> 4)) * 64 + 16 * q + (f & 15)) * 16);
    }
    if (q == 0) {
      float* const parP = (float*)(lds + kS_ParP);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int f = feat_of(t);
        const float b = a.b_out[f];
        // (the accumulator of tile t starts as b for each of its 4 edges: stored replicated, so that the layer reads it with one
        //  ds_read_b128 straight into the accumulator - a VALU move in front of an asm MFMA needs wait states nobody inserts)
        ((f32x4*)(lds + kS_ParT))[(tw * 16 + j) * 4 + t] = f32x4{b, b, b, b};
        const int pos = 64 * tw + 16 * (j >> 2) + 4 * t + (j & 3);
        parP[pos] = a.gamma[f];
        parP[256 + pos] = a.beta[f];
      }
    }
  } else {
    const char* wsrc = team_b ? a.w_out : a.w_mid;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        wr[t][s] = *(const bf16x8*)(wsrc + ((size_t)(s * 16 + 4 * tw + t) * 64 + lane) * 16);
        if (F16MID && !team_b) {  // (fragment by fragment: converting all 128 registers at once spills)
          typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
          f16x8 h;
#pragma unroll
          for (int i = 0; i < 8; ++i) h[i] = (_Float16)(float)wr[t][s][i];
          // (a value made by vector instructions is born in the VGPR class, and the allocator would then keep the weights there
          //  and copy them in front of every MFMA: re-define it as an AGPR-class value here, once)
          bf16x8 conv = __builtin_bit_cast(bf16x8, h), pinned;
          asm volatile("" : "=a"(pinned) : "0"(conv));
          wr[t][s] = pinned;
          if ((s & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
  }
  if (threadIdx.x < 256) {
    float* par_w = (float*)(lds + kT_Par);
    const int i = threadIdx.x;
    par_w[i] = a.b_mid[i];
    par_w[256 + i] = a.b_out[i];
    par_w[512 + i] = a.gamma[i];
    par_w[768 + i] = a.beta[i];
    par_w[1024 + i] = a.b1[i];
  }
#define par_l ((const float*)(lds + kT_Par) + fresh(f0))  /* this lane's slice: + 256 * which + 16 * t (recomputed per use) */
  char* const h1 = lds + kT_H1;
  char* const h2 = lds + kT_H2;
  float* const stage = (float*)(lds + kT_Stage);
  int* const gdl = (int*)(lds + kT_Gd);
  float* const lnp = (float*)(lds + kT_Ln);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  // ---- tile list of this workgroup: XCD x = workgroup & 7 owns a contiguous range of edge blocks; unit = (edge block, chunk
  // of bc batch elements); the workgroups of an XCD take units round-robin and walk the bc samples of a unit in a row ----
  const int slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const TileWalk twk = tile_walk(blockIdx.x & 7, a.neb, a.nchunk);
  const int n_my_units = twk.n_units > slot ? (twk.n_units - slot + nslot - 1) / nslot : 0;
  const int n = n_my_units * a.bc;
  auto tile_at = [&](int i) -> TileId {
    const int k = i / a.bc, bi = i - k * a.bc;
    const int u = slot + k * nslot;
    const int ebl = u / a.nchunk, c = u - ebl * a.nchunk;
    return TileId{twk.eb_start + ebl, c * a.bc + bi};
  };

  // gw_debug_timestamps(kind 3): phase clocks of each workgroup's 4th pipeline step (32 x u64 per workgroup: team A [0, 16),
  // team B [16, 32)); tuning builds: a.tune bit 0 = no s_setprio around the MFMA phases, bit 1 = no residual loads (zeros)
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
  const bool use_prio = (GW_TUNE_ARG(a) & 1) == 0;
  const bool use_res = (GW_TUNE_ARG(a) & 2) == 0;
  const bool t_skip_ln = (GW_TUNE_ARG(a) & 4) != 0;     // (wrong results: timing experiments) team B skips LayerNorm / staging
  const bool t_skip_a2 = (GW_TUNE_ARG(a) & 8) != 0;     // team A skips its half-2 work (segment sums, gather finish)
  const bool t_skip_bseg = (GW_TUNE_ARG(a) & 16) != 0;  // team B skips its segment sums

  // =========================================== team A: Hbuf1 of a tile ===============================================
  // GATHER: thread -> (column c of 32, 16-byte piece) and two column passes; 8 lanes read one 128-byte line of a projected row.
  const int gpiece = threadIdx.x & 7;
  const int gcol = (threadIdx.x >> 3) & 31;
  int gidx[2][3] = {{0, 0, 0}, {0, 0, 0}};
  // per column pass and K-step: b1 + rows of the batch-shared tables as fp16 pairs, valid for `cached_eb`; pass 0 in registers,
  // pass 1 in a thread-private LDS slot (16 + 16 registers would not fit beside the gathered rows)
  half2_t zc[8][2];
#pragma unroll
  for (int s = 0; s < 8; ++s) zc[s][0] = zc[s][1] = half2_t{(_Float16)0.f, (_Float16)0.f};
  typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
  half4_t* const zc1 = (half4_t*)(lds + kT_Zc) + (threadIdx.x & 255);  // [s][thread]: + 256 s
  // (segment-aligned form: pass 0's part sits in LDS too - 16 registers carried around team A's loop were 9 scratch loads per tile)
  half4_t* const zc0 = (half4_t*)(lds + (SEGT ? kS_Zc0 : kT_Zc)) + (threadIdx.x & 255);
  int cached_eb = -1;

  // Unit u = 0..7 of a (column, piece) thread: 4 features = 8 bytes of one B-fragment slot of Hbuf1.
  //   PH == false (fp32 tables): piece p reads floats 32 u + 4 p .. + 3 of a row (8 lanes = one 128-byte line per unit);
  //                              slot: K-step u, q = p & 3, half p >> 2.
  //   PH == true  (the per-sample table is fp16 rows): piece p reads halves 64 i + 8 p .. + 7 (16 bytes, 8 lanes = one line)
  //                              for i = 0..3; unit u = 2 i + e holds features 64 i + 8 p + 4 e .. + 3:
  //                              K-step 2 i + (p >> 2), q = 2 (p & 1) + e, half (p & 3) >> 1   (k = 32 s + 16 half + 4 q + r).
  // (each offset = a per-thread part, recomputed where it is used, + a compile-time part of the unit that folds into the
  //  instruction's immediate)
  auto poff = [&]() -> int { return fresh(PH ? 8 * gpiece : 4 * gpiece); };                      // floats into a row
  auto uoff = [](int u) -> int { return PH ? 64 * (u >> 1) + 4 * (u & 1) : 32 * u; };
  auto pslot = [&](int col) -> int {                                                              // bytes into Hbuf1
    const int ks = PH ? (gpiece >> 2) : 0;
    const int qq = PH ? 2 * (gpiece & 1) : (gpiece & 3);
    const int hf = PH ? (gpiece & 3) >> 1 : gpiece >> 2;
    return fresh((col >> 4) * 8192 + ks * 1024 + (16 * qq + (col & 15)) * 16 + hf * 8);
  };
  auto uslot = [](int u) -> int { return PH ? (u >> 1) * 2048 + (u & 1) * 256 : u * 1024; };
  auto prep_chunk = [&](int eb) {  // new edge block: row indices + the batch-shared part of layer 1
    const float* b1l = (const float*)(lds + kT_Par) + 1024 + poff();
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      const int kr = eb * kTileCols + 32 * cp + gcol;
      const int k = kr < a.n_edges ? kr : a.n_edges - 1;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        gidx[cp][p] = p < a.n_proj ? (a.p_kind[p] == 0 ? ldgi(a.src + k) : (a.p_kind[p] == 1 ? ldgi(a.dst + k) : k)) : 0;
      if (SEGT) {  // (padding columns carry dst = -1: any valid row will do, their results are never summed)
#pragma unroll
        for (int p = 0; p < 3; ++p) gidx[cp][p] = gidx[cp][p] < 0 ? 0 : gidx[cp][p];
      }
    }
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      f32x4 z[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = *(const f32x4*)(b1l + uoff(u));
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (p < a.n_proj && a.p_rows_pb[p] == 0) {  // (batch-shared tables are fp32 rows: the host checks)
          const float* row = a.p_ptr[p] + (size_t)gidx[cp][p] * (size_t)a.p_ld[p] + poff();
          f32x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = ldg4(row + uoff(u));
#pragma unroll
          for (int u = 0; u < 8; ++u) z[u] += v[u];
        }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (cp == 0 && SEGT) {
          zc0[256 * u] = half4_t{(_Float16)z[u].x, (_Float16)z[u].y, (_Float16)z[u].z, (_Float16)z[u].w};
        } else if (cp == 0) {
          zc[u][0] = half2_t{(_Float16)z[u].x, (_Float16)z[u].y};
          zc[u][1] = half2_t{(_Float16)z[u].z, (_Float16)z[u].w};
        } else {
          zc1[256 * u] = half4_t{(_Float16)z[u].x, (_Float16)z[u].y, (_Float16)z[u].z, (_Float16)z[u].w};
        }
      }
    }
    cached_eb = eb;
  };
  auto gather_store = [&](const f32x4 (&z)[8], int cp, bool cvalid) {
    char* out = h1 + pslot(32 * cp + gcol);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      bf16x4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (__bf16)(cvalid ? fmaxf(z[u][r], 0.f) : 0.f);
      *(bf16x4*)(out + uslot(u)) = o4;
    }
  };
  auto cache_add = [&](f32x4 (&z)[8], int cp) {  // z += the cached batch-shared part of this column pass
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (cp == 0 && !SEGT) {
        z[u] += f32x4{(float)zc[u][0][0], (float)zc[u][0][1], (float)zc[u][1][0], (float)zc[u][1][1]};
      } else {
        const half4_t c = cp == 0 ? zc0[256 * u] : zc1[256 * u];
        z[u] += f32x4{(float)c[0], (float)c[1], (float)c[2], (float)c[3]};
      }
    }
  };
  // The gather of a tile is split so that its load round trips pass under other work.  fp32 table: gather_issue0 requests the
  // rows of column pass 0 at the END of half 1 - in flight across the barrier; part 1 (before the segment sums of half 2)
  // finishes pass 0 and requests pass 1 - a wave's loads queue behind its own earlier stores, so every load of the half is
  // issued before the aggregate stores; part 2 (after them) finishes pass 1 (both passes in flight from half 1 would need 64
  // registers across the barrier: it spills).  fp16 table: a pass is 4 loads of 16 bytes = 16 registers, so BOTH passes are
  // requested at the end of half 1 and half 2 issues no load of its own.
  int dyn = 0;  // the per-sample table of this launch (rows_pb != 0): exactly one (the host checks)
  if (GATHER) {
#pragma unroll
    for (int p = 2; p >= 0; --p)
      if (p < a.n_proj && a.p_rows_pb[p] != 0) dyn = p;
  }
  typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
  auto row_of = [&](TileId t, int cp) -> size_t {  // element offset of the dyn table's row for this thread's column of pass cp
    return ((size_t)t.b * (size_t)a.p_rows_pb[dyn] + (size_t)gidx[cp][dyn]) * (size_t)a.p_ld[dyn];
  };
  auto load_rows = [&](f32x4 (&x)[8], TileId t, int cp) {
    const float* row = a.p_ptr[dyn] + row_of(t, cp) + poff();
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = ldg4(row + uoff(u));
  };
  auto load_rows_h = [&](half8_t (&xh)[4], TileId t, int cp) {
    const _Float16* row = (const _Float16*)a.p_ptr[dyn] + row_of(t, cp) + poff();
#pragma unroll
    for (int i = 0; i < 4; ++i) xh[i] = *(const GW_AS1 half8_t*)(row + 64 * i);
  };
  auto widen = [&](f32x4 (&x)[8], const half8_t (&xh)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = f32x4{(float)xh[i][0], (float)xh[i][1], (float)xh[i][2], (float)xh[i][3]};
      x[2 * i + 1] = f32x4{(float)xh[i][4], (float)xh[i][5], (float)xh[i][6], (float)xh[i][7]};
    }
  };
  // F16MID: one column pass of layer 1 in packed fp16 arithmetic: relu(row + cached part), capped at the largest fp16 (an
  // overflowed sum would be +inf, then NaN under the matrix product), 8 bytes per unit straight into the B-operand slots
  auto gather_finish_h = [&](const half8_t (&xh)[4], int cp) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    char* out = h1 + pslot(32 * cp + gcol);
    const h4_t zero = h4_t{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const h4_t cap = h4_t{(_Float16)65504.f, (_Float16)65504.f, (_Float16)65504.f, (_Float16)65504.f};
    // (all 8 cached parts first: reads interleaved with the Hbuf1 writes below are serialised behind them - the compiler cannot
    //  tell the two LDS regions apart - and cost an LDS round trip per unit)
    h4_t c4[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c4[u] = cp == 0 ? zc0[256 * u] : zc1[256 * u];  // (F16MID implies the segment-aligned form: both in LDS)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = u >> 1, e = u & 1;
      const h4_t x4 = h4_t{xh[i][4 * e], xh[i][4 * e + 1], xh[i][4 * e + 2], xh[i][4 * e + 3]};
      const h4_t hsum = __builtin_elementwise_min(__builtin_elementwise_max(x4 + c4[u], zero), cap);
      *(h4_t*)(out + uslot(u)) = hsum;
    }
  };
  // gx: 32 registers of gathered rows in flight (fp32: pass 0 as 8 x f32x4; fp16: pass 0 in gx[0..3], pass 1 in gx[4..7])
  auto gather_issue0 = [&](TileId t, f32x4 (&gx)[8]) {
    if (t.eb != cached_eb) prep_chunk(t.eb);
    if constexpr (PH) {
      half8_t(&gh)[8] = reinterpret_cast<half8_t(&)[8]>(gx);
      load_rows_h(reinterpret_cast<half8_t(&)[4]>(gh[0]), t, 0);
      load_rows_h(reinterpret_cast<half8_t(&)[4]>(gh[4]), t, 1);
    } else {
      load_rows(gx, t, 0);
    }
  };
  auto gather_part1 = [&](TileId t, f32x4 (&gx)[8]) {
    const bool v0 = t.eb * kTileCols + gcol < a.n_edges;
    if constexpr (F16MID) {
      half8_t(&gh)[8] = reinterpret_cast<half8_t(&)[8]>(gx);
      gather_finish_h(reinterpret_cast<half8_t(&)[4]>(gh[0]), 0);
    } else if constexpr (PH) {
      half8_t(&gh)[8] = reinterpret_cast<half8_t(&)[8]>(gx);
      f32x4 x[8];
      widen(x, reinterpret_cast<half8_t(&)[4]>(gh[0]));
      cache_add(x, 0);
      gather_store(x, 0, v0);
    } else {
      cache_add(gx, 0);
      gather_store(gx, 0, v0);
      load_rows(gx, t, 1);
    }
  };
  auto gather_part2 = [&](TileId t, f32x4 (&gx)[8]) {
    const bool v1 = t.eb * kTileCols + 32 + gcol < a.n_edges;
    if constexpr (F16MID) {
      half8_t(&gh)[8] = reinterpret_cast<half8_t(&)[8]>(gx);
      gather_finish_h(reinterpret_cast<half8_t(&)[4]>(gh[4]), 1);
    } else if constexpr (PH) {
      half8_t(&gh)[8] = reinterpret_cast<half8_t(&)[8]>(gx);
      f32x4 x[8];
      widen(x, reinterpret_cast<half8_t(&)[4]>(gh[4]));
      cache_add(x, 1);
      gather_store(x, 1, v1);
    } else {
      cache_add(gx, 1);
      gather_store(gx, 1, v1);
    }
  };
  auto tile_row = [&](TileId t) -> size_t { return (size_t)(t.b * a.neb + t.eb); };
  auto prep_dma_issue = [&](TileId t) {  // 32 KiB of layer-1 activations -> Hbuf1: 8 LDS-DMA pieces of 1 KiB per team-A wave
    const char* src = a.h1g + tile_row(t) * kHBytes;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int piece = 8 * tw + i;
      glds16_asm_s((const float*)(src + piece * 1024), (unsigned)lane * 16u,
                   __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kT_H1 + piece * 1024)));
    }
  };
  auto publish_dst = [&](TileId t, int ring) {  // (wave 0) destination rows of the tile's 64 columns, for its segment sums
    const int kr = t.eb * kTileCols + lane;
    if constexpr (SEGT) {
      // destination slots of the tile: a column starts a slot where its destination differs from its left neighbour's
      // (runs are contiguous, padding columns sit at the end of the tile); slot of a column = slot starts up to it - 1
      const int d = ldgi(a.dst + kr);
      const int dp = lane > 0 ? ldgi(a.dst + kr - 1) : -2;
      const bool valid = d >= 0;
      const bool start = valid && d != dp;
      const unsigned long long sm = __ballot(start);
      const unsigned long long vm = __ballot(valid);
      const int slot = __popcll(sm & ((2ull << lane) - 1ull)) - 1;
      ((unsigned char*)(lds + kS_Slot))[ring * kTileCols + lane] = (unsigned char)(valid ? slot : 255);
      if (start) {
        const unsigned long long rest = (sm >> lane) >> 1;  // slot starts to the right of this one
        const int len = rest != 0ull ? __builtin_ctzll(rest) + 1 : __popcll(vm >> lane);
        ((int*)(lds + kS_Dsl))[ring * kTileCols + slot] = d;  // (row within a batch element: the table serves the whole chunk)
        ((float*)(lds + kS_Cnt))[ring * kTileCols + slot] = (float)len;
        int part = 0;
        if (a.seg_split) {  // does the run go on before column 0 / behind column 63 of this tile?
          if (lane == 0 && t.eb > 0 && ldgi(a.dst + kr - 1) == d) part = 1;
          if (lane + len == kTileCols && t.eb + 1 < a.neb && ldgi(a.dst + kr + len) == d) part = 1;
        }
        ((int*)(lds + kS_Part))[ring * kTileCols + slot] = part;
      }
      const int nsl_ = __popcll(sm);
      if (lane == 0) ((int*)(lds + kS_Nsl))[ring] = nsl_;
      // S[edge][slot] as the B operand of the segment-sum product, once per edge block for all four team-B waves and all tiles
      // of the unit: lane (n, qq) of slot group nt, half h: 8 K entries = edges 4 qq + r of groups 2 h, 2 h + 1 -> 1.0 (bf16)
      // where the edge's slot is 16 nt + n.  (The slot bytes were written by this wave just above: LDS keeps a wave's order.)
      {
        const unsigned* const sw = (const unsigned*)(lds + kS_Slot) + ring * 16;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4* const sm_out = (u32x4*)(lds + kS_Smat) + ring * (4 * 2 * 64);
        const int n_ = lane & 15, qq = lane >> 4;
        for (int nt = 0; nt * 16 < nsl_; ++nt) {
          const unsigned me = (unsigned)(16 * nt + n_);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned w0 = sw[8 * h + qq], w1 = sw[8 * h + 4 + qq];
            u32x4 pk;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              pk[i] = ((((w0 >> (16 * i)) & 255u) == me) ? 0x3F80u : 0u) | ((((w0 >> (16 * i + 8)) & 255u) == me) ? 0x3F800000u : 0u);
              pk[2 + i] = ((((w1 >> (16 * i)) & 255u) == me) ? 0x3F80u : 0u) | ((((w1 >> (16 * i + 8)) & 255u) == me) ? 0x3F800000u : 0u);
            }
            sm_out[(nt * 2 + h) * 64 + lane] = pk;
          }
        }
      }
    } else {
      gdl[ring * kTileCols + lane] = kr < a.n_edges ? t.b * a.n_dst + ldgi(a.dst + kr) : -1;
    }
  };

  // ===================================== both teams: segment sums of a staged tile ====================================
  // Thread (f, h) owns feature f = thread & 255 for the columns 32 h .. 32 h + 31 (h = team).  Segment ends are the same for
  // every thread: lane i compares column i's destination with column i + 1's, the ballot is a 64-bit scalar mask, the walk
  // tests one bit per column.  A segment wholly inside the thread's columns is complete: plain store.  The first and the last
  // one may continue elsewhere (neighbouring tiles, or across the middle of this one) and are added with atomics.
  // (Measured and kept as is: uneven column splits between the teams - 16 / 48, 24 / 40, 40 / 24 - and 8 or 32 staged values
  // per batch instead of 16 change nothing beyond the box-to-box noise, or lose 12-20 % in the DMA-fed form.)
  auto segment_sums = [&](TileId t, int ring) {
    constexpr int COLS = 32;
    int f = threadIdx.x & 255;
    asm volatile("" : "+v"(f));
    const int hh = team_b ? 1 : 0;
    const int c0 = hh * COLS;
    constexpr int VH = 16;  // staged values read at a time (registers: team A holds gathered rows and its cache across the walk)
    float vv[VH];
#pragma unroll
    for (int i = 0; i < VH; ++i) vv[i] = stage[(c0 + i) * kStageLd + f];
    const int gdv = gdl[ring * kTileCols + lane];
    const int gdn = gdl[ring * kTileCols + (lane < kTileCols - 1 ? lane + 1 : lane)];
    const unsigned long long ends = __ballot(lane == kTileCols - 1 || gdn != gdv);
    const bool mid_open = ((ends >> (COLS - 1)) & 1ull) == 0;  // a segment straddles columns 31 | 32
    const unsigned long long mine = (ends >> c0) | (1ull << (COLS - 1));
    float run = 0.f;
    bool first = true;
#pragma unroll
    for (int i = 0; i < COLS; ++i) {
      if (i > 0 && i % VH == 0) {  // next batch of staged values
#pragma unroll
        for (int i2 = 0; i2 < VH; ++i2) vv[i2] = stage[(c0 + i + i2) * kStageLd + f];
      }
      run += vv[i % VH];
      if (__builtin_expect((mine >> i) & 1ull, 0)) {
        const int cur = __builtin_amdgcn_readlane(gdv, c0 + i);
        if (cur >= 0 && GW_SKIP(a) != 1) {
          float* dstp = a.agg + (size_t)cur * 256 + f;
          const bool open_lo = first && (hh == 0 || mid_open);
          const bool open_hi = i == COLS - 1 && (hh == 1 || mid_open);
          if (open_lo || open_hi) __hip_atomic_fetch_add((GW_AS1 float*)dstp, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else stg1(dstp, run);
        }
        first = false;
        run = 0.f;
      }
    }
    if (a.e_out_tiles != nullptr) {
      // e' as bf16 edge tiles from the staged tile: wave w packs K-step w of all 4 groups - lane (j, q) holds features
      // 32 w + 16 (i >> 2) + 4 q + (i & 3) of column 16 g + j - one coalesced 1 KiB store per wave and group
      const size_t tile = tile_row(t);
      int so = 32 * wave + 4 * q;
      asm volatile("" : "+v"(so));
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        const int col = 16 * g + j;
        const f32x4 lo = *(const f32x4*)(stage + col * kStageLd + so);
        const f32x4 hi = *(const f32x4*)(stage + col * kStageLd + so + 16);
        bf16x8 pk = to_bf16x8(lo, hi);
        if (t.eb * kTileCols + col >= a.n_edges) pk = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        *(GW_AS1 bf16x8*)(a.e_out_tiles + (tile * kGroups + g) * 8192 + (size_t)wave * 1024 + (size_t)(fresh(lane) * 16)) = pk;
      }
    }
    if (a.e_out != nullptr) {  // e' as fp32 rows from the staged tile (callers that want rows back): 8 rows per wave
      int l4 = 4 * lane;
      asm volatile("" : "+v"(l4));
      const int k0 = t.eb * kTileCols;
#pragma unroll 4
      for (int i = 0; i < 8; ++i) {
        const int col = 8 * wave + i;
        if (k0 + col < a.n_edges) {
          const f32x4 v = *(const f32x4*)(stage + col * kStageLd + l4);
          stg4(a.e_out + ((size_t)t.b * a.n_edges + k0 + col) * 256 + l4, v);
        }
      }
    }
  };

  if (n == 0) return;  // (uniform for the workgroup: no barrier is skipped by part of it)

  if (!team_b) {
    // ================================================ team A ========================================================
    TileId t_next = tile_at(0);
    int res_warm = 0;
    if (GATHER) {
      __syncthreads();  // the parameter block (b1) is visible  [team B meets this barrier below]
      f32x4 gx0[8];
      gather_issue0(t_next, gx0);
      gather_part1(t_next, gx0);
      gather_part2(t_next, gx0);
    } else {
      __syncthreads();
      prep_dma_issue(t_next);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (SEGT: the slot tables depend on the edge block only - one table per unit of bc tiles, in ring entry (tile / bc) & 3)
    if (tw == 0) publish_dst(t_next, 0);
    TileId t_cur = t_next, t_prev = t_next;
#pragma unroll 1
    for (int s = 0; s <= n; ++s) {
      t_prev = t_cur;
      t_cur = t_next;
      if (s + 1 < n) t_next = tile_at(s + 1);
      stamp = a.dbg != nullptr && s == 3 && (int)blockIdx.x < a.dbg_cap;
      f32x4 gx[8];  // GATHER: rows of column pass 0 of the next tile, in flight from the end of half 1 into half 2 (declared per
                    // step: as a loop-carried variable it would hold 32 registers through the MFMA phase as well)
      GW_TS(0)
      team_barrier();  // (alpha) Hbuf1 of tile s complete; staged tile s - 2 free
      GW_TS(1)
      const bool has_next = s + 1 < n;
      // F16MID: both column passes of the next tile's rows are 32 registers - requested HERE, in front of the middle layer, they
      // have the whole MFMA phase to arrive (requested behind it, the first pass was still under way 1.5 k cycles into half 2)
      if (F16MID && has_next) gather_issue0(t_next, gx);
      if (s < n) {
        // ---- middle layer of tile s: Hbuf1 -> Hbuf2 ----
        f32x4 acc[2][4];
        unsigned pk[8];  // bf16 pairs of the group being packed: row tile t -> pk[2 t], pk[2 t + 1]
        if (use_prio) __builtin_amdgcn_s_setprio(1);
        // (SEGT: ONE address register for the layer's bias reads - 16-byte reads at immediate offsets instead of a recomputed
        //  address and two 8-byte reads per row tile and group: the layer is bound by what a wave can issue between its MFMAs)
        const float* const pbm = SEGT ? par_l : nullptr;
        team_layer<2, false, F16MID, (SEGT ? 18 : 28)>(  // (SEGT: the pieces read the accumulators in slots 2..17 only)
            acc, wr, h1, lane,
            [&](f32x4& dst, int t) {  // b_mid
              if constexpr (SEGT) dst = *(const f32x4*)__builtin_assume_aligned(pbm + 16 * t, 16);
              else dst = *(const f32x4*)(par_l + 16 * t);
            },
            [&](int g, int m, f32x4 (&ac)[4]) {
              // m = 0..15: relu of one accumulator value; 16..23: bf16 pack of a pair; 24 / 25: the 16-byte store of K-step s0 / s0 + 1
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              if constexpr (SEGT) {
                // relu on the PACKED pairs: a negative bf16 is a negative 16-bit integer, so max(., 0) per half clears it and leaves
                // the others as they are - 8 instead of 16 instructions per group
                if (m < 16) {
                  if ((m & 1) == 0) {
                    const int pi = m >> 1;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[pi]) : "v"(ac[pi >> 1][2 * (pi & 1)]), "v"(ac[pi >> 1][2 * (pi & 1) + 1]));
                  } else {
                    const int pi = m >> 1;
                    asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[pi]) : "v"(pk[pi]));
                  }
                }
              } else if (m < 16) {
                ac[m >> 2][m & 3] = relu1(ac[m >> 2][m & 3]);
              } else if (m < 24) {
                const int pi = m - 16;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[pi]) : "v"(ac[pi >> 1][2 * (pi & 1)]), "v"(ac[pi >> 1][2 * (pi & 1) + 1]));
              }
              if (m == 24) {
                *(u32x4*)(h2 + ((g * 8 + s0) * 64 + fresh(lane)) * 16) = u32x4{pk[0], pk[1], pk[2], pk[3]};
              } else if (m == 25) {
                *(u32x4*)(h2 + ((g * 8 + s0 + 1) * 64 + fresh(lane)) * 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
              }
            });
        __builtin_amdgcn_s_setprio(0);
        GW_TS(2)
      }
      if (GATHER && !F16MID && has_next) gather_issue0(t_next, gx);  // (registers only: Hbuf1 is still being read by the other waves)
      GW_TS(6)
      team_barrier();  // (beta) Hbuf2 of tile s and the staged tile s - 1 complete; Hbuf1 free
      GW_TS(7)
      if (!GATHER && has_next) prep_dma_issue(t_next);  // in flight under the segment sums
      if (RES && s < n && !a.res_tiles_shared) {
        // the residual of tile s (per-sample bf16 edge tiles, 256 lines of 128 bytes) -> this XCD's L2: team B reads it in its
        // next half-step just in time, 8 bytes at a time - a round trip to HBM there would be exposed
        res_warm ^= ldgi((const int*)(a.res_tiles + tile_row(t_cur) * kHBytes + (size_t)(threadIdx.x & 255) * 128));
      }
      if (GATHER && has_next && !t_skip_a2) gather_part1(t_next, gx);
      GW_TS(9)
      if constexpr (!SEGT) {
        if (s >= 1 && !t_skip_a2) segment_sums(t_prev, (s - 1) & 3);
      }
      GW_TS(8)
      if (has_next) {
        if (GATHER && !t_skip_a2) gather_part2(t_next, gx);
        GW_TS(10)
        if constexpr (SEGT) {
          if (tw == 0 && (s + 1) % a.bc == 0) publish_dst(t_next, ((s + 1) / a.bc) & 3);
        } else {
          if (tw == 0) publish_dst(t_next, (s + 1) & 3);
        }
        if (!GATHER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      GW_TS(11)
    }
    asm volatile("" ::"v"(res_warm));
  } else {
    // ================================================ team B ========================================================
    __syncthreads();  // (pairs with team A's barrier before its first gather / DMA)
    if constexpr (SEGT) {
      // ---- segment-aligned form: transposed output layer, LayerNorm across lanes, segment sums on the matrix cores ----
      f32x4 o[kGroups][4];  // o[g][t][r]: feature feat_of(t) (this lane's column of output tile t), edge 16 g + 4 q + r
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) o[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      float* const ln1 = lnp;                     // [wave][edge]: sum over the wave's 64 features
      float* const ln2 = lnp + 4 * kTileCols;     // [wave][edge]: sum of squares
      float* const lnc = (float*)(lds + kS_Lnc);  // [wave][edge] (rstd, -mean rstd)
      const f32x4* const parT = (const f32x4*)(lds + kS_ParT);   // b_out of the lane's features
      const float* const parP = (const float*)(lds + kS_ParP);    // gamma, beta in position order
      const float* const cntf = (const float*)(lds + kS_Cnt);
      typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
      const u32x4s* const smat = (const u32x4s*)(lds + kS_Smat);
      const int* const dsl = (const int*)(lds + kS_Dsl);
      const int* const nsl = (const int*)(lds + kS_Nsl);
      TileId t_next = tile_at(0), t_cur = t_next, t_prev = t_next;
#pragma unroll 1
      for (int s = 0; s <= n; ++s) {
        t_prev = t_cur;
        t_cur = t_next;
        if (s + 1 < n) t_next = tile_at(s + 1);
        stamp = a.dbg != nullptr && s == 3 && (int)blockIdx.x < a.dbg_cap;
        GW_TS(0)
        team_barrier();  // (alpha) LayerNorm partial sums of tile s - 1 visible
        GW_TS(1)
        if (s >= 1 && !t_skip_ln) {
          const int ring = ((s - 1) / a.bc) & 3;  // slot tables of the unit (edge block) tile s - 1 belongs to
          // LayerNorm statistics: lane (j, q) combines the four waves' partial sums of ONE edge, 16 (j >> 2) + 4 q + (j & 3),
          // and parks (rstd, -mean rstd) in the wave's own slot; each lane then reads those of its 16 edges back (a wave's
          // LDS accesses complete in order: no barrier)
          {
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
          // n = (o - mean) rstd, rounded to bf16 in the A-operand form of the segment-sum product: ypk[t][h] = the lane's feature of
          // output tile t for the 8 edges 4 q + r of groups 2 h, 2 h + 1 (group by group: the accumulators die here).  gamma and
          // beta wait until after the sums - sum_k (n_k gamma + beta) = gamma sum_k n_k + count beta - where they cost one fma per
          // SUM instead of one per value and no registers in this phase.
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 ypk[4][2];
          // ((rstd, -mean rstd) of the next group requested before this group's arithmetic: one LDS latency, not four)
          f32x4 gnx0 = *(const f32x4*)(lnc + (tw * kTileCols + 4 * fresh(q)) * 2);
          f32x4 gnx1 = *(const f32x4*)(lnc + (tw * kTileCols + 4 * fresh(q) + 2) * 2);
#pragma unroll
          for (int g = 0; g < kGroups; ++g) {
            const f32x4 gab0 = gnx0;  // (rstd, -mean rstd) x edges 4 q + 0, 1
            const f32x4 gab1 = gnx1;  // ... 4 q + 2, 3
            if (g + 1 < kGroups) {
              gnx0 = *(const f32x4*)(lnc + (tw * kTileCols + 16 * (g + 1) + 4 * fresh(q)) * 2);
              gnx1 = *(const f32x4*)(lnc + (tw * kTileCols + 16 * (g + 1) + 4 * fresh(q) + 2) * 2);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float n0 = fmaf(o[g][t][0], gab0[0], gab0[1]), n1 = fmaf(o[g][t][1], gab0[2], gab0[3]);
              const float n2 = fmaf(o[g][t][2], gab1[0], gab1[1]), n3 = fmaf(o[g][t][3], gab1[2], gab1[3]);
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ypk[t][g >> 1][2 * (g & 1)]) : "v"(n0), "v"(n1));
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ypk[t][g >> 1][2 * (g & 1) + 1]) : "v"(n2), "v"(n3));
            }
            __builtin_amdgcn_sched_barrier(0);  // (group by group: the next group's statistics would only hold registers)
          }
          GW_TS(2)
          const int nslots = __builtin_amdgcn_readfirstlane(nsl[ring]);
#pragma unroll 1
          for (int nt = 0; nt * 16 < nslots; ++nt) {
            // S[edge][slot] of slot group nt (made once per edge block by the publishing wave of team A)
            u32x4 sb[2];
            sb[0] = smat[((ring * 4 + nt) * 2 + 0) * 64 + fresh(lane)];
            sb[1] = smat[((ring * 4 + nt) * 2 + 1) * 64 + fresh(lane)];
            // (asm MFMAs on plain vector registers: the builtin lets the allocator put these accumulators into the AGPR half
            //  and rotate the resident weights out of their way; wait states as in layer_group - inline asm is opaque to the
            //  hazard recogniser; an accumulator is touched by every 4th MFMA)
            // (the destination row, the slot's edge count and gamma / beta of the lane's 16 positions are requested in front of the
            //  product: their LDS round trips run under its MFMAs instead of behind them)
            const int slot = 16 * nt + j;
            const bool mine = slot < nslots && GW_SKIP(a) != 1;
            const int p0 = fresh(64 * tw + 16 * q);
            const int drow = dsl[ring * kTileCols + (mine ? slot : 0)];
            const float cnt = cntf[ring * kTileCols + (mine ? slot : 0)];
            f32x4 gm[4], bt[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              gm[t] = *(const f32x4*)(parP + p0 + 4 * t);
              bt[t] = *(const f32x4*)(parP + 256 + p0 + 4 * t);
            }
            __builtin_amdgcn_sched_barrier(0);
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
              // agg[row][position p0 + 4 t + r] = gamma sum + count beta (parameters in position order: one float4 per t)
              const size_t row = (size_t)(t_prev.b * a.n_dst + drow);
#pragma unroll
              for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsum[t][r] = fmaf(dsum[t][r], gm[t][r], cnt * bt[t][r]);
              if (a.agg_bf16k) {
                __bf16* dstp = (__bf16*)a.agg + row * 256 + p0;
                *(GW_AS1 bf16x8*)dstp = to_bf16x8(dsum[0], dsum[1]);
                *(GW_AS1 bf16x8*)(dstp + 8) = to_bf16x8(dsum[2], dsum[3]);
              } else if (a.seg_split && ((const int*)(lds + kS_Part))[ring * kTileCols + slot] != 0) {
                float* dstp = a.agg + row * 256 + p0;  // a piece of a run that spans tiles: partial sums meet in atomics
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    __hip_atomic_fetch_add((GW_AS1 float*)(dstp + 4 * t + r), dsum[t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              } else {
                float* dstp = a.agg + row * 256 + p0;
#pragma unroll
                for (int t = 0; t < 4; ++t) stg4(dstp + 4 * t, dsum[t]);
              }
            }
          }
          GW_TS(3)
        }
        team_barrier();  // (beta) Hbuf2 of tile s complete
        GW_TS(7)
        if (s < n) {
          // ---- output layer of tile s, transposed: Hbuf2 -> registers; per group (in the shadow of the next group's MFMAs) the
          // sums and sums of squares over this wave's 64 features: 4 row tiles in registers, then the 16 feature lanes of the row ----
          float s1[4], s2[4], ra[4], rb[2], rc[2], rd[2];  // (running sums of 4 edges; registers of their 16-lane reduce-scatter)
          if (use_prio) __builtin_amdgcn_s_setprio(1);
          team_layer<4, true, false, 4, true>(  // (four accumulator sets: the next group's is free - its bias is read 28 MFMAs ahead)
              o, wr, h2, lane,
              [&](f32x4& dst, int t) { dst = parT[(tw * 16 + fresh(j)) * 4 + t]; },  // b_out of the lane's feature of tile t x 4 edges
              [&](int g, int mm, f32x4 (&ac)[4]) {
                if (mm < 16) {  // one accumulator value into its edge's sums
                  const int t = mm >> 2, r = mm & 3;
                  const float x = ac[t][r];
                  s1[r] = t == 0 ? x : s1[r] + x;
                  s2[r] = t == 0 ? x * x : fmaf(x, x, s2[r]);
                } else {  // 12 slots: 4 rotate-and-add steps for each of the 8 sums, then the partial sums of the 4 edges -> LDS
                  reduce_ops(g, mm, s1, s2, ra, rb, rc, rd);
                  if (mm == 27 && (j & 3) == 0) {  // bank b = j >> 2 holds the sums (2 (b & 1), + 1) of s1 (b < 2) / s2: edges 4 q + ...
                    *(float2*)(ln1 + (fresh(j) >> 3) * (4 * kTileCols) + tw * kTileCols + 16 * g + 4 * fresh(q) + 2 * ((fresh(j) >> 2) & 1)) = float2{rd[0], rd[1]};
                  }
                }
              });
          __builtin_amdgcn_s_setprio(0);
          GW_TS(9)
        } else {
          // (last iteration: no output layer.  The compiler cannot see that this path leaves the loop and would keep the 64
          //  accumulators alive from their last use - the top of half 1 - across the segment sums, rotating resident weights
          //  through scratch to make room: redefine them from nothing on this path, as the layer does on the other)
#pragma unroll
          for (int g = 0; g < kGroups; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "=v"(o[g][t]));
        }
        GW_TS(13)
      }
      return;
    }
    f32x4 o[kGroups][4];
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) o[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    TileId t_next = tile_at(0), t_cur = t_next, t_prev = t_next;
#pragma unroll 1
    for (int s = 0; s <= n; ++s) {
      t_prev = t_cur;
      t_cur = t_next;
      if (s + 1 < n) t_next = tile_at(s + 1);
      stamp = a.dbg != nullptr && s == 3 && (int)blockIdx.x < a.dbg_cap;
      GW_TS(0)
      team_barrier();  // (alpha) LayerNorm partial sums of tile s - 1 visible; staged tile s - 2 free
      GW_TS(1)
      if (s >= 1 && !t_skip_ln) {
        // ---- LayerNorm (eps 1e-5, biased variance), residual, staging of tile s - 1 ----
        // v = (o - mean) rstd gamma + beta + e = o (rstd gamma) + ((-mean rstd) gamma + (beta + e)), one feature tile (16 rows)
        // at a time for all 4 groups: the accumulators of a tile die as it is staged, and the second half of the residual is
        // requested after the first tile - its registers take the place of the dead accumulators.
        // The residual (bf16 edge tiles: feature tile t of this wave = 8 bytes of the lane's 16-byte slot of K-step
        // s0 + (t >> 1)) is read just in time, two feature tiles ahead: team A warmed its lines into L2 half a step ago, this
        // half-step issues no stores, and the stores of the previous one have had its output layer to drain.
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        bf16x4_t rb[2][kGroups];
        const char* rbase = nullptr;
        if constexpr (RES) {
          const size_t rtile = a.res_tiles_shared ? (size_t)t_prev.eb : tile_row(t_prev);
          rbase = a.res_tiles + rtile * kHBytes + (size_t)s0 * 1024 + (size_t)(fresh(lane) * 16);
        }
        auto res_load = [&](bf16x4_t (&r)[kGroups], int t) {
#pragma unroll
          for (int g = 0; g < kGroups; ++g)
            r[g] = use_res ? *(const GW_AS1 bf16x4_t*)(rbase + (size_t)g * 8192 + (t >> 1) * 1024 + (t & 1) * 8) : bf16x4_t{0, 0, 0, 0};
        };
        if constexpr (RES) res_load(rb[0], 0);
        float ga[kGroups], gb[kGroups];
        const float* const lnr = lnp + fresh(q * kTileCols + j) * 2;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
          const float2 pr = *(const float2*)(lnr + 32 * g);  // row q reads team-B wave q's partial sums
          const float s1 = sum_rows(pr.x);
          const float s2 = sum_rows(pr.y);
          const float mean = s1 * (1.0f / 256.0f);
          const float var = fmaxf(s2 * (1.0f / 256.0f) - mean * mean, 0.f);
          ga[g] = __builtin_amdgcn_rsqf(var + 1e-5f);  // v_rsq_f32 (1 ulp): in front of bf16 matrix products
          gb[g] = -mean * ga[g];
        }
        GW_TS(2)
        float* const srow = stage + fresh(j) * kStageLd + fresh(f0);  // + 16 g rows, + 16 t floats: immediates
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          __builtin_amdgcn_sched_barrier(0);  // (one feature tile at a time: hoisted parameter reads would hold registers)
          const f32x4 gm = *(const f32x4*)(par_l + 512 + 16 * t);
          const f32x4 bt = *(const f32x4*)(par_l + 768 + 16 * t);
#pragma unroll
          for (int g = 0; g < kGroups; ++g) {
            f32x4 rv = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (RES) {
              const bf16x4_t rr = rb[t & 1][g];
              rv = f32x4{(float)rr[0], (float)rr[1], (float)rr[2], (float)rr[3]};
            }
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(o[g][t][r], ga[g] * gm[r], fmaf(gb[g], gm[r], bt[r] + rv[r]));
            *(f32x4*)(srow + 16 * g * kStageLd + 16 * t) = v;
            if constexpr (RES) {
              // the next feature tile's residual: requested half a tile ahead (two groups of accumulators have died by now)
              if (g == 1 && t + 1 < 4) {
                __builtin_amdgcn_sched_barrier(0);
                res_load(rb[(t + 1) & 1], t + 1);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          GW_TS(3 + t)
        }
      }
      team_barrier();  // (beta) Hbuf2 of tile s and the staged tile s - 1 complete
      GW_TS(7)
      if (s >= 1 && !t_skip_bseg) segment_sums(t_prev, (s - 1) & 3);
      GW_TS(8)
      if (s < n) {
        // ---- output layer of tile s: Hbuf2 -> registers, LayerNorm partial sums (per group, in the shadow of the next group) ----
        float s1 = 0.f, s2 = 0.f;
        if (use_prio) __builtin_amdgcn_s_setprio(1);
        team_layer<4>(
            o, wr, h2, lane,
            [&](f32x4& dst, int t) { dst = *(const f32x4*)(par_l + 256 + 16 * t); },  // b_out
            [&](int g, int m, f32x4 (&ac)[4]) {
              // m = 0..15: one accumulator value into the column's sum and sum of squares; 16..19: the sums over the four
              // 16-lane rows (q) in two lane-swap steps each; 20: the wave's partial sums of the group's 16 columns -> LDS
              if (m < 16) {
                const float x = ac[m >> 2][m & 3];
                s1 = m == 0 ? x : s1 + x;
                s2 = m == 0 ? x * x : fmaf(x, x, s2);
              } else if (m == 16 || m == 18) {
                float& v = m == 16 ? s1 : s2;
                const unsigned u = __float_as_uint(v);
                const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
              } else if (m == 17 || m == 19) {
                float& v = m == 17 ? s1 : s2;
                const unsigned u = __float_as_uint(v);
                const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
              } else if (m == 20) {
                if (q == 0) *(float2*)(lnp + (tw * kTileCols + 16 * g + fresh(j)) * 2) = float2{s1, s2};
              }
            });
        __builtin_amdgcn_s_setprio(0);
        GW_TS(9)
      }
      GW_TS(13)
    }
  }
}

template <typename K>
int launch_team(K kernel, int n_wg, const Edge16Args& a, void* stream) {
  static DeviceOnce once;  // per template instantiation and device
  if (once.first()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kT_Total);
  hipLaunchKernelGGL(kernel, dim3((unsigned)n_wg), dim3(512), kT_Total, (hipStream_t)stream, a);
  return check_launch("edge16t_kernel launch");
}

}  // namespace

namespace gw {

int edge16t_launch(const void* edge16_args, bool gather, int n_wg, void* stream) {
  const Edge16Args& a = *(const Edge16Args*)edge16_args;
  // Three forms exist (the others measured slower than, or did not fit the registers beside, the lock-step kernel's):
  //   layer-1 tiles by DMA + residual tiles     (processor blocks 1..: per-sample edge features)
  //   layer 1 gathered, no residual, the per-sample table as fp32 or fp16 rows   (decoder)
  const bool res = a.res_tiles != nullptr;
  if (a.seg_tiles && (!gather || res))
    return set_error(GW_E_UNSUPPORTED, "edge16t: segment-aligned tiles come with a gathered layer 1 and no residual");
  if (!gather) {
    if (!res) return set_error(GW_E_UNSUPPORTED, "edge16t: the form with a raw edge operand adds its residual from bf16 edge tiles");
    return launch_team(edge16t_kernel<false, false, true>, n_wg, a, stream);
  }
  if (res) return set_error(GW_E_UNSUPPORTED, "edge16t: the gather form runs without residual");
  int n_dyn = 0, dyn = 0;
  for (int p = a.n_proj - 1; p >= 0; --p)
    if (a.p_rows_pb[p] != 0) {
      ++n_dyn;
      dyn = p;
    }
  for (int p = 0; p < a.n_proj; ++p)
    if (a.p_half[p] && (p != dyn || n_dyn != 1)) return set_error(GW_E_UNSUPPORTED, "edge16t: only the per-sample projected table may be fp16 rows");
  if (a.seg_tiles) {
    if (n_dyn != 1) return set_error(GW_E_UNSUPPORTED, "edge16t: segment-aligned tiles take exactly one per-sample projected table");
    return a.p_half[dyn] ? launch_team(edge16t_kernel<true, true, false, true>, n_wg, a, stream)
                         : launch_team(edge16t_kernel<true, false, false, true>, n_wg, a, stream);
  }
  if (n_dyn == 1)
    return a.p_half[dyn] ? launch_team(edge16t_kernel<true, true, false>, n_wg, a, stream) : launch_team(edge16t_kernel<true, false, false>, n_wg, a, stream);
  // (two per-sample tables - the first processor block - stay on the lock-step kernel: measured 0.43 ms there against 0.49 ms
  //  here, its gather needs both tables of a column pass in flight and does not fit the team's register budget)
  return set_error(GW_E_UNSUPPORTED, "edge16t: the gather form takes exactly one per-sample projected table");
}

}  // namespace gw
