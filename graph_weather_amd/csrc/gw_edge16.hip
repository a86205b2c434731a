// gw_edge16.hip - bf16-MFMA edge update with the weights held on chip (BASELINE.json configs[2]).
//
//   e'[c] = LayerNorm(W_out . relu(W_mid . relu(z1[c]) + b_mid) + b_out) + e[c],   agg[dst(c)] += e'[c]
//   z1[c] = b1 + sum_p P_p[row_p(c)]  (+ W_e . e[c])           (layer 1 split: projected operands are gathered)
//
// The first bf16 version (gw_bf16.hip) streams the packed weights through LDS for every 128-column tile like the fp32
// kernels do.  At bf16 MFMA rates (16x fp32) that stream - 128 KiB per layer and tile - and its barriers cost ten times
// the matrix time (measured: 10 % MFMA utilisation).  Here the weights never move: workgroups are persistent (one per
// CU) and the ACTIVATIONS travel, 16-column groups of bf16 in the MFMA B-operand layout
//       H[group][K-step s][lane][8 x bf16],   k(s, q, i) = 32 s + 16 (i >> 2) + 4 q + (i & 3)
// (the K order of gw_pack_linear_bf16: a wave that owns output row tile T of the producing layer writes the 8-byte half
//  s = T >> 1, half = T & 1 of its own lane - no shuffles).
//
// A tile is 64 consecutive destination-sorted edges of ONE batch element = 4 groups = 32 KiB in that layout.  The same layout
// is the HBM format of the per-sample edge features between processor blocks ("edge tiles", GW_LAYOUT_EDGE_TILES_BF16):
// block n writes e' as tiles, block n+1 reads them as the B operand of its W_e pass and as its residual - half the bytes
// of fp32 rows, every access a fully coalesced 1 KiB per wave instruction, no conversion anywhere.
//
// Launches of one edge update:
//  A. every operand projected (decoder, first processor block): ONE launch, edge16_kernel<NW, RES, GATHER = true> - layer 1
//     is a pure gather-add and is done at the top of each tile inside the persistent kernel (see GATHER below).
//  B. the per-sample edge features are a raw operand (processor blocks 1..): two launches -
//     edge16_l1_kernel: W_e lives in LDS (128 KiB, loaded once per persistent workgroup), every wave works on its own
//       16-column groups with no inter-wave synchronisation: b1 + P_s[src] + P_d[dst] gathered into the accumulators, 128 MFMAs
//       against the edge tile, relu, bf16, one 16-byte store per K-step into a workspace of layer-1 tiles;
//     edge16_kernel<NW, RES, GATHER = false>: those tiles arrive by LDS-DMA, prefetched one tile ahead.
//     (edge16_gather_kernel + GATHER = false is the round-1 two-launch form of case A, kept for A/B runs in tuning builds.)
//  edge16_kernel (persistent, W_mid and W_out in registers: wave w keeps rows 256/NW * w .. of both matrices as MFMA A
//     fragments in AGPRs for the whole kernel) - per tile: layer-1 activations in Hbuf1 | barrier | middle layer -> Hbuf2 |
//     barrier | output layer, LayerNorm partial sums through LDS | barrier | LayerNorm, residual, [e' tile store], staging |
//     barrier | per-feature segment sums (plain stores for segments inside the tile, atomics - or carry records in the
//     deterministic mode - for the two that may continue in a neighbour) [| e' rows from the staged tile].
//     NW = 8 (512 threads, two waves per SIMD, 128 weight registers each) is the default; NW = 4 is the deterministic form
//     (one thread walks a whole tile) and the A/B form of tuning builds.
// Tiles are walked batch-innermost and XCD-aware: the workgroups of an XCD work on the same edge blocks of all batch
// elements at a time, so rows of batch-shared tables are fetched from HBM once and hit in that XCD's L2 afterwards.
// Everything outside the matrix products is fp32, as in gw_bf16.hip.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_edge16.hpp"

using namespace gw;
using namespace gw16;

namespace {


// Launch 1a: one workgroup per EDGE BLOCK (wave = 16-column group), all batch elements in turn.  Eight lanes read one
// 128-byte line of a row (features 32 s .. 32 s + 31), so an instruction touches 8 full cache lines; lane piece p = lane & 7
// holds features 32 s + 4 p .. + 3, which is half (p >> 2) of the B fragment of lane (j, q = p & 3): written there directly as
// 8 bytes of bf16.  Indices, the bias and the rows of batch-shared tables (the cached per-edge products of the encoder /
// decoder / first processor block) are fetched once per edge block and reused for every batch element.
__global__ __launch_bounds__(256) void edge16_gather_kernel(const Edge16Args a) {
  const int lane = threadIdx.x & 63;
  const int g = threadIdx.x >> 6;
  const int piece = lane & 7;
  const int eb = blockIdx.x;
  const int q = piece & 3, half = piece >> 2;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {  // columns j = 8 h + (lane >> 3)
    const int j = 8 * h + (lane >> 3);
    const int kr = eb * kTileCols + 16 * g + j;
    const bool valid = kr < a.n_edges;
    const int k = valid ? kr : a.n_edges - 1;
    int ridx[3] = {0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (p < a.n_proj) ridx[p] = a.p_kind[p] == 0 ? ldgi(a.src + k) : (a.p_kind[p] == 1 ? ldgi(a.dst + k) : k);
    // batch-independent part: bias + rows of the tables shared by the batch
    f32x4 zs[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) zs[s] = ldg4(a.b1 + 32 * s + 4 * piece);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (p < a.n_proj && a.p_rows_pb[p] == 0) {
        const float* row = a.p_ptr[p] + (size_t)ridx[p] * (size_t)a.p_ld[p] + 4 * piece;
        f32x4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = ldg4(row + 32 * s);
#pragma unroll
        for (int s = 0; s < 8; ++s) zs[s] += v[s];
      }
#pragma unroll 1
    for (int b = 0; b < a.batch; ++b) {
      f32x4 z[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) z[s] = zs[s];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (p < a.n_proj && a.p_rows_pb[p] != 0) {
          // all row pieces are requested before the first one is used: one round trip per wave, not one per K-step
          const float* row = a.p_ptr[p] + ((size_t)b * (size_t)a.p_rows_pb[p] + (size_t)ridx[p]) * (size_t)a.p_ld[p] + 4 * piece;
          f32x4 v[8];
#pragma unroll
          for (int s = 0; s < 8; ++s) v[s] = ldg4(row + 32 * s);
#pragma unroll
          for (int s = 0; s < 8; ++s) z[s] += v[s];
        }
      char* out = a.h1g + ((size_t)(b * a.neb + eb) * kGroups + g) * 8192 + (16 * q + j) * 16 + half * 8;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (__bf16)(valid ? fmaxf(z[s][r], 0.f) : 0.f);
        *(bf16x4*)(out + s * 1024) = v;
      }
    }
  }
}

// Launch 1b: layer 1 with a raw edge operand.  Persistent workgroups of 8 waves; W_e (packed bf16 stream, 128 KiB:
// [K-step s][row tile t][lane][8]) is copied into LDS once, then every wave takes 16-column groups on its own:
//   acc[t] = b1 + P_s[src] + P_d[dst]   (accumulator layout: lane (j, q) holds features 16 t + 4 q .. + 3 of column j)
//   acc[t] += W_e[tile t][s] . E[s]      (E = the group's 8 KiB of the edge tile, one coalesced 16-byte load per K-step)
//   H1[s] = bf16(relu(acc[2 s]), relu(acc[2 s + 1]))   (one coalesced 16-byte store per K-step)
// No inter-wave synchronisation after the weight copy: the gathers of one wave overlap the MFMAs of its neighbours.
// 12 waves = 3 per SIMD (168 registers each after the accumulators were halved: no spill): the kernel is bound by the latency of
// its gathers and tile loads, and a third wave per SIMD hides more of it (8 waves: 0.215 ms per block, 12: 0.194, 16 without the
// fragment prefetch - 128 registers - 0.203; A/B on one box, 1 degree, batch 16)
constexpr int kL1Waves = 12;
constexpr bool kL1Prefetch = kL1Waves <= 12;  // next group's e fragments in flight under this group's MFMAs (32 registers)
constexpr int kL1Lds = 128 * 1024 + 1024;  // W_e + b1
// Round 4: a group is worked off in two halves of 8 output row tiles (32 accumulator registers instead of 64), which leaves room
// to request the NEXT group's edge-tile fragments (8 KiB per group: the only bytes of this kernel that come from HBM) and its row
// indices while the current group is on the matrix cores - the dependent chain index -> gather -> MFMA -> store of a group no
// longer starts with an HBM round trip.  b1 lives in LDS beside W_e (it was 16 global loads per group).
__global__ __launch_bounds__(64 * kL1Waves, 2) void edge16_l1_kernel(const Edge16Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  {
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    for (int p = wave; p < 128; p += kL1Waves)
      glds16_asm_s((const float*)(a.w_raw + (size_t)p * 1024), (unsigned)lane * 16u,
                   __builtin_amdgcn_readfirstlane(lds0 + (unsigned)p * 1024u));
    if (threadIdx.x < 256) ((float*)(lds + 128 * 1024))[threadIdx.x] = a.b1[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // XCD-local walk: workgroup i runs on XCD i % 8 (each XCD has its own L2); XCD x takes the contiguous range of groups
  // [x n / 8, (x + 1) n / 8) - whole batch elements, walked in destination order - so the P_d rows of consecutive
  // destination-sorted edges and the P_s rows of their mesh neighbourhoods stay in that XCD's L2 instead of every L2 seeing
  // the whole node tables.
  const int n_groups = a.batch * a.neb * kGroups;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int g_lo = (int)((long long)n_groups * xcd / 8), g_hi = (int)((long long)n_groups * (xcd + 1) / 8);
  const int stride = nslot * kL1Waves;
  const float* const b1l = (const float*)(lds + 128 * 1024) + 4 * q;
  const char* const wl = lds + lane * 16;

  struct Grp {
    size_t goff, eoff;
    int b, k;
    bool valid;
  };
  auto locate = [&](int u) -> Grp {
    const int tile = u >> 2, g = u & 3;
    const int b = tile / a.neb;
    const int eb = tile - b * a.neb;
    const int kr = eb * kTileCols + 16 * g + j;
    Grp r;
    r.valid = kr < a.n_edges;
    r.k = r.valid ? kr : a.n_edges - 1;
    r.b = b;
    r.goff = ((size_t)tile * kGroups + g) * 8192 + (size_t)lane * 16;
    r.eoff = a.e_tiles_shared ? ((size_t)eb * kGroups + g) * 8192 + (size_t)lane * 16 : r.goff;
    return r;
  };
  int u = g_lo + slot * kL1Waves + wave;
  if (u >= g_hi) return;
  Grp cur = locate(u);
  bf16x8 bf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bf[s] = *(const GW_AS1 bf16x8*)(a.e_tiles + cur.eoff + s * 1024);
  int ridx[2] = {0, 0};
#pragma unroll
  for (int p = 0; p < 2; ++p)
    if (p < a.n_proj) ridx[p] = a.p_kind[p] == 0 ? ldgi(a.src + cur.k) : (a.p_kind[p] == 1 ? ldgi(a.dst + cur.k) : cur.k);
#pragma unroll 1
  for (;;) {
    const int un = u + stride;
    const bool more = un < g_hi;
    Grp nxt = cur;
    bf16x8 bfn[8];
    int ridn[2] = {0, 0};
    if (more) {  // the next group's fragments and row indices: in flight under this group's 128 MFMAs
      nxt = locate(un);
      if constexpr (kL1Prefetch) {
#pragma unroll
        for (int s = 0; s < 8; ++s) bfn[s] = *(const GW_AS1 bf16x8*)(a.e_tiles + nxt.eoff + s * 1024);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (p < a.n_proj) ridn[p] = a.p_kind[p] == 0 ? ldgi(a.src + nxt.k) : (a.p_kind[p] == 1 ? ldgi(a.dst + nxt.k) : nxt.k);
    }
    char* out = a.h1g + cur.goff;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // acc[t] = b1 + P_s[src] + P_d[dst] for the row tiles 8 half + t   (lane (j, q): features 16 T + 4 q .. + 3 of column j)
      f32x4 acc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = *(const f32x4*)(b1l + 16 * (8 * half + t));
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (p < a.n_proj) {
          int r = ridx[p];
          r = r < 0 ? 0 : r;  // (segment-aligned tiles: padding columns carry dst = -1; their results are never summed)
          const size_t ro = ((size_t)cur.b * (size_t)a.p_rows_pb[p] + (size_t)r) * (size_t)a.p_ld[p] + 4 * q + 128 * half;
          if (a.p_half[p]) {  // node products as fp16 rows (GW_LAYOUT_ROWS_F16): 8 bytes per row tile and lane
            typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
            const _Float16* row = (const _Float16*)a.p_ptr[p] + ro;
            half4_t h[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) h[t] = *(const GW_AS1 half4_t*)(row + 16 * t);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += f32x4{(float)h[t][0], (float)h[t][1], (float)h[t][2], (float)h[t][3]};
          } else {
            const float* row = a.p_ptr[p] + ro;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += ldg4(row + 16 * t);
          }
        }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int t4 = 0; t4 < 2; ++t4) {  // A fragments four at a time (16 registers)
          bf16x8 af[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) af[t] = *(const bf16x8*)(wl + (s * 16 + 8 * half + 4 * t4 + t) * 1024);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[4 * t4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bf[s], acc[4 * t4 + t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // output K-steps 4 half + s <- row tiles 2 s, 2 s + 1 of this half
        bf16x8 v = to_bf16x8(relu4(acc[2 * s]), relu4(acc[2 * s + 1]));
        if (!cur.valid) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        *(GW_AS1 bf16x8*)(out + (4 * half + s) * 1024) = v;
      }
    }
    if (!more) break;
    u = un;
    cur = nxt;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if constexpr (kL1Prefetch) bf[s] = bfn[s];
      else bf[s] = *(const GW_AS1 bf16x8*)(a.e_tiles + cur.eoff + s * 1024);  // (more waves per SIMD cover the round trip instead)
    }
    ridx[0] = ridn[0];
    ridx[1] = ridn[1];
  }
}

// fp32 edge rows -> bf16 edge tiles (entry of the tile format: per-sample edge features handed over as rows by a caller)
__global__ __launch_bounds__(256) void rows_to_tiles_kernel(int batch, int n_edges, int neb, const float* __restrict__ rows,
                                                            int rows_pb, int ld, char* __restrict__ tiles) {
  const int lane = threadIdx.x & 63;
  const int g = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x;
  const int b = tile / neb, eb = tile - b * neb;
  const int kr = eb * kTileCols + 16 * g + j;
  const bool valid = kr < n_edges;
  const float* row = rows + ((size_t)b * (size_t)rows_pb + (size_t)(valid ? kr : 0)) * (size_t)ld + 4 * q;
  char* out = tiles + ((size_t)tile * kGroups + g) * 8192 + (size_t)lane * 16;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) v = to_bf16x8(ldg4(row + 32 * s), ldg4(row + 32 * s + 16));
    *(GW_AS1 bf16x8*)(out + s * 1024) = v;
  }
}

// RES_TILES: the residual e comes from bf16 edge tiles (a.res_tiles) instead of fp32 rows (a.res_ptr) - a template parameter
// so that only one set of prefetch registers exists.
// GATHER: layer 1 is a pure gather-add (every operand projected: encoder-free forecaster decoder, first processor block) and
// is done HERE, at the top of each tile - relu(b1 + sum of projected rows) -> bf16 -> Hbuf1 - instead of by a separate launch
// through a workspace in HBM: at batch 16 that workspace round trip (3.7 GB written and 3.7 GB read for the 1 degree decoder)
// was two thirds of the decoder's HBM traffic (profiles/r02_pmc_c3_edge16_v1.json).
template <int NW, bool RES_TILES, bool GATHER>
__global__ __launch_bounds__(64 * NW, NW / 4) void edge16_kernel(const Edge16Args a) {
  constexpr int RT = 16 / NW;         // output row tiles (16 features) per wave
  constexpr int PPW = 32 / NW;        // LDS-DMA pieces (1 KiB) of a 32 KiB tile per wave
  constexpr int KS = RT / 2;          // K-steps of the next layer this wave's outputs fill (2 for RT 4, 1 for RT 2)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  const int f0 = 16 * RT * wave + 4 * q;  // this lane's features: f0 + 16 t + r
  const int s0 = (RT * wave) >> 1;        // first K-step of the B layout this wave's outputs belong to (RT even)

  // ---- resident weights: rows 16 RT wave .. of both matrices, all 8 K-steps (packed stream: [s][16 tiles][lane][8]) ----
  bf16x8 wm[RT][8], wo[RT][8];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const size_t off = ((size_t)(s * 16 + RT * wave + t) * 64 + lane) * 16;
      wm[t][s] = *(const bf16x8*)(a.w_mid + off);
      wo[t][s] = *(const bf16x8*)(a.w_out + off);
    }
  // biases / LayerNorm parameters live in LDS (4 KiB) and are re-read where each phase needs them
  if (threadIdx.x < 256) {
    float* par_w = (float*)(lds + kOffPar);
    const int i = threadIdx.x;
    par_w[i] = a.b_mid[i];
    par_w[256 + i] = a.b_out[i];
    par_w[512 + i] = a.gamma[i];
    par_w[768 + i] = a.beta[i];
    par_w[1024 + i] = a.b1[i];
  }
  const float* const par_l = (const float*)(lds + kOffPar) + f0;  // this lane's slice: + 256 * which + 16 * t

  char* const h2 = lds + kOffH2;
  float* const stage = (float*)(lds + kOffStage);
  int* const gdl = (int*)(lds + kOffGd);
  float* const lnp = (float*)(lds + kOffLn);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;  // LDS byte address of the window

  // ---- tile walk: XCD x = workgroup & 7 owns a contiguous range of edge blocks and walks it batch-innermost ----
  const int slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const TileWalk tw = tile_walk(blockIdx.x & 7, a.neb, a.batch);

  // 32 KiB of layer-1 activations of unit u -> Hbuf1[par]: 32 LDS-DMA pieces of 1 KiB, PPW per wave (asynchronous).  The DMA
  // queue of a wave is shallow - eight back-to-back issues stall it for ~5 k cycles (measured) - so in steady state the
  // pieces are issued a few per 16-column group of the output layer, when no other load is outstanding.
  auto tile_index = [&](int u) -> size_t {
    const int eb = tw.eb_start + u / a.batch;
    const int b = u - (u / a.batch) * a.batch;
    return (size_t)(b * a.neb + eb);
  };
  auto prefetch_piece = [&](const char* src, int par, int i) {
    const int piece = PPW * wave + i;
    glds16_asm_s((const float*)(src + piece * 1024), (unsigned)lane * 16u,
                 __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kOffH1 + par * kHBytes + piece * 1024)));
  };
  if (!GATHER && slot < tw.n_units) {
    const char* src = a.h1g + tile_index(slot) * kHBytes;
#pragma unroll
    for (int i = 0; i < PPW; ++i) prefetch_piece(src, 0, i);
  }
  // GATHER: thread -> (column, 16-byte piece) pairs of the tile: 8 lanes read one 128-byte line of a projected row (as in
  // edge16_gather_kernel); the row indices of a tile are fetched one tile ahead.
  constexpr int GP = GATHER ? 512 / (64 * NW) : 1;  // column passes (1 for 512 threads, 2 for 256)
  const int gpiece = threadIdx.x & 7;
  int gidx[GP][3];
  auto load_gather_indices = [&](int u) {
    const int eb = tw.eb_start + u / a.batch;
#pragma unroll
    for (int cp = 0; cp < GP; ++cp) {
      const int kr = eb * kTileCols + cp * (8 * NW) + (int)(threadIdx.x >> 3);
      const int k = kr < a.n_edges ? kr : a.n_edges - 1;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        gidx[cp][p] = p < a.n_proj ? (a.p_kind[p] == 0 ? ldgi(a.src + k) : (a.p_kind[p] == 1 ? ldgi(a.dst + k) : k)) : 0;
#pragma unroll
      for (int p = 0; p < 3; ++p) gidx[cp][p] = gidx[cp][p] < 0 ? 0 : gidx[cp][p];  // (padding columns of segment-aligned tiles: dst = -1)
    }
  };
  // GATHER: layer 1 of a tile = relu(b1 + sum_p P_p[row_p]) -> bf16 -> Hbuf1[buffer], at the top of the tile.  The rows it
  // reads were touched one tile earlier by gather_warm() (one 4-byte load per thread and table: thread (column, piece) touches
  // line `piece` of the column's row), so they come from this XCD's L2 instead of paying the HBM round trip of the
  // batch-shared tables; the row indices are fetched one tile ahead as well.
  auto gather_finish = [&](int u, int buffer, bool more) {
    const int eb = tw.eb_start + u / a.batch;
    const int b = u - (u / a.batch) * a.batch;
    int gp4 = 4 * gpiece;
    asm volatile("" : "+v"(gp4));
    const float* b1l = (const float*)(lds + kOffPar) + 1024 + gp4;
#pragma unroll
    for (int cp = 0; cp < GP; ++cp) {
      const int col = cp * (8 * NW) + (int)(threadIdx.x >> 3);
      const bool cvalid = eb * kTileCols + col < a.n_edges;
      char* out = lds + kOffH1 + buffer * kHBytes + (col >> 4) * 8192 + (16 * (gpiece & 3) + (col & 15)) * 16 + (gpiece >> 2) * 8;
      // every load of the tile in flight at once: ONE round trip (two half-tile passes measured 7.4 k cycles against 5.9 k)
      f32x4 z[8], v[8];  // table 0 lands in z, table 1 in v: both in flight together (64 registers)
      {
        const float* row = a.p_ptr[0] + ((size_t)b * (size_t)a.p_rows_pb[0] + (size_t)gidx[cp][0]) * (size_t)a.p_ld[0] + gp4;
#pragma unroll
        for (int s = 0; s < 8; ++s) z[s] = ldg4(row + 32 * s);
      }
#pragma unroll
      for (int p = 1; p < 3; ++p)
        if (p < a.n_proj) {
          const float* row = a.p_ptr[p] + ((size_t)b * (size_t)a.p_rows_pb[p] + (size_t)gidx[cp][p]) * (size_t)a.p_ld[p] + gp4;
#pragma unroll
          for (int s = 0; s < 8; ++s) v[s] = ldg4(row + 32 * s);
#pragma unroll
          for (int s = 0; s < 8; ++s) z[s] += v[s];
        }
#pragma unroll
      for (int s = 0; s < 8; ++s) z[s] += *(const f32x4*)(b1l + 32 * s);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bf16x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = (__bf16)(cvalid ? fmaxf(z[s][r], 0.f) : 0.f);
        *(bf16x4*)(out + s * 1024) = o4;
      }
    }
    if (more) load_gather_indices(u + nslot);  // the indices of the tile after this one: in flight for a whole tile
  };
  int warm = 0;
  auto gather_warm = [&](int u) {  // (call when the indices of tile u have arrived: gidx)
    const int b = u - (u / a.batch) * a.batch;
#pragma unroll
    for (int cp = 0; cp < GP; ++cp)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (p < a.n_proj) {
          const float* row = a.p_ptr[p] + ((size_t)b * (size_t)a.p_rows_pb[p] + (size_t)gidx[cp][p]) * (size_t)a.p_ld[p];
          warm ^= ldgi((const int*)(row + 32 * gpiece));
        }
  };
  if (GATHER && slot < tw.n_units) {
    load_gather_indices(slot);
    __syncthreads();  // the parameter block (b1) is visible
  }

  int par = 0;
#pragma unroll 1
  for (int u = slot; u < tw.n_units; u += nslot) {
    const int eb = tw.eb_start + u / a.batch;
    const int b = u - (u / a.batch) * a.batch;
    const int k0 = eb * kTileCols;
    const size_t tile = (size_t)(b * a.neb + eb);
    const char* h1 = lds + kOffH1 + (GATHER ? 0 : par * kHBytes);

    unsigned long long ts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamp = a.dbg != nullptr && u == slot + 2 * nslot;
    if (stamp) ts[0] = gw_clock();
    const bool more = u + nslot < tw.n_units;
    if constexpr (GATHER) {
      gather_finish(u, 0, more);  // layer 1 of this tile -> Hbuf1[0] (its readers of the previous tile left before barrier (2))
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces (and its stores of the previous tile) are done
    }
    if (stamp) ts[1] = gw_clock();
    wg_barrier();  // (1) Hbuf1 complete; every wave has left the previous tile's last phase
    if (stamp) ts[2] = gw_clock();
    // next tile -> Hbuf1[par ^ 1] (last read before barrier (2) of the previous tile), a few pieces per group step below
    const char* nsrc = (!GATHER && more) ? a.h1g + tile_index(u + nslot) * kHBytes : nullptr;

    f32x4 bmv[RT];  // (after barrier (1): on the first tile it also publishes the parameter block)
#pragma unroll
    for (int t = 0; t < RT; ++t) bmv[t] = *(const f32x4*)(par_l + 16 * t);
    int kk[kGroups];
    bool valid[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const int kr = k0 + 16 * g + j;
      valid[g] = kr < a.n_edges;
      kk[g] = valid[g] ? kr : a.n_edges - 1;
    }
    int gd_mine = -1;  // wave 0: destination row of column `lane` (written to LDS with the staged tile)
    if (wave == 0) {
      const int kr = k0 + lane;
      const int dk = kr < a.n_edges ? ldgi(a.dst + kr) : -1;
      gd_mine = dk >= 0 ? b * a.n_dst + dk : -1;  // (dst < 0: a padding column of segment-aligned tiles - never summed)
    }

    // ---- residual: requested here, used after barrier (3) - its latency passes under the middle layer, and it is back
    // before the DMA pieces of the next tile are issued (the wave's memory queue is shallow).  fp32 rows: 16-byte pieces of
    // this lane's features; bf16 edge tiles: the lane's own 16-byte slots of K-steps s0 .. (its features are exactly those).
    f32x4 resv[RES_TILES ? 1 : kGroups][RES_TILES ? 1 : RT];
    bf16x8 rest[RES_TILES ? kGroups : 1][RES_TILES ? KS : 1];
    if constexpr (RES_TILES) {
      const size_t rtile = a.res_tiles_shared ? (size_t)eb : tile;
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          rest[g][ks] = *(const GW_AS1 bf16x8*)(a.res_tiles + (rtile * kGroups + g) * 8192 + (size_t)(s0 + ks) * 1024 + (size_t)lane * 16);
    }
    if (stamp) ts[3] = gw_clock();
    // ---- middle layer -> Hbuf2 (the residual rows of the fp32-row form are requested one group per step: a burst of 8+
    // loads stalls the wave's memory queue) ----
    bf16x8 bfr[8];
    load_frags(bfr, h1, lane);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      f32x4 acc[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = bmv[t];
      layer_group(acc, wm, bfr);
      // the MFMAs have read their operands: the same registers take the next group's fragments, whose LDS latency passes
      // under this group's epilogue
      if (g + 1 < kGroups) load_frags(bfr, h1 + (g + 1) * 8 * 1024, lane);
      if constexpr (!RES_TILES) {
        const float* rrow = a.res_ptr + ((size_t)b * (size_t)a.res_rows_pb + (size_t)kk[g]) * (size_t)a.res_ld + f0;
#pragma unroll
        for (int t = 0; t < RT; ++t) resv[g][t] = ldg4(rrow + 16 * t);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int T = RT * wave + t;  // output row tile -> K-step T >> 1, half T & 1 of this lane's slot
        *(bf16x4*)(h2 + ((g * 8 + (T >> 1)) * 64 + lane) * 16 + (T & 1) * 8) = to_bf16x4(relu4(acc[t]));
      }
      if (stamp && g < 4) ts[4 + g] = gw_clock();
    }
    wg_barrier();  // (2) Hbuf2 complete, Hbuf1 free
    if (stamp) ts[8] = gw_clock();
    if constexpr (GATHER) {
      if (more) gather_warm(u + nslot);  // the next tile's rows -> L2 (its indices, requested at the top of this tile, are here)
    }

    f32x4 bov[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) bov[t] = *(const f32x4*)(par_l + 256 + 16 * t);

    // ---- output layer + LayerNorm partial sums ----
    f32x4 o[kGroups][RT];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      if (!GATHER && more) {
#pragma unroll
        for (int i = 0; i < PPW / kGroups; ++i) prefetch_piece(nsrc, par ^ 1, (PPW / kGroups) * g + i);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) o[g][t] = bov[t];
      if constexpr (RES_TILES && NW == 8) {
        // room for all 8 fragments: one LDS wait per group, and the next group's reads are issued behind this group's MFMAs
        if (g == 0) load_frags(bfr, h2, lane);
        layer_group(o[g], wo, bfr);
        if (g + 1 < kGroups) load_frags(bfr, h2 + (g + 1) * 8 * 1024, lane);
      } else {
        layer_group_lds(o[g], wo, h2 + g * 8 * 1024, lane);
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += o[g][t][r];
          s2 = fmaf(o[g][t][r], o[g][t][r], s2);
        }
      s1 = sum_rows(s1);
      s2 = sum_rows(s2);
      if (q == 0) *(float2*)(lnp + (wave * kTileCols + 16 * g + j) * 2) = float2{s1, s2};
    }
    if (stamp) ts[9] = gw_clock();
    wg_barrier();  // (3) partial sums of all feature slices visible
    if (stamp) ts[10] = gw_clock();
    f32x4 gmv[RT], btv[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      gmv[t] = *(const f32x4*)(par_l + 512 + 16 * t);
      btv[t] = *(const f32x4*)(par_l + 768 + 16 * t);
    }

    // ---- LayerNorm (eps 1e-5, biased variance), residual, staging, e' tile ----
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const int col = 16 * g + j;
      float s1 = 0.f, s2 = 0.f;  // row q of the wave reads the partial sums of waves q, q + 4, ...; sum_rows combines the rows
#pragma unroll
      for (int w4 = 0; w4 < NW / 4; ++w4) {
        const float2 pr = *(const float2*)(lnp + ((q + 4 * w4) * kTileCols + col) * 2);
        s1 += pr.x;
        s2 += pr.y;
      }
      s1 = sum_rows(s1);
      s2 = sum_rows(s2);
      const float mean = s1 * (1.0f / 256.0f);
      const float var = fmaxf(s2 * (1.0f / 256.0f) - mean * mean, 0.f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
      float* srow = stage + col * kStageLd + f0;
      f32x4 v[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        f32x4 rv;
        if constexpr (RES_TILES) {
          const bf16x8 rr = rest[g][t >> 1];
          rv = (t & 1) ? f32x4{(float)rr[4], (float)rr[5], (float)rr[6], (float)rr[7]}
                       : f32x4{(float)rr[0], (float)rr[1], (float)rr[2], (float)rr[3]};
        } else {
          rv = resv[g][t];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = (o[g][t][r] - mean) * rstd * gmv[t][r] + btv[t][r] + rv[r];
        *(f32x4*)(srow + 16 * t) = v[t];
        if (NW == 4 && a.e_out != nullptr && valid[g]) stg4(a.e_out + ((size_t)b * a.n_edges + kk[g]) * 256 + f0 + 16 * t, v[t]);
      }
      if (a.e_out_tiles != nullptr) {  // e' as bf16 edge tiles: this lane's own 16-byte slots, 1 KiB per wave instruction
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 pk = to_bf16x8(v[2 * ks], v[2 * ks + 1]);
          if (!valid[g]) pk = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          *(GW_AS1 bf16x8*)(a.e_out_tiles + (tile * kGroups + g) * 8192 + (size_t)(s0 + ks) * 1024 + (size_t)lane * 16) = pk;
        }
      }
    }
    if (wave == 0) gdl[lane] = gd_mine;
    if (stamp) ts[11] = gw_clock();
    wg_barrier();  // (4) staged tile + destination ids visible
    if (stamp) ts[12] = gw_clock();

    {
      // ---- per-feature segment sums over the 64 destination-sorted columns ----
      // Thread (f, h) owns feature f = thread & 255 for the columns 32 h .. 32 h + 31 (NW = 8: h = 0, 1; NW = 4: one thread walks
      // both halves).  All LDS reads first (independent), then a straight-line walk over registers.  Segment ends are the same
      // for every thread: lane i compares column i's destination with column i + 1's, the ballot is a 64-bit scalar mask, and
      // the walk tests one bit per column (a scalar branch that is rarely taken).  A segment that is wholly inside the
      // thread's columns is complete: plain store.  The first and the last one may continue elsewhere - in the neighbouring
      // tiles, or across the middle of the tile when column 31 does not end a segment - and are added with atomics.
      constexpr int HALVES = NW / 4, COLS = kTileCols / HALVES;
      int f = threadIdx.x & 255;
      asm volatile("" : "+v"(f));  // keeps agg + f out of the kernel-lifetime registers (it was hoisted out of the tile loop and spilled)
      const int hh = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
      const int c0 = hh * COLS;
      constexpr int VH = COLS;  // columns read from LDS at a time
      float vv[VH];
#pragma unroll
      for (int i = 0; i < VH; ++i) vv[i] = stage[(c0 + i) * kStageLd + f];
      const int gdv = gdl[lane];
      const int gdn = gdl[lane < kTileCols - 1 ? lane + 1 : lane];
      const unsigned long long ends = __ballot(lane == kTileCols - 1 || gdn != gdv);  // bit i: a segment ends with column i
      const bool mid_open = HALVES == 2 && ((ends >> (COLS - 1)) & 1ull) == 0;        // a segment straddles columns 31 | 32
      const unsigned long long mine = (ends >> c0) | (1ull << (COLS - 1));              // ... the thread's walk ends there anyway
      if (stamp) ts[13] = gw_clock();
      // deterministic mode (carry records instead of atomics, NW == 4 only: one thread walks all 64 columns): is the first /
      // last run open towards the neighbouring tile of the same batch element?
      bool det_lo = false, det_hi = false;
      float* rec = nullptr;
      if (HALVES == 1 && a.carry != nullptr) {
        rec = a.carry + tile * kCarryFloats;
        const int gd_prev = eb > 0 ? b * a.n_dst + ldgi(a.dst + k0 - 1) : -2;
        const int gd_next = k0 + kTileCols < a.n_edges ? b * a.n_dst + ldgi(a.dst + k0 + kTileCols) : -2;
        det_lo = gd_prev == __builtin_amdgcn_readlane(gdv, 0);
        det_hi = gd_next == __builtin_amdgcn_readlane(gdv, kTileCols - 1);
        if (f == 0) {
          rec[512] = __int_as_float(-1);
          rec[513] = __int_as_float(-1);
          rec[514] = __int_as_float(0);
        }
      }
      float run = 0.f;
      bool first = true;
#pragma unroll
      for (int i = 0; i < COLS; ++i) {
        if (VH < COLS && i > 0 && i % VH == 0) {  // next batch of staged values
#pragma unroll
          for (int i2 = 0; i2 < VH; ++i2) vv[i2] = stage[(c0 + i + i2) * kStageLd + f];
        }
        run += vv[i % VH];
        if (__builtin_expect((mine >> i) & 1ull, 0)) {
          const int cur = __builtin_amdgcn_readlane(gdv, c0 + i);
          if (cur >= 0 && GW_SKIP(a) != 1) {
            float* dstp = a.agg + (size_t)cur * 256 + f;
            if (rec != nullptr) {
              const bool lo = first && det_lo, hi = i == COLS - 1 && det_hi;
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
            } else {
              const bool open_lo = first && (hh == 0 || mid_open);                  // may continue before this thread's columns
              const bool open_hi = i == COLS - 1 && (hh == HALVES - 1 || mid_open);  // ... or after them
              if (open_lo || open_hi) __hip_atomic_fetch_add((GW_AS1 float*)dstp, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else stg1(dstp, run);
            }
          }
          first = false;
          run = 0.f;
        }
      }
    }
    if (NW == 8 && a.e_out != nullptr) {
      // ---- e' as fp32 rows from the staged tile: one coalesced 1 KiB store per row, 8 rows per wave ----
      int l4 = 4 * lane;
      asm volatile("" : "+v"(l4));  // (keeps e_out + 4 lane out of the kernel-lifetime registers)
#pragma unroll 4
      for (int i = 0; i < 8; ++i) {
        const int col = 8 * wave + i;
        if (k0 + col < a.n_edges) {
          const f32x4 v = *(const f32x4*)(stage + col * kStageLd + l4);
          stg4(a.e_out + ((size_t)b * a.n_edges + k0 + col) * 256 + l4, v);
        }
      }
    }
    // the next tile's barrier (1) separates these reads from the next Hbuf2 / staging writes
    if (stamp) {
      ts[14] = gw_clock();
      if (threadIdx.x == 0 && (int)blockIdx.x < a.dbg_cap)
        for (int i = 0; i < 15; ++i) a.dbg[(size_t)blockIdx.x * 16 + i] = ts[i];
    }
    par ^= 1;
  }
}

template <typename K>
int launch_resident(K kernel, int threads, int n_wg, const Edge16Args& a, void* stream) {
  static DeviceOnce once;  // per template instantiation and device
  if (once.first()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
  hipLaunchKernelGGL(kernel, dim3((unsigned)n_wg), dim3((unsigned)threads), kLdsTotal, (hipStream_t)stream, a);
  return check_launch("edge16_kernel launch");
}

inline bool is_proj16(const gw_operand* o) { return o->k > 0 && o->projected != 0; }
inline bool is_raw16(const gw_operand* o) { return o->k > 0 && o->projected == 0; }

}  // namespace

namespace gw {

// Eligible: bf16 weights, one middle layer, and layer 1 = projected operands (+ at most the edge operand raw, as bf16 tiles).
bool edge16_eligible(const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in, const gw_mlp_weights* w) {
  if (w->weight_dtype != GW_DTYPE_BF16 || w->n_mid != 1 || !w->ln_gamma) return false;
  if (w->ln_width > 0 && w->ln_width != 256) return false;
  if (is_raw16(x_src) || is_raw16(x_dst)) return false;
  // node operands: fp32 rows, or (projected only) fp16 rows of layer-1 products
  const gw_operand* nodes[2] = {x_src, x_dst};
  for (const gw_operand* o : nodes)
    if (o->layout != GW_LAYOUT_ROWS_F32 && !(o->layout == GW_LAYOUT_ROWS_F16 && is_proj16(o))) return false;
  const bool raw_e = is_raw16(e_in);
  if (raw_e && (e_in->layout != GW_LAYOUT_EDGE_TILES_BF16 || !w->w1[2])) return false;
  if (!raw_e && e_in->k > 0 && e_in->layout != GW_LAYOUT_ROWS_F32) return false;
  const int n_proj = (is_proj16(x_src) ? 1 : 0) + (is_proj16(x_dst) ? 1 : 0) + (is_proj16(e_in) ? 1 : 0);
  if (n_proj < 1 && !raw_e) return false;
  static const int impl = GW_TUNE("GW_EDGE16_IMPL", 1);  // tuning builds: 0 forces the streaming kernel of gw_bf16.hip
  return impl != 0;
}

size_t edge16_workspace_bytes(int32_t batch, int32_t n_edges) {
  return (size_t)batch * (size_t)((n_edges + kTileCols - 1) / kTileCols) * (size_t)kHBytes;
}
// tuning builds: GW_EDGE16_FUSE_GATHER=0 restores the two-launch form (gather kernel + workspace) of the all-projected case
static bool fuse_gather_on() {
  static const bool on = GW_TUNE("GW_EDGE16_FUSE_GATHER", 1) != 0;
  return on;
}
// Workspace of an edge16 call: the layer-1 tiles (only when a separate launch makes them: raw edge operand, or the
// two-launch gather form), then the carry records of the deterministic mode.
size_t edge16_workspace_needed(int32_t batch, int32_t n_edges, const gw_operand* e_in, bool deterministic) {
  const bool raw_e = e_in->k > 0 && e_in->projected == 0;
  const size_t h1 = (raw_e || !fuse_gather_on()) ? edge16_workspace_bytes(batch, n_edges) : 0;
  return h1 + (deterministic ? segment_carry_bytes((int64_t)batch * ((n_edges + kTileCols - 1) / kTileCols)) : 0);
}

int edge16_rows_to_tiles(int32_t batch, int32_t n_edges, const float* rows, int32_t rows_per_batch, int32_t ld, void* tiles,
                         void* stream) {
  const int neb = (n_edges + kTileCols - 1) / kTileCols;
  hipLaunchKernelGGL(rows_to_tiles_kernel, dim3((unsigned)(batch * neb)), dim3(256), 0, (hipStream_t)stream, batch, n_edges, neb, rows,
                     rows_per_batch, ld, (char*)tiles);
  return check_launch("rows_to_tiles_kernel launch");
}

int edge16_launch(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst, const gw_operand* x_src,
                  const gw_operand* x_dst, const gw_operand* e_in, const gw_operand* e_res, const gw_mlp_weights* w,
                  float* e_out, void* e_out_tiles, float* agg, int32_t n_dst, void* workspace, int32_t flags, void* stream) {
  const bool deterministic = (flags & GW_EDGE_DETERMINISTIC) != 0;
  Edge16Args a;
  memset(&a, 0, sizeof(a));
  a.seg_tiles = (flags & GW_EDGE_SEGMENT_TILES) != 0;
  a.agg_bf16k = (flags & GW_EDGE_AGG_BF16K) != 0;
  a.seg_split = (flags & GW_EDGE_SEGMENT_SPLIT) != 0;
  a.batch = batch;
  a.n_edges = n_edges;
  a.n_dst = n_dst;
  a.neb = (n_edges + kTileCols - 1) / kTileCols;
  a.src = src;
  a.dst = dst;
  const gw_operand* ops[3] = {x_src, x_dst, e_in};
  for (int i = 0; i < 3; ++i)
    if (is_proj16(ops[i])) {
      a.p_ptr[a.n_proj] = ops[i]->ptr;
      a.p_rows_pb[a.n_proj] = ops[i]->rows_per_batch;
      a.p_ld[a.n_proj] = ops[i]->ld;
      a.p_kind[a.n_proj] = i;
      a.p_half[a.n_proj] = ops[i]->layout == GW_LAYOUT_ROWS_F16;
      ++a.n_proj;
    }
  const bool raw_e = is_raw16(e_in);
  a.b1 = w->b1;
  a.w_raw = raw_e ? (const char*)w->w1[2] : nullptr;
  a.e_tiles = raw_e ? (const char*)e_in->ptr : nullptr;
  a.e_tiles_shared = raw_e && e_in->rows_per_batch == 0;
  a.w_mid = (const char*)w->w_mid;
  a.b_mid = w->b_mid;
  a.w_out = (const char*)w->w_out;
  a.b_out = w->b_out;
  a.gamma = w->ln_gamma;
  a.beta = w->ln_beta;
  const bool no_res = e_res->k == 0 || !e_res->ptr;  // no residual: the caller pre-added the segment sums of e (team kernel only)
  if (no_res) {
    if (e_out || e_out_tiles) return set_error(GW_E_UNSUPPORTED, "edge16: e' was requested without its residual operand");
  } else if (e_res->layout == GW_LAYOUT_EDGE_TILES_BF16) {
    a.res_tiles = (const char*)e_res->ptr;
    a.res_tiles_shared = e_res->rows_per_batch == 0;
  } else {
    a.res_ptr = e_res->ptr;
    a.res_rows_pb = e_res->rows_per_batch;
    a.res_ld = e_res->ld;
  }
  a.e_out = e_out;
  a.e_out_tiles = (char*)e_out_tiles;
  a.agg = agg;
  a.h1g = (char*)workspace;
  if (deterministic)  // behind the layer-1 tiles, if this call has any
    a.carry = (float*)((char*)workspace + ((raw_e || !fuse_gather_on()) ? edge16_workspace_bytes(batch, n_edges) : 0));
#ifdef GW_TUNING
  {
    static const int skip = GW_TUNE("GW_EDGE16_SKIP", 0);
    a.skip = skip;
    static const int tune = GW_TUNE("GW_EDGE16_TUNE", 0);
    a.tune = tune;
  }
#endif
  if (g_dbg != nullptr && g_dbg_kind == 3) {
    a.dbg = g_dbg;
    a.dbg_cap = g_dbg_cap;
  }
  static const int n_wg = (GW_TUNE("GW_EDGE16_WGS", 256) + 7) / 8 * 8;  // persistent workgroups: one per CU, a multiple of 8 (XCD round-robin)
  const bool fuse_gather = fuse_gather_on();
  static const int nw = GW_TUNE("GW_EDGE16_NW", 8);
  static const int team = GW_TUNE("GW_EDGE16_TEAM", 1);
  // what only the team-pipelined kernel takes - checked before anything is enqueued (launch 1 below writes the workspace)
  bool any_half = false;
  for (int p = 0; p < a.n_proj; ++p) any_half = any_half || a.p_half[p];
  const bool team_only = nw == 4 || deterministic || team == 0;
  // fp16 product rows are read by the layer-1 kernel (raw edge operand) and by the team kernel's gather - not by the lock-step
  // kernels' gather
  if (any_half && !raw_e && (team_only || !fuse_gather))
    return set_error(GW_E_UNSUPPORTED, "edge16: fp16 product rows with all operands projected need the team-pipelined kernel (atomics mode)");
  if (no_res && team_only)
    return set_error(GW_E_UNSUPPORTED, "edge16: an edge update without residual runs on the team-pipelined kernel only (atomics mode)");
  // segment-aligned tiles: (a) every operand projected, no residual, no e' -> the gather form of the team kernel (agg rows WRITTEN);
  // (b) raw edge tiles + per-sample residual tiles -> layer-1 kernel + edge16p_kernel (agg rows += in place: the running sum of
  // the processor stack); (c) every operand projected with a (batch-shared) residual - the first processor block - runs on
  // the lock-step kernel, which only has to skip the padding columns (agg += by atomics on the caller's zero fill)
  if (a.seg_tiles && (team_only || !fuse_gather || a.res_ptr != nullptr || e_out != nullptr || (raw_e && (no_res || a.res_tiles_shared)) ||
                      (a.agg_bf16k && (raw_e || !no_res || a.seg_split)) || (a.seg_split && (raw_e || !no_res))))
    return set_error(GW_E_UNSUPPORTED, "edge16: segment-aligned tiles take projected operands without residual, or bf16 edge tiles "
                                       "(operand and residual), atomics mode, e' as tiles or dropped");
  if (no_res || (any_half && !raw_e)) {
    int n_dyn = 0;
    for (int p = 0; p < a.n_proj; ++p) n_dyn += a.p_rows_pb[p] != 0 ? 1 : 0;
    if (n_dyn != 1)
      return set_error(GW_E_UNSUPPORTED, "edge16: an edge update without residual / with fp16 product rows / on segment-aligned tiles "
                                         "needs exactly one per-sample projected table");
  }
  // launch 1: layer 1 -> workspace tiles
  if (raw_e) {
    static DeviceOnce once_l1;
    if (once_l1.first()) (void)hipFuncSetAttribute((const void*)edge16_l1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kL1Lds);
    hipLaunchKernelGGL(edge16_l1_kernel, dim3((unsigned)n_wg), dim3(64 * kL1Waves), kL1Lds, (hipStream_t)stream, a);
    if (int rc = check_launch("edge16_l1_kernel launch")) return rc;
  } else if (!fuse_gather) {
    hipLaunchKernelGGL(edge16_gather_kernel, dim3((unsigned)a.neb), dim3(256), 0, (hipStream_t)stream, a);  // one workgroup per edge block
    if (int rc = check_launch("edge16_gather_kernel launch")) return rc;
  }
  if (a.seg_tiles && raw_e) return edge16p_launch(&a, n_wg, stream);  // (b): the processor form on segment-aligned tiles
  // launch 2: the resident layers
  const bool rt = a.res_tiles != nullptr;
  const bool ga = !raw_e && fuse_gather;  // layer 1 gathered inside the resident kernel
  int rc;
  if (nw == 4 || deterministic) {  // the deterministic walk is a whole-tile walk per thread: the 4-wave form
    if (ga) rc = rt ? launch_resident(edge16_kernel<4, true, true>, 256, n_wg, a, stream) : launch_resident(edge16_kernel<4, false, true>, 256, n_wg, a, stream);
    else rc = rt ? launch_resident(edge16_kernel<4, true, false>, 256, n_wg, a, stream) : launch_resident(edge16_kernel<4, false, false>, 256, n_wg, a, stream);
    if (rc != GW_OK || !deterministic) return rc;
    return segment_fixup_launch((int64_t)batch * a.neb, a.carry, agg, stream);
  }
  // team-pipelined form (gw_edge16t.hip): residual as bf16 tiles or none, atomics mode; the gather form needs one or two per-sample tables
  if (team != 0 && ((ga && no_res) || (!ga && rt)) && nw != 4 && !deterministic) {
    int n_dyn = 0, n_shared = 0;
    for (int p = 0; p < a.n_proj; ++p) (a.p_rows_pb[p] != 0 ? n_dyn : n_shared) += 1;
    if (!ga || n_dyn == 1) {  // (two per-sample tables: the lock-step kernel below is the faster one, see gw_edge16t.hip)
      // chunk = batch elements of one edge block a workgroup walks in a row: as many as leave every workgroup >= 16 units
      int bc = 1;
      if (ga && n_shared > 0)
        for (int d = 1; d <= batch; ++d)
          if (batch % d == 0 && (long long)a.neb * (batch / d) >= 16ll * n_wg) bc = d;
      static const int bc_force = GW_TUNE("GW_EDGE16_BC", 0);
      if (bc_force > 0 && batch % bc_force == 0) bc = bc_force;
      a.bc = bc;
      a.nchunk = batch / bc;
      return edge16t_launch(&a, ga, n_wg, stream);
    }
  }
  if (ga) return rt ? launch_resident(edge16_kernel<8, true, true>, 512, n_wg, a, stream) : launch_resident(edge16_kernel<8, false, true>, 512, n_wg, a, stream);
  return rt ? launch_resident(edge16_kernel<8, true, false>, 512, n_wg, a, stream) : launch_resident(edge16_kernel<8, false, false>, 512, n_wg, a, stream);
}

}  // namespace gw
