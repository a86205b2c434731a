// gw_edge16.hip - bf16-MFMA edge update with the weights held in registers (BASELINE.json configs[2]).
//
//   e'[c] = LayerNorm(W_out . relu(W_mid . relu(z1[c]) + b_mid) + b_out) + e_res[c],   agg[dst(c)] += e'[c]
//   z1[c] = b1 + sum_p P_p[row_p(c)]  (+ W_raw . x_raw[c])        (layer 1 split: projected operands are gathered)
//
// The first bf16 version (gw_bf16.hip) streams the packed weights through LDS for every 128-column tile like the fp32
// kernels do.  At bf16 MFMA rates (16x fp32) that stream - 128 KiB per layer and tile - and its barriers cost ten times
// the matrix time (measured: 10 % MFMA utilisation).  Here the weights never move: a workgroup is persistent (one per
// CU), wave w keeps rows 64w .. 64w+63 of every layer's matrix in registers as MFMA A fragments (4 row tiles x 8 K-steps
// x 4 VGPRs = 128 VGPRs per layer), and the ACTIVATIONS travel instead - 16-column groups, 8 KiB per layer as bf16,
// exchanged between the four waves through LDS in the B-operand layout
//       Hbuf[group][K-step s][lane][8 x bf16],   k(s, q, i) = 32 s + 16 (i >> 2) + 4 q + (i & 3)
// (the K order of gw_pack_linear_bf16, so wave w / row tile t of the producing layer writes the 8-byte half
//  s = 2w + (t >> 1), half = t & 1 of its own lane: no shuffles).
//
// A tile is 64 consecutive destination-sorted edges of ONE batch element.  Tiles are walked batch-innermost and XCD-aware:
// the 32 workgroups of an XCD work on the same two edge blocks of all batch elements at a time, so rows of batch-shared
// tables (the cached per-edge products and edge embeddings of the encoder / decoder) are fetched from HBM once and hit in
// that XCD's L2 for the other batch elements.
//
// Two launches per edge update:
//  1. edge16_gather_kernel - the layer-1 gather-add (b1 + sum of projected rows), relu, bf16, written to a workspace in the
//     B-operand layout, 32 KiB per tile.  Pure data movement with thousands of waves in flight: the dependent
//     index -> row -> 16-byte-piece loads that a one-workgroup-per-CU kernel cannot hide are hidden by occupancy here.
//  2. edge16_kernel (persistent, weights in registers) - per tile: the 32 KiB of layer-1 activations arrive by LDS-DMA,
//     prefetched one tile ahead | barrier | middle layer (32 MFMAs per group and wave) -> Hbuf2 | barrier | output layer,
//     LayerNorm partial sums through LDS | barrier | LayerNorm, residual, [e' store], staging | barrier | per-feature
//     segment sums (plain stores for segments inside the tile, atomics for the two that may continue in a neighbour).
// Everything outside the matrix products is fp32, as in gw_bf16.hip.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kTileCols = 64;                       // columns (edges) per tile: 4 groups of 16
constexpr int kGroups = 4;
constexpr int kHBytes = kGroups * 8 * 1024;         // one activation exchange buffer: 4 groups x 8 K-steps x 1 KiB
constexpr int kStageLd = 260;                       // staging row stride in floats (256 + 4: conflict-free column walks)
constexpr int kOffH1 = 0;                           // two layer-1 buffers (DMA prefetch of the next tile)
constexpr int kOffH2 = 2 * kHBytes;                 // layer-2 activations; the staging area reuses it (dead by then)
constexpr int kOffStage = kOffH2;
constexpr int kOffGd = kOffStage + kTileCols * kStageLd * 4;
constexpr int kOffLn = kOffGd + kTileCols * 4;
constexpr int kOffPar = kOffLn + 4 * kTileCols * 8;    // after [wave][column] (sum, sum of squares): b_mid, b_out, gamma, beta
constexpr int kLdsTotal = kOffPar + 4 * 256 * 4;
static_assert(kTileCols * kStageLd * 4 >= kHBytes, "staging area covers Hbuf2");
static_assert(kLdsTotal <= 160 * 1024, "LDS budget of one CU");

struct Edge16Args {
  int batch, n_edges, n_dst;
  int neb;  // edge blocks of 64
  const int* src;
  const int* dst;
  int n_proj;
  const float* p_ptr[3];
  int p_rows_pb[3];
  int p_ld[3];
  int p_kind[3];  // 0: row = src[k], 1: dst[k], 2: k
  const float* b1;
  const char* w_mid;
  const float* b_mid;
  const char* w_out;
  const float* b_out;
  const float* gamma;
  const float* beta;
  const float* res_ptr;
  int res_rows_pb;
  int res_ld;
  float* e_out;
  float* agg;
  char* h1g;  // workspace: layer-1 activations, [batch * neb tiles][4 groups][8 K-steps][64 lanes][8 bf16]
  int skip;                 // tuning aid (GW_EDGE16_SKIP): 1 = no aggregate writes (results are then wrong)
  unsigned long long* dbg;  // gw_debug_timestamps(kind 3): phase clocks of each workgroup's third tile
  int dbg_cap;
};

__device__ __forceinline__ void wg_barrier() { __syncthreads(); }

__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
  return r;
}

// B fragments of one 16-column group: 8 K-steps x 16 bytes per lane, all reads issued back to back
__device__ __forceinline__ void load_frags(bf16x8 (&bf)[8], const char* __restrict__ hbuf_g, int lane) {
#pragma unroll
  for (int s = 0; s < 8; ++s) bf[s] = *(const bf16x8*)(hbuf_g + s * 1024 + lane * 16);
}
// One resident layer on one 16-column group: acc[t] (4 row tiles of this wave) += W[tile t][K-step s] . B[s].
// The MFMAs are written as asm with the weight fragment constrained to an accumulation register ("a"): the 256 weight
// registers then live in the AGPR half of the file for the whole kernel and feed the matrix cores from there.  Left to
// itself the allocator treats AGPRs as spill space and copies every fragment back to a VGPR before use (~700 copies per
// tile, measured).  Inline asm is opaque to the hazard recogniser, so the wait states it would insert are explicit:
// before the first MFMA (accumulator written by a VALU move) and after the last one (accumulator read by VALU code).
__device__ __forceinline__ void mfma_a(f32x4& acc, const bf16x8& w, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}
__device__ __forceinline__ void layer_group(f32x4 (&acc)[4], const bf16x8 (&w)[4][8], const bf16x8 (&bf)[8]) {
  asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) mfma_a(acc[t], w[t][s], bf[s]);
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}

// unit u of XCD x -> (edge block, batch element); workgroups / loop iterations with u >= n_units have nothing to do
struct TileWalk {
  int eb_start, n_units;
};
__device__ __forceinline__ TileWalk tile_walk(int xcd, int neb, int batch) {
  const int eb_base = neb / 8, eb_rem = neb % 8;
  TileWalk w;
  w.eb_start = xcd * eb_base + (xcd < eb_rem ? xcd : eb_rem);
  w.n_units = (eb_base + (xcd < eb_rem ? 1 : 0)) * batch;
  return w;
}

// Launch 1: one workgroup per tile (wave = 16-column group).  Eight lanes read one 128-byte line of a row (features
// 32 s .. 32 s + 31), so an instruction touches 8 full cache lines; lane piece p = lane & 7 holds features 32 s + 4 p .. + 3,
// which is half (p >> 2) of the B fragment of lane (j, q = p & 3): written there directly as 8 bytes of bf16.
__global__ __launch_bounds__(256) void edge16_gather_kernel(const Edge16Args a) {
  const int lane = threadIdx.x & 63;
  const int g = threadIdx.x >> 6;
  const int piece = lane & 7;
  const TileWalk tw = tile_walk(blockIdx.x & 7, a.neb, a.batch);
  const int u = blockIdx.x >> 3;
  if (u >= tw.n_units) return;
  const int eb = tw.eb_start + u / a.batch;
  const int b = u - (u / a.batch) * a.batch;
  char* out_g = a.h1g + ((size_t)(b * a.neb + eb) * kGroups + g) * 8192;
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // columns j = 8 h + (lane >> 3)
    const int j = 8 * h + (lane >> 3);
    const int kr = eb * kTileCols + 16 * g + j;
    const bool valid = kr < a.n_edges;
    const int k = valid ? kr : a.n_edges - 1;
    const float* rows[3] = {nullptr, nullptr, nullptr};
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (p < a.n_proj) {
        const int r = a.p_kind[p] == 0 ? ldgi(a.src + k) : (a.p_kind[p] == 1 ? ldgi(a.dst + k) : k);
        rows[p] = a.p_ptr[p] + ((size_t)b * (size_t)a.p_rows_pb[p] + (size_t)r) * (size_t)a.p_ld[p] + 4 * piece;
      }
    // all row pieces are requested before the first one is used: one round trip per wave, not one per K-step
    f32x4 z[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) z[s] = ldg4(a.b1 + 32 * s + 4 * piece);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (p < a.n_proj) {
        f32x4 v[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = ldg4(rows[p] + 32 * s);
#pragma unroll
        for (int s = 0; s < 8; ++s) z[s] += v[s];
      }
    const int q = piece & 3, half = piece >> 2;
    char* out = out_g + (16 * q + j) * 16 + half * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      bf16x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (__bf16)(valid ? fmaxf(z[s][r], 0.f) : 0.f);
      *(bf16x4*)(out + s * 1024) = v;
    }
  }
}

__global__ __launch_bounds__(256, 1) void edge16_kernel(const Edge16Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int q = lane >> 4;
  const int f0 = 64 * wave + 4 * q;  // this lane's features: f0 + 16 t + r

  // ---- resident weights: rows 64 wave .. +63 of both matrices, all 8 K-steps (packed stream: [s][16 tiles][lane][8]) ----
  bf16x8 wm[4][8], wo[4][8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const size_t off = ((size_t)(s * 16 + 4 * wave + t) * 64 + lane) * 16;
      wm[t][s] = *(const bf16x8*)(a.w_mid + off);
      wo[t][s] = *(const bf16x8*)(a.w_out + off);
    }
  // biases / LayerNorm parameters live in LDS (4 KiB) and are re-read where each phase needs them: with 256 registers of
  // weights, 64 more resident ones would push the fragment and residual prefetches into scratch
  {
    float* par_w = (float*)(lds + kOffPar);
    const int i = threadIdx.x;
    par_w[i] = a.b_mid[i];
    par_w[256 + i] = a.b_out[i];
    par_w[512 + i] = a.gamma[i];
    par_w[768 + i] = a.beta[i];
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

  // 32 KiB of layer-1 activations of unit u -> Hbuf1[par]: 32 LDS-DMA pieces of 1 KiB, 8 per wave (asynchronous).  The DMA
  // queue of a wave is shallow - eight back-to-back issues stall it for ~5 k cycles (measured) - so in steady state the
  // pieces are issued two per 16-column group of the output layer, ~1 k cycles apart, when no other load is outstanding.
  auto tile_src = [&](int u) -> const char* {
    const int eb = tw.eb_start + u / a.batch;
    const int b = u - (u / a.batch) * a.batch;
    return a.h1g + (size_t)(b * a.neb + eb) * kHBytes;
  };
  auto prefetch_piece = [&](const char* src, int par, int i) {
    const int piece = 8 * wave + i;
    glds16_asm_s((const float*)(src + piece * 1024), (unsigned)lane * 16u,
                 __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kOffH1 + par * kHBytes + piece * 1024)));
  };
  auto prefetch = [&](int u, int par) {
    const char* src = tile_src(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) prefetch_piece(src, par, i);
  };
  if (slot < tw.n_units) prefetch(slot, 0);

  int par = 0;
#pragma unroll 1
  for (int u = slot; u < tw.n_units; u += nslot) {
    const int eb = tw.eb_start + u / a.batch;
    const int b = u - (u / a.batch) * a.batch;
    const int k0 = eb * kTileCols;
    const char* h1 = lds + kOffH1 + par * kHBytes;

    unsigned long long ts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamp = a.dbg != nullptr && u == slot + 2 * nslot;
    if (stamp) ts[0] = gw_clock();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces (and its stores of the previous tile) are done
    if (stamp) ts[1] = gw_clock();
    wg_barrier();  // (1) Hbuf1[par] complete; every wave has left the previous tile's segment sums
    if (stamp) ts[2] = gw_clock();
    // next tile -> Hbuf1[par ^ 1] (last read before barrier (2) of the previous tile), one piece per group step below
    const bool more = u + nslot < tw.n_units;
    const char* nsrc = more ? tile_src(u + nslot) : nullptr;

    f32x4 bmv[4];  // (after barrier (1): on the first tile it also publishes the parameter block)
#pragma unroll
    for (int t = 0; t < 4; ++t) bmv[t] = *(const f32x4*)(par_l + 16 * t);
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
      gd_mine = kr < a.n_edges ? b * a.n_dst + ldgi(a.dst + kr) : -1;
    }

    // ---- residual rows: requested here, used after barrier (3) - their latency passes under the middle layer, and they
    // are back before the DMA pieces of the next tile are issued (the wave's memory queue is shallow) ----
    f32x4 resv[kGroups][4];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const float* rrow = a.res_ptr + ((size_t)b * (size_t)a.res_rows_pb + (size_t)kk[g]) * (size_t)a.res_ld + f0;
#pragma unroll
      for (int t = 0; t < 4; ++t) resv[g][t] = ldg4(rrow + 16 * t);
    }
    if (stamp) ts[3] = gw_clock();
    // ---- middle layer -> Hbuf2 (the next group's B fragments are read from LDS while this group's MFMAs run) ----
    bf16x8 bfr[2][8];
    load_frags(bfr[0], h1, lane);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      if (g + 1 < kGroups) load_frags(bfr[(g + 1) & 1], h1 + (g + 1) * 8 * 1024, lane);
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = bmv[t];
      layer_group(acc, wm, bfr[g & 1]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 h = f32x4{fmaxf(acc[t].x, 0.f), fmaxf(acc[t].y, 0.f), fmaxf(acc[t].z, 0.f), fmaxf(acc[t].w, 0.f)};
        *(bf16x4*)(h2 + ((g * 8 + 2 * wave + (t >> 1)) * 64 + lane) * 16 + (t & 1) * 8) = to_bf16x4(h);
      }
      if (stamp) ts[4 + g] = gw_clock();
    }
    wg_barrier();  // (2) Hbuf2 complete, Hbuf1 free
    if (stamp) ts[8] = gw_clock();

    f32x4 bov[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bov[t] = *(const f32x4*)(par_l + 256 + 16 * t);

    // ---- output layer + LayerNorm partial sums ----
    f32x4 o[kGroups][4];
    load_frags(bfr[0], h2, lane);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      if (g + 1 < kGroups) load_frags(bfr[(g + 1) & 1], h2 + (g + 1) * 8 * 1024, lane);
      if (more) {
        prefetch_piece(nsrc, par ^ 1, 2 * g);
        prefetch_piece(nsrc, par ^ 1, 2 * g + 1);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) o[g][t] = bov[t];
      layer_group(o[g], wo, bfr[g & 1]);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += o[g][t][r];
          s2 = fmaf(o[g][t][r], o[g][t][r], s2);
        }
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (q == 0) {
        lnp[(wave * kTileCols + 16 * g + j) * 2] = s1;
        lnp[(wave * kTileCols + 16 * g + j) * 2 + 1] = s2;
      }
    }
    f32x4 gmv[4], btv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      gmv[t] = *(const f32x4*)(par_l + 512 + 16 * t);
      btv[t] = *(const f32x4*)(par_l + 768 + 16 * t);
    }
    if (stamp) ts[9] = gw_clock();
    wg_barrier();  // (3) partial sums of all four feature quarters visible
    if (stamp) ts[10] = gw_clock();

    // ---- LayerNorm (eps 1e-5, biased variance), residual, staging ----
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const int col = 16 * g + j;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        s1 += lnp[(w4 * kTileCols + col) * 2];
        s2 += lnp[(w4 * kTileCols + col) * 2 + 1];
      }
      const float mean = s1 * (1.0f / 256.0f);
      const float var = fmaxf(s2 * (1.0f / 256.0f) - mean * mean, 0.f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
      float* srow = stage + col * kStageLd + f0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 rv = resv[g][t];
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (o[g][t][r] - mean) * rstd * gmv[t][r] + btv[t][r] + rv[r];
        *(f32x4*)(srow + 16 * t) = v;
        if (a.e_out != nullptr && valid[g]) stg4(a.e_out + ((size_t)b * a.n_edges + kk[g]) * 256 + f0 + 16 * t, v);
      }
    }
    if (wave == 0) gdl[lane] = gd_mine;
    if (stamp) ts[11] = gw_clock();
    wg_barrier();  // (4) staged tile + destination ids visible
    if (stamp) ts[12] = gw_clock();

    // ---- per-feature segment sums over the 64 destination-sorted columns ----
    // all 64 LDS reads first (independent), then a straight-line walk over registers.  Segment ends are the same for every
    // thread: lane i compares column i's destination with column i + 1's, the ballot is a 64-bit scalar mask, and the walk
    // tests one bit per column (a scalar branch that is rarely taken).
    {
      const int f = threadIdx.x;
      float vv[kTileCols];
#pragma unroll
      for (int i = 0; i < kTileCols; ++i) vv[i] = stage[i * kStageLd + f];
      const int gdv = gdl[lane];
      const int gdn = gdl[lane < kTileCols - 1 ? lane + 1 : lane];
      const unsigned long long ends = __ballot(lane == kTileCols - 1 || gdn != gdv);  // bit i: a segment ends with column i
      if (stamp) ts[13] = gw_clock();
      float run = 0.f;
      bool first = true;
#pragma unroll
      for (int i = 0; i < kTileCols; ++i) {
        run += vv[i];
        if (__builtin_expect((ends >> i) & 1ull, 0)) {
          const int cur = __builtin_amdgcn_readlane(gdv, i);
          if (cur >= 0 && GW_SKIP(a) != 1) {
            float* dstp = a.agg + (size_t)cur * 256 + f;
            // the first and the last segment of a tile may continue in the neighbouring tiles: atomics; the others are complete
            if (first || i == kTileCols - 1) __hip_atomic_fetch_add((GW_AS1 float*)dstp, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else stg1(dstp, run);
          }
          first = false;
          run = 0.f;
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

inline bool is_proj16(const gw_operand* o) { return o->k > 0 && o->projected != 0; }
inline bool is_raw16(const gw_operand* o) { return o->k > 0 && o->projected == 0; }

}  // namespace

namespace gw {

// Eligible: bf16 weights, one middle layer, every non-zero operand pre-projected (encoder / decoder / first processor block).
bool edge16_eligible(const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in, const gw_mlp_weights* w) {
  if (w->weight_dtype != GW_DTYPE_BF16 || w->n_mid != 1) return false;
  if (w->ln_width > 0 && w->ln_width != 256) return false;
  const gw_operand* ops[3] = {x_src, x_dst, e_in};
  int n_proj = 0;
  for (int i = 0; i < 3; ++i) {
    if (is_raw16(ops[i])) return false;
    n_proj += is_proj16(ops[i]) ? 1 : 0;
  }
  if (n_proj < 1) return false;
  static int impl = -1;  // GW_EDGE16_IMPL=0 forces the streaming kernel of gw_bf16.hip (A/B measurements, tests of both paths)
  if (impl < 0) impl = GW_TUNE("GW_EDGE16_IMPL", 1);
  return impl != 0;
}

size_t edge16_workspace_bytes(int32_t batch, int32_t n_edges) {
  return (size_t)batch * (size_t)((n_edges + kTileCols - 1) / kTileCols) * (size_t)kHBytes;
}

int edge16_launch(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst, const gw_operand* x_src,
                  const gw_operand* x_dst, const gw_operand* e_in, const gw_operand* e_res, const gw_mlp_weights* w,
                  float* e_out, float* agg, int32_t n_dst, void* workspace, void* stream) {
  Edge16Args a;
  memset(&a, 0, sizeof(a));
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
      ++a.n_proj;
    }
  a.b1 = w->b1;
  a.w_mid = (const char*)w->w_mid;
  a.b_mid = w->b_mid;
  a.w_out = (const char*)w->w_out;
  a.b_out = w->b_out;
  a.gamma = w->ln_gamma;
  a.beta = w->ln_beta;
  a.res_ptr = e_res->ptr;
  a.res_rows_pb = e_res->rows_per_batch;
  a.res_ld = e_res->ld;
  a.e_out = e_out;
  a.agg = agg;
  a.h1g = (char*)workspace;
  {
    static int skip = -1;
    if (skip < 0) skip = GW_TUNE("GW_EDGE16_SKIP", 0);
    a.skip = skip;
  }
  if (g_dbg != nullptr && g_dbg_kind == 3) {
    a.dbg = g_dbg;
    a.dbg_cap = g_dbg_cap;
  }
  static DeviceOnce once;
  if (once.first()) (void)hipFuncSetAttribute((const void*)edge16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
  // launch 1: one workgroup per tile, numbered like the persistent kernel walks them (XCD = workgroup & 7)
  const int units_max = (a.neb / 8 + (a.neb % 8 ? 1 : 0)) * batch;
  hipLaunchKernelGGL(edge16_gather_kernel, dim3((unsigned)(8 * units_max)), dim3(256), 0, (hipStream_t)stream, a);
  if (int rc = check_launch("edge16_gather_kernel launch")) return rc;
  static int n_wg = -1;  // persistent workgroups: one per CU, a multiple of 8 (XCD round-robin)
  if (n_wg < 0) n_wg = (GW_TUNE("GW_EDGE16_WGS", 256) + 7) / 8 * 8;
  hipLaunchKernelGGL(edge16_kernel, dim3((unsigned)n_wg), dim3(256), kLdsTotal, (hipStream_t)stream, a);
  return check_launch("edge16_kernel launch");
}

}  // namespace gw
