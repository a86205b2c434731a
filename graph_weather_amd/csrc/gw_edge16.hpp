// gw_edge16.hpp - constants, launch arguments and device helpers shared by the bf16 edge-update kernels with on-chip weights
// (gw_edge16.hip: lock-step 4 / 8 wave kernels; gw_edge16t.hip: the team-pipelined kernel).  gfx950 only.
#ifndef GW_EDGE16_HPP
#define GW_EDGE16_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

namespace gw16 {
using namespace gw;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kTileCols = 64;                       // columns (edges) per tile: 4 groups of 16
constexpr int kGroups = 4;
constexpr int kHBytes = kGroups * 8 * 1024;         // one tile in the B-operand layout: 4 groups x 8 K-steps x 1 KiB
constexpr int kStageLd = 260;                       // staging row stride in floats (256 + 4: conflict-free column walks)
constexpr int kOffH1 = 0;                           // two layer-1 buffers (DMA prefetch of the next tile)
constexpr int kOffH2 = 2 * kHBytes;                 // layer-2 activations; the staging area reuses it (dead by then)
constexpr int kOffStage = kOffH2;
constexpr int kOffGd = kOffStage + kTileCols * kStageLd * 4;
constexpr int kOffLn = kOffGd + kTileCols * 4;
constexpr int kOffPar = kOffLn + 8 * kTileCols * 8;    // after [wave <= 8][column] (sum, sum of squares): b_mid, b_out, gamma, beta
constexpr int kLdsTotal = kOffPar + 5 * 256 * 4;       // ... and b1 (GATHER form)
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
  int p_half[3];  // the table is fp16 rows (GW_LAYOUT_ROWS_F16; p_ld counts halves): layer-1 kernel and team gather only
  const float* b1;
  const char* w_raw;  // packed W_e (layer-1 slice of the raw edge operand), edge16_l1_kernel only
  const char* w_mid;
  const float* b_mid;
  const char* w_out;
  const float* b_out;
  const float* gamma;
  const float* beta;
  // residual e: fp32 rows (res_ptr) or bf16 edge tiles (res_tiles); exactly one is set
  const float* res_ptr;
  int res_rows_pb;
  int res_ld;
  const char* res_tiles;
  int res_tiles_shared;  // the residual tiles are one set shared by the batch (cached edge embeddings of encoder / decoder / block 0)
  const char* e_tiles;   // raw edge operand of edge16_l1_kernel (== res_tiles in the forecaster)
  int e_tiles_shared;
  float* e_out;         // fp32 rows [batch * n_edges, 256] or null
  char* e_out_tiles;    // bf16 edge tiles or null
  float* agg;
  float* carry;  // deterministic segment sums: per-tile carry records (gw_internal.hpp), NULL = atomics; 4-wave kernel only
  char* h1g;  // workspace: layer-1 activations, [batch * neb tiles][4 groups][8 K-steps][64 lanes][8 bf16]
  int seg_tiles;            // GW_EDGE_SEGMENT_TILES: src / dst are the padded arrays of segment-aligned tiles (dst < 0 = padding
                            // column, no destination run crosses a multiple of 64): team kernel with the transposed output layer
  int agg_bf16k;            // ... and agg is bf16 rows in MFMA K order (GW_LAYOUT_ROWS_BF16K) instead of fp32 rows
  int seg_split;            // GW_EDGE_SEGMENT_SPLIT: a run longer than a tile continues over whole tiles - the first / last slot of a tile
                            // may be a PIECE of such a run: its sums are added with fp32 atomics (agg zero-filled by the caller)
  int bc, nchunk;           // team kernel (gw_edge16t.hip): batch elements of one edge block a workgroup handles in a row (divides
                            // batch) and chunks per edge block (batch / bc): per-edge data shared by the batch is fetched once per chunk
  int tune;                 // tuning builds only (GW_EDGE16_TUNE): A/B switches of the team kernel
  int skip;                 // tuning builds only (GW_EDGE16_SKIP): 1 = no aggregate writes
  unsigned long long* dbg;  // gw_debug_timestamps(kind 3): phase clocks of each workgroup's third tile
  int dbg_cap;
};

__device__ __forceinline__ void wg_barrier() { __syncthreads(); }

__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
  return r;
}
__device__ __forceinline__ bf16x8 to_bf16x8(f32x4 lo, f32x4 hi) {
  bf16x8 r;
  r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
  r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
  return r;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }

// x + (x of lane ^ 16) + (x of lane ^ 32) + (x of lane ^ 48): the sum over the four 16-lane rows, i.e. over the q index of
// the accumulator layout, on the gfx950 lane-swap instructions (v_permlane16_swap swaps the odd rows of its first operand
// with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second): two
// VALU operations per step instead of a ds_bpermute round trip through the LDS pipeline.
__device__ __forceinline__ float sum_rows(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned v = __float_as_uint(t);
  const auto r2 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

// B fragments of one 16-column group: 8 K-steps x 16 bytes per lane, all reads issued back to back
__device__ __forceinline__ void load_frags(bf16x8 (&bf)[8], const char* __restrict__ hbuf_g, int lane) {
#pragma unroll
  for (int s = 0; s < 8; ++s) bf[s] = *(const bf16x8*)(hbuf_g + s * 1024 + lane * 16);
}
// One resident layer on one 16-column group: acc[t] (RT row tiles of this wave) += W[tile t][K-step s] . B[s].
// The MFMAs are written as asm with the weight fragment constrained to an accumulation register ("a"): the weight
// registers then live in the AGPR half of the file for the whole kernel and feed the matrix cores from there.  Left to
// itself the allocator treats AGPRs as spill space and copies every fragment back to a VGPR before use (~700 copies per
// tile, measured).  Inline asm is opaque to the hazard recogniser, so the wait states it would insert are explicit:
// before the first MFMA (accumulator written by a VALU move) and after the last one (accumulator read by VALU code).
// An accumulator is touched by every 4th MFMA at most (RT = 2: even and odd K-steps accumulate separately and are added
// at the end), as in the round-1 kernel.
__device__ __forceinline__ void mfma_a(f32x4& acc, const bf16x8& w, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}
// The same instruction on fp16 operands (same rate, 11-bit significands): the middle layer of the segment-aligned team kernel,
// whose B operand - relu of a sum of fp16 product rows - is then made with packed fp16 arithmetic and no conversion.
// The first product of an accumulator chain with the bias as its C operand (acc = w . b + c): the accumulator is DEFINED here -
// no bias read from LDS per group, nothing to keep alive before its first MFMA (early clobber: the product reads its sources
// over several passes).
__device__ __forceinline__ void mfma_a_c(f32x4& acc, const bf16x8& w, const bf16x8& b, const f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(w), "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma_t_c(f32x4& acc, const bf16x8& b, const bf16x8& w, const f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(b), "a"(w), "v"(c));
}
__device__ __forceinline__ void mfma_a_f16_c(f32x4& acc, const bf16x8& w, const bf16x8& b, const f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(w), "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma_a_f16(f32x4& acc, const bf16x8& w, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}
// The transposed product: the activations fragment is the A operand (rows = the 16 edges of a group), the resident weight
// fragment the B operand (columns = 16 output features) - the same registers, the operands swapped - so the accumulator comes
// out as lane (feature, q) x 4 edges: D^T.  Used where the next contraction is over EDGES (segment sums on the matrix cores).
__device__ __forceinline__ void mfma_t(f32x4& acc, const bf16x8& b, const bf16x8& w) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "a"(w));
}
// x of the lane n places further round this lane's row of 16 (DPP row_ror: no LDS, no extra instruction once fused into the add)
// x + (x of the lane N places further round this lane's row of 16): one VALU instruction with a DPP operand (no LDS; written as
// asm because the builtin comes out as v_mov_b32_dpp into a zeroed register + the add).  A DPP operand must not have been written
// by one of the two preceding VALU instructions (asm is opaque to the hazard recogniser): callers keep producer and use apart.
// a * b + c as ONE scalar VALU instruction the SLP vectoriser cannot pair (v_pk_fma_f32 needs adjacent register pairs: it assembles
// them with moves, and a packed fp32 instruction beside MFMAs costs more than the two it replaces - MI355X_MICROARCH.md)
__device__ __forceinline__ float fma_s(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// Reduce-scatter of 8 per-lane sums over the 16 lanes of a DPP row in 16 instructions (an all-reduce of each takes 4 x 8):
//   step A  row_mirror       (l <-> 15 - l): lanes 0-7 keep v[0..3], lanes 8-15 keep v[4..7]        8 ops -> 4 registers
//   step B  row_half_mirror  (l <-> 7 - l within 8): banks 0 / 2 keep the first two, banks 1 / 3 the other two   4 ops -> 2
//   steps C, D  quad_perm [2,3,0,1], [1,0,3,2] on both                                                 4 ops
// bank_mask picks the 4-lane banks an instruction writes, so no select is needed.  Afterwards every lane of bank b holds the
// complete sums (v[2 (b & 1)], v[2 (b & 1) + 1]) of the first (b < 2) or second (b >= 2) four inputs.
// OP = 0 .. 15 in issue order; WAIT: two wait states in front (no instruction lies between producer and DPP use).
#define GW_DPP_ADD(ctrl, bank, dstc, dst, src)                                                                           \
  do {                                                                                                                   \
    if constexpr (WAIT) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:" bank : dstc(dst) : "v"(src)); \
    else asm volatile("v_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:" bank : dstc(dst) : "v"(src));          \
  } while (0)
template <int OP, bool WAIT>
__device__ __forceinline__ void row_reduce_scatter_op(const float (&lo)[4], const float (&hi)[4], float (&a)[4], float (&b)[2],
                                                      float (&c)[2], float (&d)[2]) {
  if constexpr (OP < 8) {  // step A: a[k] <- lo[k] (lanes 0-7), hi[k] (lanes 8-15)
    constexpr int k = OP >> 1;
    if constexpr ((OP & 1) == 0) GW_DPP_ADD("row_mirror", "0x3", "=v", a[k], lo[k]);
    else GW_DPP_ADD("row_mirror", "0xc", "+v", a[k], hi[k]);
  } else if constexpr (OP < 12) {  // step B: b[k] <- a[k] (banks 0, 2), a[k + 2] (banks 1, 3)
    constexpr int k = (OP - 8) >> 1;
    if constexpr ((OP & 1) == 0) GW_DPP_ADD("row_half_mirror", "0x5", "=v", b[k], a[k]);
    else GW_DPP_ADD("row_half_mirror", "0xa", "+v", b[k], a[k + 2]);
  } else if constexpr (OP < 14) {
    GW_DPP_ADD("quad_perm:[2,3,0,1]", "0xf", "=v", c[OP - 12], b[OP - 12]);
  } else {
    GW_DPP_ADD("quad_perm:[1,0,3,2]", "0xf", "=v", d[OP - 14], c[OP - 14]);
  }
}
#undef GW_DPP_ADD

template <int N, bool WAIT = false>
__device__ __forceinline__ float add_row_ror(float x) {
  float y;
  // (WAIT: the operand may have been produced by the instruction right in front - two wait states)
  if constexpr (N == 8 && WAIT) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  else if constexpr (N == 8) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  else if constexpr (N == 4) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  else if constexpr (N == 2) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  else asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x));
  return y;
}
__device__ __forceinline__ void layer_group(f32x4 (&acc)[4], const bf16x8 (&w)[4][8], const bf16x8 (&bf)[8]) {
  asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) mfma_a(acc[t], w[t][s], bf[s]);
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}
__device__ __forceinline__ void layer_group(f32x4 (&acc)[2], const bf16x8 (&w)[2][8], const bf16x8 (&bf)[8]) {
  f32x4 odd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(odd[0]), "+v"(odd[1]));
#pragma unroll
  for (int s = 0; s < 8; s += 2) {
    mfma_a(acc[0], w[0][s], bf[s]);
    mfma_a(acc[1], w[1][s], bf[s]);
    mfma_a(odd[0], w[0][s + 1], bf[s + 1]);
    mfma_a(odd[1], w[1][s + 1], bf[s + 1]);
  }
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(odd[0]), "+v"(odd[1]));
  acc[0] += odd[0];
  acc[1] += odd[1];
}
// The same layer with the B fragments read from LDS in two halves of 4 K-steps (16 fragment registers instead of 32: the
// 8-wave kernel has 128 VGPRs beside its 128 weight registers); the partner wave on the SIMD covers the second LDS latency.
__device__ __forceinline__ void layer_group_lds(f32x4 (&acc)[2], const bf16x8 (&w)[2][8], const char* __restrict__ hbuf_g, int lane) {
  f32x4 odd[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  bf16x8 bf[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) bf[s] = *(const bf16x8*)(hbuf_g + (4 * h + s) * 1024 + lane * 16);
    asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(odd[0]), "+v"(odd[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]));
#pragma unroll
    for (int s = 0; s < 4; s += 2) {
      mfma_a(acc[0], w[0][4 * h + s], bf[s]);
      mfma_a(acc[1], w[1][4 * h + s], bf[s]);
      mfma_a(odd[0], w[0][4 * h + s + 1], bf[s + 1]);
      mfma_a(odd[1], w[1][4 * h + s + 1], bf[s + 1]);
    }
    // the fragment registers are rewritten by the next half's LDS reads: keep those behind the MFMAs that read them
    asm volatile("s_nop 7" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]));
  }
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(odd[0]), "+v"(odd[1]));
  acc[0] += odd[0];
  acc[1] += odd[1];
}
__device__ __forceinline__ void layer_group_lds(f32x4 (&acc)[4], const bf16x8 (&w)[4][8], const char* __restrict__ hbuf_g, int lane) {
  bf16x8 bf[8];
  load_frags(bf, hbuf_g, lane);
  layer_group(acc, w, bf);
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

}  // namespace gw16

#endif  // GW_EDGE16_HPP
