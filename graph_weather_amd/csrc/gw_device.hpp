// gw_device.hpp - device helpers shared by the kernels of libgw_amd.so (gfx950 only).
#ifndef GW_DEVICE_HPP
#define GW_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gw {


typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;         // 4 waves per workgroup; two workgroups per CU (2 waves per SIMD)
constexpr int kColsPerWave = 16;
constexpr int kColsPerWG = 64;
constexpr int kChunkSteps = 8;        // K-steps (4 k's each) per LDS buffer: K = 32 per chunk
constexpr int kLdsBufFloats = kChunkSteps * 4 * 256;  // 8 steps x 16 tiles x 16 rows x 4 k = 32 KiB
constexpr int kLdsBytes = 2 * kLdsBufFloats * 4;       // double buffered: 64 KiB per workgroup

#define GW_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ unsigned long long gw_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define GW_STAMP(i) \
  if (a.dbg != nullptr) { ts[i] = gw::gw_clock(); }
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const GW_AS1 f32x4*)p; }
__device__ __forceinline__ float ldg1(const float* p) { return *(const GW_AS1 float*)p; }
__device__ __forceinline__ int ldgi(const int* p) { return *(const GW_AS1 int*)p; }
__device__ __forceinline__ void stg4(float* p, f32x4 v) { *(GW_AS1 f32x4*)p = v; }
__device__ __forceinline__ void stg1(float* p, float v) { *(GW_AS1 float*)p = v; }

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The same LDS-DMA issued from an asm statement.  global_load_lds is a FLAT-class instruction: while hipcc has one
// pending in its s_waitcnt model, EVERY wait it generates is a full drain (vmcnt(0) / lgkmcnt(0)) - also the lgkmcnt
// in front of each MFMA group that consumes ds_read results, which then exposes the LDS latency of the reads issued
// for the NEXT step.  Hidden in asm, the DMA is counted by the kernel's own vmcnt waits and hipcc's ds_read waits are
// the normal counted ones.  lds_byte_addr: wave-uniform LDS byte address of this wave's 1 KiB piece (lane l lands at
// +16 l).  M0 is saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16_asm(const float* g, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_byte_addr)
      : "memory");
}

// Scalar-base form: g_uniform is a wave-uniform pointer (SGPR pair), lane_off_bytes each lane's byte offset (16 l).
__device__ __forceinline__ void glds16_asm_s(const float* g_uniform, unsigned lane_off_bytes, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off_bytes), "s"(g_uniform), "s"(lds_byte_addr)
      : "memory");
}

// L2 prefetch of a weight stream.  Workgroups of a mesh-sized launch start together and walk the same packed stream in step:
// every 32 KiB chunk is then a first touch for all of them at once, and with the other matrices of the forward (20+ MB) between
// two uses of a block's weights in a 4 MiB L2 it comes from the Infinity Cache / HBM - one miss latency per chunk, 48 chunks
// deep, where the chunk's MFMAs are shorter than the miss (bf16x3) nothing hides it.  One wave instruction touches one
// 128-byte line per lane (64 lines = 8 KiB); the 4 bytes per lane land in a scratch area of LDS (lds_byte_addr .. + 256), so no
// register is written and the request is counted by vmcnt like any DMA piece: issue before the first chunk barrier, into a
// region nothing reads before a later barrier.
__device__ __forceinline__ void l2_touch_lds(const char* lane_ptr, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_ptr), "s"(lds_byte_addr)
               : "memory");
}
// The workgroups of one XCD (blockIdx.x % 8 - observed placement, used for speed only) share the chunks of `n_mats` packed
// matrices of 8 x 32 KiB among themselves: chunk g is touched by the workgroup whose index within the XCD is g mod (workgroups
// per XCD).  Waves 0-3 issue one instruction per chunk each; chunk 0 of the first matrix is being fetched by everybody anyway.
__device__ __forceinline__ void l2_prefetch_stream(const char* const* mats, int n_mats, int lane, int wave, unsigned scratch_lds) {
  if (wave >= 4) return;
  const int per_xcd = (int)(gridDim.x >> 3) > 0 ? (int)(gridDim.x >> 3) : 1;
  const int me = (int)(blockIdx.x >> 3) % per_xcd;
  for (int mi = 0; mi < n_mats; ++mi) {
    const char* p = mats[mi];
    if (p == nullptr) continue;
    for (int c = (mi == 0 ? 1 : 0); c < 8; ++c)
      if ((mi * 8 + c) % per_xcd == me)
        l2_touch_lds(p + (size_t)c * 32768 + (size_t)(wave * 64 + lane) * 128, __builtin_amdgcn_readfirstlane(scratch_lds + (unsigned)wave * 256u));
  }
}

// ---- LayerNorm backward inside the fused-MLP kernels (the input-gradient chains of csrc/gw_kernels.hip and csrc/gw_split.hip) ----
// sums over the 16 lanes of a DPP row (lanes 16 q .. 16 q + 15 = the 16 table rows of a wave at one q) for four values at once;
// every lane gets the totals.  v_add_f32_dpp from an asm statement (hipcc emits v_mov_b32_dpp + v_add_f32 for the builtin): one
// VALU instruction per step and value.  A DPP operand written by the previous VALU instruction needs two wait states: the four
// values alternate, so inside the block every read is three instructions behind its write; s_nop 1 covers the block's entry.
__device__ __forceinline__ void row16_sum4(f32x4& v) {
  float a = v.x, b = v.y, c = v.z, d = v.w;
#define GW_DPP4(ctrl)                                                 \
  "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t" GW_DPP4("quad_perm:[1,0,3,2]") GW_DPP4("quad_perm:[2,3,0,1]") GW_DPP4("row_half_mirror") GW_DPP4("row_mirror")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef GW_DPP4
  v = f32x4{a, b, c, d};
}
// One table row per lane group: lane (j = lane & 15: row of the wave, q = lane >> 4) holds columns 16 t + 4 q + r of its row, the
// layout of the kernels' accumulators and operands.  dn: gradient at the output of the LayerNorm (width 256, eps 1e-5, biased
// variance), y: the saved pre-norm row; g receives the gradient at the norm's input.  Row sums are two shuffles over q (as in the
// forward's LayerNorm); the column sums of d gamma = dn * xhat and d beta = dn over the wave's 16 rows go through DPP row sums into
// red[0..255 | 256..511] (LDS, this wave's 512 floats) - rows past the end of the table (valid == false: copies of the last row)
// add nothing.  After a workgroup barrier ln_backward_flush adds the four waves' sums to dgamma / dbeta (one atomic per column).
// drow2 (wave-uniformly NULL or not): a second row added to dn (the gradient of e' that arrives beside the gathered aggregate's).
__device__ __forceinline__ void ln_backward_rows16(f32x4 (&g)[16], const float* __restrict__ yrow, const float* __restrict__ drow,
                                                   const float* __restrict__ drow2, const float* __restrict__ gamma, bool valid, int q,
                                                   int j, float* red) {
  f32x4 yv[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    yv[t] = ldg4(yrow + 16 * t + 4 * q);
    g[t] = ldg4(drow + 16 * t + 4 * q);
  }
  if (drow2 != nullptr) {
#pragma unroll
    for (int t = 0; t < 16; ++t) g[t] += ldg4(drow2 + 16 * t + 4 * q);
  }
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) sum += (yv[t].x + yv[t].y) + (yv[t].z + yv[t].w);
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float mean = sum * (1.0f / 256.0f);
  float var = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    yv[t] = yv[t] - mean;
    var += (yv[t].x * yv[t].x + yv[t].y * yv[t].y) + (yv[t].z * yv[t].z + yv[t].w * yv[t].w);
  }
  var += __shfl_xor(var, 16);
  var += __shfl_xor(var, 32);
  const float rstd = 1.0f / sqrtf(var * (1.0f / 256.0f) + 1e-5f);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int col = 16 * t + 4 * q;
    const f32x4 xh = yv[t] * rstd;
    yv[t] = xh;
    const f32x4 dv = valid ? g[t] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pg = dv * xh, pb = dv;
    row16_sum4(pg);
    row16_sum4(pb);
    if (j == 0) {
      *(f32x4*)(red + col) = pg;
      *(f32x4*)(red + 256 + col) = pb;
    }
    const f32x4 gg = g[t] * ldg4(gamma + col);
    g[t] = gg;
    sg += (gg.x + gg.y) + (gg.z + gg.w);
    sgx += (gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w);
  }
  sg += __shfl_xor(sg, 16);
  sg += __shfl_xor(sg, 32);
  sgx += __shfl_xor(sgx, 16);
  sgx += __shfl_xor(sgx, 32);
  const float mg = sg * (1.0f / 256.0f), mgx = sgx * (1.0f / 256.0f);
#pragma unroll
  for (int t = 0; t < 16; ++t) g[t] = (g[t] - mg - yv[t] * mgx) * rstd;
}
// red_all: [4 waves][d gamma | d beta][256] floats of LDS; tid = threadIdx.x of a 256-thread workgroup
__device__ __forceinline__ void ln_backward_flush(const float* red_all, float* dgamma, float* dbeta, int tid) {
  const float dg = (red_all[tid] + red_all[512 + tid]) + (red_all[1024 + tid] + red_all[1536 + tid]);
  const float db = (red_all[256 + tid] + red_all[768 + tid]) + (red_all[1280 + tid] + red_all[1792 + tid]);
  __hip_atomic_fetch_add((GW_AS1 float*)(dgamma + tid), dg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add((GW_AS1 float*)(dbeta + tid), db, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Column sums of a row-per-lane-group tile (same layout) over the workgroup's 64 rows, added to out[256]: red_all = [4 waves][256]
// floats of LDS; `barrier` is the caller's workgroup barrier.
template <typename Barrier>
__device__ __forceinline__ void colsum_rows64(const f32x4 (&v)[16], bool valid, int q, int j, int wave, int tid, float* red_all,
                                              float* out, Barrier barrier) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    f32x4 s = valid ? v[t] : f32x4{0.f, 0.f, 0.f, 0.f};
    row16_sum4(s);
    if (j == 0) *(f32x4*)(red_all + wave * 256 + 16 * t + 4 * q) = s;
  }
  barrier();
  __hip_atomic_fetch_add((GW_AS1 float*)(out + tid), (red_all[tid] + red_all[256 + tid]) + (red_all[512 + tid] + red_all[768 + tid]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Each wave DMAs its share of `nfloats` (multiple of 256) from the packed weight stream into an LDS buffer.
__device__ __forceinline__ void issue_chunk(const float* __restrict__ g, int nfloats, float* ldsbuf, int lane, int wave) {
  const int npieces = nfloats >> 8;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf;  // LDS byte address
  for (int p = wave; p < npieces; p += 4)
    glds16_asm_s(g + (size_t)p * 256, (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)p * 1024u));
}


}  // namespace gw

#endif  // GW_DEVICE_HPP
