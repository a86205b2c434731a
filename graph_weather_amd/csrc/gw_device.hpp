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

// Each wave DMAs its share of `nfloats` (multiple of 256) from the packed weight stream into an LDS buffer.
__device__ __forceinline__ void issue_chunk(const float* __restrict__ g, int nfloats, float* ldsbuf, int lane, int wave) {
  const int npieces = nfloats >> 8;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf;  // LDS byte address
  for (int p = wave; p < npieces; p += 4)
    glds16_asm_s(g + (size_t)p * 256, (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)p * 1024u));
}


}  // namespace gw

#endif  // GW_DEVICE_HPP
