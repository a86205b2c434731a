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

// Each wave DMAs its share of `nfloats` (multiple of 256) from the packed weight stream into an LDS buffer.
__device__ __forceinline__ void issue_chunk(const float* __restrict__ g, int nfloats, float* ldsbuf, int lane, int wave) {
  const int npieces = nfloats >> 8;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ldsbuf;  // LDS byte address
  for (int p = wave; p < npieces; p += 4)
    glds16_asm_s(g + (size_t)p * 256, (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)p * 1024u));
}


}  // namespace gw

#endif  // GW_DEVICE_HPP
