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

// Each wave DMAs its share of `nfloats` (multiple of 256) from the packed weight stream into an LDS buffer.
__device__ __forceinline__ void issue_chunk(const float* __restrict__ g, int nfloats, float* ldsbuf, int lane, int wave) {
  const int npieces = nfloats >> 8;
  for (int p = wave; p < npieces; p += 4) glds16(g + (size_t)p * 256 + lane * 4, ldsbuf + p * 256);
}


}  // namespace gw

#endif  // GW_DEVICE_HPP
