// Micro-probe: cycles per v_mfma_f32_16x16x4_f32 in the access pattern of the chain kernels (16 independent
// accumulators per K-step, A fragments from LDS via ds_read_b128 one step ahead).  One wave per SIMD (256 threads, 1 WG/CU)
// or two (launch 2 WGs/CU).  build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0, 0, 0, 0};
  float b = 1.0f + lane * 1e-6f;
  const float* bl = lds + lane * 4;
  f32x4 a_cur[4];
  for (int b4 = 0; b4 < 4; ++b4) a_cur[b4] = *(const f32x4*)(bl + b4 * 256);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      f32x4 a_nxt[4];
      if (MODE >= 1) {
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) a_nxt[b4] = *(const f32x4*)(bl + ((s + 1) & 7) * 1024 + b4 * 256);
      }
      if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t >> 2][t & 3], b, acc[t], 0, 0, 0);
      if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 1) {
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) a_cur[b4] = a_nxt[b4];
      }
      if (MODE >= 3) b = fmaxf(b * 1.0001f, 0.f);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int t = 0; t < 16; ++t) s += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int wgs, int lds_bytes, float* out, unsigned long long* cyc) {
  const int iters = 200;
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), lds_bytes, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), lds_bytes, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double mf = 128.0 * iters;
  printf("%-34s wgs=%4d lds=%6d  cycles/mfma(memtime)=%.2f  wall: %.3f ms -> %.1f TF/s\n", name, wgs, lds_bytes, h[0] / mf, ms,
         (double)wgs * 4 * mf * 2 * 16 * 16 * 4 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
  for (int wgs : {256, 512}) {
    const int lds = wgs == 256 ? 100 * 1024 : 65536;
    run<0>("mfma only", wgs, lds, out, cyc);
    run<1>("+ds_read_b128 one step ahead", wgs, lds, out, cyc);
    run<2>("+sched_barrier", wgs, lds, out, cyc);
    run<3>("+valu on B", wgs, lds, out, cyc);
  }
  return 0;
}
