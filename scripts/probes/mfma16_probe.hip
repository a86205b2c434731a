// Micro-probe: cycles per v_mfma_f32_16x16x32_bf16 as a function of how many independent accumulators rotate (the team kernel's
// layers reuse an accumulator every 4th MFMA).  build: hipcc --offload-arch=gfx950 -O3 mfma16_probe.hip -o mfma16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int FILL>
__global__ __launch_bounds__(512, 2) void probe(float* out, unsigned long long* cyc, int iters, int waves_active) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  f32x4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1e-3f * (lane + i)); b[i] = (__bf16)(1.0f + 1e-3f * i); }
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  unsigned long long t0 = 0, t1 = 0;
  if (wave < waves_active) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
#pragma unroll
        for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(m + f) & 7]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  }
  float s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int FILL>
void run(int waves, float* out, unsigned long long* cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((probe<NACC, FILL>), dim3(256), dim3(512), 0, 0, out, cyc, iters, waves);
  hipLaunchKernelGGL((probe<NACC, FILL>), dim3(256), dim3(512), 0, 0, out, cyc, iters, waves);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("accumulators %2d  fillers/mfma %d  active waves/CU %d : %.1f cycles per MFMA (wave 0)\n", NACC, FILL, waves, h[0] / (32.0 * iters));
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 4096 * 8);
  for (int waves : {4, 8}) {
    run<1, 0>(waves, out, cyc); run<2, 0>(waves, out, cyc); run<4, 0>(waves, out, cyc); run<8, 0>(waves, out, cyc); run<16, 0>(waves, out, cyc);
    run<4, 1>(waves, out, cyc); run<4, 2>(waves, out, cyc); run<4, 3>(waves, out, cyc); run<8, 1>(waves, out, cyc); run<8, 2>(waves, out, cyc);
    run<8, 3>(waves, out, cyc); run<8, 4>(waves, out, cyc);
  }
  return 0;
}
