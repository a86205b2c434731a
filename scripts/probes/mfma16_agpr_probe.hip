// Micro-probe: v_mfma_f32_16x16x32_bf16 throughput with the weight operand in the AGPR half (as in the resident-weight kernels)
// vs plain VGPRs, 32 distinct weight fragments, fragments of the other operand from LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>  // 0: A = agpr weight, B = vgpr; 1: A = vgpr, B = agpr weight; 2: both vgpr (weights in VGPRs)
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters, const bf16x8* wsrc) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) ((float*)lds)[i] = 1e-3f * (i & 255);
  __syncthreads();
  bf16x8 w[4][8];
  for (int t = 0; t < 4; ++t)
    for (int s = 0; s < 8; ++s) w[t][s] = wsrc[(t * 8 + s) * 64 + lane];
  f32x4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 fr[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) fr[s] = *(const bf16x8*)(lds + (h * 4 + s) * 1024 + lane * 16);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int ks = m >> 2, t = m & 3;
        if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "a"(w[t][4 * h + ks]), "v"(fr[ks]));
        else if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(fr[ks]), "a"(w[t][4 * h + ks]));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(w[t][4 * h + ks]), "v"(fr[ks]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc, const bf16x8* w) {
  const int iters = 200;
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters, w);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters, w);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-40s %.1f cycles per MFMA\n", name, h[0] / (32.0 * iters));
}

int main() {
  float* out; unsigned long long* cyc; bf16x8* w;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 4096 * 8); hipMalloc(&w, 32 * 64 * 16); hipMemset(w, 0, 32 * 64 * 16);
  run<0>("weight = A operand in AGPRs", out, cyc, w);
  run<1>("weight = B operand in AGPRs", out, cyc, w);
  run<2>("weight in VGPRs", out, cyc, w);
  return 0;
}
