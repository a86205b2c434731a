// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// One wave: M0 = LDS byte address A, instruction offset K.  Prints where the 1 KiB piece landed.
//   hipcc --offload-arch=gfx950 -O2 -o glds_offset_probe glds_offset_probe.hip && ./glds_offset_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int K>
__global__ void probe(const float* g, float* out, unsigned m0_base) {
  __shared__ __attribute__((aligned(16))) float lds[4096];  // 16 KiB
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = -1.0f;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
  unsigned keep;
  const unsigned lane_off = threadIdx.x * 16u;
  const float* gu = g;  // wave-uniform base
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
      : "=&s"(keep)
      : "v"(lane_off), "s"(gu), "s"(__builtin_amdgcn_readfirstlane(lds0 + m0_base)), "n"(K)
      : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<float> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 8192 * 4);
  hipMalloc(&o, 4096 * 4);
  hipMemcpy(g, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  std::vector<float> r(4096);
  auto report = [&](const char* what) {
    hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1, n = 0;
    for (int i = 0; i < 4096; ++i)
      if (r[i] >= 0.f) { if (first < 0) first = i; ++n; }
    printf("%s: %d floats landed, first at LDS float %d (byte %d) holding global float %.0f (byte %.0f)\n", what, n, first, first * 4,
           first >= 0 ? r[first] : -1.f, first >= 0 ? r[first] * 4 : -1.f);
  };
  hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, g, o, 2048u);
  report("M0 = base + 2048, offset 0    ");
  hipLaunchKernelGGL(probe<1024>, dim3(1), dim3(64), 0, 0, g, o, 2048u);
  report("M0 = base + 2048, offset 1024 ");
  hipLaunchKernelGGL(probe<-1024>, dim3(1), dim3(64), 0, 0, g + 1024, o, 2048u);
  report("M0 = base + 2048, offset -1024 (global base + 4096 B)");
  hipLaunchKernelGGL(probe<4080>, dim3(1), dim3(64), 0, 0, g, o, 2048u);
  report("M0 = base + 2048, offset 4080 ");
  return 0;
}
