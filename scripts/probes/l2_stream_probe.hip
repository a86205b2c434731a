// How fast can every CU pull the same 512 KB (two packed 256 x 256 bf16x3 matrices) out of L2 into LDS, nothing else going on?
// The split edge kernels do exactly that once per 64-column tile (gw_split.hip); this is the roof of that stream.
//   hipcc --offload-arch=gfx950 -O2 -o l2_stream_probe l2_stream_probe.hip && ./l2_stream_probe
// Workgroups of 4 waves, double-buffered 32 KiB chunks by global_load_lds_dwordx4 (a wave's 8 pieces of a chunk in one statement,
// immediate offsets), vmcnt(0) + barrier per chunk as in pass_x3; WGS_PER_CU = 1 or 2 (64 KiB of LDS each).
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void burst8(const char* g_mid, unsigned lane_off, unsigned lds_mid) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:-4096\n\tglobal_load_lds_dwordx4 %1, %2 offset:-3072\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:-2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:-1024\n\t"
      "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off), "s"(g_mid), "s"(lds_mid)
      : "memory");
}

__global__ __launch_bounds__(256, 2) void stream(const char* w, int nchunks_total, int tiles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  int parity = 0;
  for (int t = 0; t < tiles; ++t) {
    for (int c = 0; c < nchunks_total; ++c) {
      burst8(w + (size_t)c * 32768 + wave * 8192 + 4096, lane * 16u, __builtin_amdgcn_readfirstlane(lds0 + parity * 32768u + wave * 8192u + 4096u));
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the previous chunk of this wave has landed
      asm volatile("s_barrier" ::: "memory");
      parity ^= 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink != nullptr) sink[blockIdx.x * 256 + threadIdx.x] = ((float*)lds)[threadIdx.x];
}

int main() {
  const size_t bytes = 512 * 1024;
  char* w;
  float* sink;
  hipMalloc(&w, bytes);
  hipMemset(w, 1, bytes);
  hipMalloc(&sink, 4096 * 256 * 4);
  hipFuncSetAttribute((const void*)stream, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    for (int cus = 64; cus <= 256; cus *= 2) {
      const int grid = cus * wgs_per_cu, tiles = 40, nch = (int)(bytes / 32768);
      hipLaunchKernelGGL(stream, dim3(grid), dim3(256), 65536, 0, w, nch, 2, sink);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(stream, dim3(grid), dim3(256), 65536, 0, w, nch, tiles, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double total = (double)grid * tiles * (double)bytes;
      printf("%d workgroup(s) per CU x %3d workgroups-worth of CUs (grid %3d): %.3f ms, %.2f TB/s L2 -> LDS, %.1f B/clk per workgroup at 2.4 GHz, "
             "%.0f cycles per 256 KB pass\n", wgs_per_cu, cus, grid, ms, total / (ms * 1e-3) / 1e12, total / grid / (ms * 1e-3 * 2.4e9),
             (ms * 1e-3 * 2.4e9) / (tiles * 2.0));
    }
  }
  return 0;
}
