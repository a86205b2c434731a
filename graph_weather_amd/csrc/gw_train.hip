// gw_train.hip - building blocks of the backward pass / training step of the hot path (SURVEY.md section 8f row 1,
// appendix G): what autograd launches through ATen + torch_scatter in the reference (train/run.py:509-521:
// loss.backward(); optimizer.step()).  First, correctness-oriented version: activations are materialised in HBM by the
// forward kernels ("save" pointers) and the backward is composed from the generic kernels below.
//
//   gw_gemm_f32            fp32-MFMA GEMM with bounds checks: NN (input gradients  dX = dZ . W) and
//                          TN (weight gradients dW += dZ^T . X, split over row slabs, accumulated with atomics)
//   gw_relu_backward       dZ = dH * (H > 0) and bias gradient db += column sums of dZ        (nn.ReLU / nn.Linear.bias)
//   gw_layernorm_backward  dY, dgamma +=, dbeta += from dN and the saved pre-norm rows        (nn.LayerNorm, eps 1e-5)
//   gw_gather_rows         out[b, k] = table[b, idx[k]] (+ add[b, k])     dual of the segment sum (graph_net_block.py:188)
//   gw_segment_sum_rows    out[b, n] (+)= sum_{i in seg(n)} rows[b, perm[i]]      dual of the x[row] / x[col] gathers (:221-228)
//   gw_normalized_mse_backward, gw_adamw_step
// All fp32; matrix products are fmaf chains on v_mfma_f32_16x16x4_f32 like the forward.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

// ---- GEMM ------------------------------------------------------------------------------------------------------------
// Block = 4 waves, block tile 64 (m) x 64 (n); wave w owns rows 16w..16w+15 of the tile, 4 MFMA tiles along n.
// TN: C[m][n] += sum_k A[k][m] * B[k][n]   (k = row index of both operands; blockIdx.z = slab of k; atomics)
// NN: C[m][n]  = sum_k A[m][k] * B[k][n]
template <bool TN>
__global__ __launch_bounds__(256) void gemm_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                   const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                   int k_slab) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 64 + wave * 16;
  const int n0 = blockIdx.y * 64;
  const int k_begin = TN ? blockIdx.z * k_slab : 0;
  const int k_end = TN ? (k_begin + k_slab < K ? k_begin + k_slab : K) : K;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int m = m0 + i;
  const bool m_ok = m < M;
#pragma unroll 4
  for (int k0 = k_begin; k0 < k_end; k0 += 4) {
    const int k = k0 + kq;
    const bool k_ok = k < k_end;
    float a = 0.f;
    if (m_ok && k_ok) a = TN ? ldg1(A + (size_t)k * lda + m) : ldg1(A + (size_t)m * lda + k);
    float b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = n0 + 16 * t + i;
      b[t] = (k_ok && n < N) ? ldg1(B + (size_t)k * ldb + n) : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
  }
  // D layout: column (n) = lane & 15, rows (m) = 4 * (lane >> 4) + r
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 16 * t + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mm = m0 + 4 * kq + r;
      if (mm < M && n < N) {
        float* p = C + (size_t)mm * ldc + n;
        if (TN) __hip_atomic_fetch_add((GW_AS1 float*)p, acc[t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else stg1(p, acc[t][r]);
      }
    }
  }
}

// TN with more register reuse: block tile 128 (m) x 128 (n), wave (wm, wn) owns 64 x 64 = 4 x 4 MFMA tiles, so one
// K-step (4 rows of A and B) costs 8 dword loads per lane for 16 MFMAs.
__global__ __launch_bounds__(256) void gemm_tn_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                      const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                      int k_slab, float* __restrict__ colsum_a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 128 + (wave >> 1) * 64;
  const int n0 = blockIdx.y * 128 + (wave & 1) * 64;
  const int k_begin = blockIdx.z * k_slab;
  const int k_end = k_begin + k_slab < K ? k_begin + k_slab : K;
  f32x4 acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool m_ok[4], n_ok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    m_ok[t] = m0 + 16 * t + i < M;
    n_ok[t] = n0 + 16 * t + i < N;
  }
  // column sums of A (the bias gradient when A = dZ): taken once per m, by the waves of the first n-block column
  const bool do_sum = colsum_a != nullptr && blockIdx.y == 0 && (wave & 1) == 0;
  float asum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int k0 = k_begin; k0 < k_end; k0 += 4) {
    const int k = k0 + kq;
    const bool k_ok = k < k_end;
    const float* arow = A + (size_t)k * lda + m0 + i;
    const float* brow = B + (size_t)k * ldb + n0 + i;
    float a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t] = (k_ok && m_ok[t]) ? ldg1(arow + 16 * t) : 0.f;
      b[t] = (k_ok && n_ok[t]) ? ldg1(brow + 16 * t) : 0.f;
      asum[t] += a[t];
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  }
  if (do_sum) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = asum[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0 && m_ok[t])
        __hip_atomic_fetch_add((GW_AS1 float*)(colsum_a + m0 + 16 * t + i), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + 16 * tn + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = m0 + 16 * tm + 4 * kq + r;
        if (mm < M && n < N)
          __hip_atomic_fetch_add((GW_AS1 float*)(C + (size_t)mm * ldc + n), acc[tm][tn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
}

// TN for the common aligned case (M, N multiples of 128; lda, ldb multiples of 4): both operand tiles are staged through
// LDS with coalesced 16-byte loads (double buffered, 16 rows of k per stage), MFMA operands come from LDS.
constexpr int kTnKC = 16;     // k rows per stage
constexpr int kTnLd = 144;    // LDS row stride in floats (128 + 16: the 4 k-rows of an MFMA step hit disjoint banks)
__global__ __launch_bounds__(256) void gemm_tn_lds_kernel(int K, const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                          int k_slab, float* __restrict__ colsum_a) {
  __shared__ __attribute__((aligned(16))) float sm[2][2][kTnKC * kTnLd];  // [stage][A|B]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int mB = blockIdx.x * 128, nB = blockIdx.y * 128;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int k_begin = blockIdx.z * k_slab;
  const int k_end = k_begin + k_slab < K ? k_begin + k_slab : K;
  const int lrow = threadIdx.x >> 5;        // 0..7 (+8): k row inside a stage
  const int lcol = (threadIdx.x & 31) * 4;  // float offset inside the 128-wide tile
  f32x4 acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_sum = colsum_a != nullptr && blockIdx.y == 0 && (wave & 1) == 0;
  float asum[4] = {0.f, 0.f, 0.f, 0.f};

  f32x4 ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + lrow + 8 * h;
      if (k < k_end) {
        ra[h] = ldg4(A + (size_t)k * lda + mB + lcol);
        rb[h] = ldg4(B + (size_t)k * ldb + nB + lcol);
      } else {
        ra[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        rb[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto stash = [&](int st) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *(f32x4*)(&sm[st][0][(lrow + 8 * h) * kTnLd + lcol]) = ra[h];
      *(f32x4*)(&sm[st][1][(lrow + 8 * h) * kTnLd + lcol]) = rb[h];
    }
  };
  fetch(k_begin);
  stash(0);
  __syncthreads();
  int st = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += kTnKC) {
    const bool more = k0 + kTnKC < k_end;
    if (more) fetch(k0 + kTnKC);  // global loads of the next stage fly under this stage's MFMAs
    const float* as = &sm[st][0][kq * kTnLd + wm + i];
    const float* bs = &sm[st][1][kq * kTnLd + wn + i];
#pragma unroll
    for (int ks = 0; ks < kTnKC / 4; ++ks) {
      float a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = as[ks * 4 * kTnLd + 16 * t];
        b[t] = bs[ks * 4 * kTnLd + 16 * t];
        asum[t] += a[t];
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    if (more) stash(st ^ 1);
    __syncthreads();
    st ^= 1;
  }
  if (do_sum) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = asum[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) __hip_atomic_fetch_add((GW_AS1 float*)(colsum_a + mB + wm + 16 * t + i), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = nB + wn + 16 * tn + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = mB + wm + 16 * tm + 4 * kq + r;
        __hip_atomic_fetch_add((GW_AS1 float*)(C + (size_t)mm * ldc + n), acc[tm][tn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
}

// TN on split operands (GW_GEMM_TN_BF16X3; mixed-precision training): the same tile and slab scheme, but
// C += A_hi^T B_hi + A_hi^T B_lo + A_lo^T B_hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate; v = hi + lo, hi = bf16(v),
// lo = bf16(v - hi): 16 significant bits per operand).  Both operands are k-strided in memory ([rows][features]) while an MFMA
// fragment is 8 consecutive k of one column, so a stage (32 rows) is split ONCE, by the thread that loaded it, and written
// to LDS transposed as bf16 planes [A_hi | A_lo | B_hi | B_lo][column][32 k]: thread (r = tid >> 5, c = tid & 31) loads rows
// r + 8 h (h = 0..3) of columns c + 32 j (j = 0..3; 128 contiguous bytes per half wave and row) and stores, per column, the four
// k's it holds as one 8-byte word at k slot 4 r + h - the same permutation of the stage's rows for A and B, which is all the
// product needs.  A lane's fragment is then 16 contiguous bytes (slots 8 (lane >> 4) ..+7) of its column: 16 LDS reads and no
// conversion per stage and wave for 48 MFMAs (round 5 kept fp32 tiles in LDS and split in every wave that read them: 64 reads,
// 64 splits and ~200 register moves per stage - the loop was VALU bound at twice the MFMA time).  Column stride 72 bytes: the
// 8-byte writes of 32 consecutive columns and the fragment reads of 16 consecutive columns spread over all banks.
// Global loads run TWO stages ahead of the MFMAs in registers (a stage is ~0.4 us of arithmetic, an HBM round trip under load
// 2 us; with the loads of one stage in flight the kernel sat at 2.6 TB/s on the decoder's 905 k rows), and the barriers wait
// for LDS only, so the loads stay in flight across them.
// (Measured and not kept in round 5: a 256 x 256 tile per 8-wave workgroup, every operand row read once - 531 vs 602 us at the
// decoder's 905 k rows, but 125 vs 90 us at a processor block's 82 k and 37 vs 26 us at 11.7 k: one workgroup per CU, 256 KiB of
// atomics per row slab; the training step went 28.9 -> 31.0 ms.)
constexpr int kTx3KC = 32;                    // k rows per stage (one bf16 MFMA K-step)
constexpr int kTx3ColB = 72;                  // bytes per column in a plane: 32 k x 2 + 8
constexpr int kTx3Plane = 128 * kTx3ColB;     // one bf16 plane of a stage
constexpr int kTx3Stage = 4 * kTx3Plane;      // A_hi | A_lo | B_hi | B_lo = 36 864 bytes
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// (hi, lo) of two values: one packed conversion each way; the hi halves come back as floats by a shift / a mask of the pair
__device__ __forceinline__ void split2_tn(float v0, float v1, unsigned& h, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v0, v1}, bf16x2_t));
  const float h0 = __uint_as_float(h << 16), h1 = __uint_as_float(h & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v0 - h0, v1 - h1}, bf16x2_t));
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8_t lds_frag8(const unsigned char* p) {  // 8-byte aligned (column stride 72)
  const bf16x4_t a = *(const bf16x4_t*)p, b = *(const bf16x4_t*)(p + 8);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void lds_barrier_tn() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// RAGGED: m or n is not a multiple of 128 (the 102 input / 78 output features of the node encoder's first and the head's last
// Linear): columns past the edge are read from the last real column and zeroed with the rows past k_end; nothing is stored for them.
template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                            const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                            int k_slab, float* __restrict__ colsum_a, int tiles_m, int tiles, int tune) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];  // [stage 2][plane 4][column 128][72 bytes]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  // One-dimensional grid, XCD aware: workgroup L runs on XCD L % 8 (observed placement, used for speed only), and the tiles of
  // one row slab read the same operand rows - they are given to consecutive workgroups of ONE XCD, so the second, third and
  // fourth reader of a row find it in that XCD's L2 instead of fetching it again over the fabric.
  const int xl = (int)blockIdx.x >> 3;
  const bool plain = (tune & 2) != 0;  // (tuning builds: launch order, the tiles of a slab on different XCDs)
  const int tile = plain ? (int)blockIdx.x % tiles : xl % tiles;
  const int slab = plain ? (int)blockIdx.x / tiles : (xl / tiles) * 8 + ((int)blockIdx.x & 7);
  const int k_begin = slab * k_slab;
  if (k_begin >= K) return;
  const int bx = tile % tiles_m, by = tile / tiles_m;
  const int mB = bx * 128, nB = by * 128;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int k_end = k_begin + k_slab < K ? k_begin + k_slab : K;
  const int lrow = threadIdx.x >> 5;  // 0..7 (+ 8 h): k row inside a stage
  const int lc = threadIdx.x & 31;    // (+ 32 j): column inside the 128-wide tile
  f32x4 acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  // column sums of A (the bias gradient when A = dZ): taken once per m, by the workgroups of the first n-block column
  const bool do_sum = colsum_a != nullptr && by == 0;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  float r0[32], r1[32];  // two stages in flight: [A | B][h][j]
  // rows past k_end (the ragged last stage of the matrix, and the one or two stages fetched past the end of the slab) are read
  // from row k_end - 1 - no branch, no register merge - and zeroed when the stage is split
  int ca[4], cb[4];       // RAGGED: column offsets from (mB + lc) / (nB + lc), clamped to the last real column
  bool aok[4], bok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    aok[j] = !RAGGED || mB + lc + 32 * j < M;
    bok[j] = !RAGGED || nB + lc + 32 * j < N;
    ca[j] = aok[j] ? 32 * j : M - 1 - mB - lc;
    cb[j] = bok[j] ? 32 * j : N - 1 - nB - lc;
  }
  auto fetch = [&](int k0, float (&r)[32]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      int k = k0 + lrow + 8 * h;
      k = k < k_end ? k : k_end - 1;
      const float* pa = A + (size_t)k * lda + mB + lc;
      const float* pb = B + (size_t)k * ldb + nB + lc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[4 * h + j] = ldg1(pa + (RAGGED ? ca[j] : 32 * j));
        r[16 + 4 * h + j] = ldg1(pb + (RAGGED ? cb[j] : 32 * j));
      }
    }
  };
  auto stash = [&](int st, int k0, const float (&r)[32]) {
    unsigned char* s = smx + st * kTx3Stage + lrow * 8;
    bool ok[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) ok[h] = k0 + lrow + 8 * h < k_end;
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cok = !RAGGED || (op == 0 ? aok[j] : bok[j]);
        const float v0 = (ok[0] && cok) ? r[16 * op + j] : 0.f, v1 = (ok[1] && cok) ? r[16 * op + 4 + j] : 0.f;
        const float v2 = (ok[2] && cok) ? r[16 * op + 8 + j] : 0.f, v3 = (ok[3] && cok) ? r[16 * op + 12 + j] : 0.f;
        if (op == 0) csum[j] += (v0 + v1) + (v2 + v3);
        unsigned h01, l01, h23, l23;
        split2_tn(v0, v1, h01, l01);
        split2_tn(v2, v3, h23, l23);
        unsigned char* d = s + (2 * op) * kTx3Plane + (lc + 32 * j) * kTx3ColB;
        *(u32x2_t*)d = u32x2_t{h01, h23};
        *(u32x2_t*)(d + kTx3Plane) = u32x2_t{l01, l23};
      }
  };
  auto compute = [&](int st) {
    const unsigned char* s = smx + st * kTx3Stage + kq * 16;
    bf16x8_t ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned char* pa = s + (wm + 16 * t + i) * kTx3ColB;
      const unsigned char* pb = s + 2 * kTx3Plane + (wn + 16 * t + i) * kTx3ColB;
      ah[t] = lds_frag8(pa);
      al[t] = lds_frag8(pa + kTx3Plane);
      bh[t] = lds_frag8(pb);
      bl[t] = lds_frag8(pb + kTx3Plane);
    }
    if (tune & 4) {  // (tuning builds: the loop without its MFMAs - the fragments are still read)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][0][0] += (float)ah[t][0] + (float)al[t][1] + (float)bh[t][2] + (float)bl[t][3];
      return;
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
  };
  fetch(k_begin, r0);
  stash(0, k_begin, r0);
  fetch(k_begin + kTx3KC, r0);
  fetch(k_begin + 2 * kTx3KC, r1);
  lds_barrier_tn();
  for (int k0 = k_begin;; k0 += 2 * kTx3KC) {
    compute(0);
    if (k0 + kTx3KC >= k_end) break;
    stash(1, k0 + kTx3KC, r0);  // (stage 1 was last read before the barrier that closed the previous round)
    fetch(k0 + 3 * kTx3KC, r0);
    lds_barrier_tn();
    compute(1);
    if (k0 + 2 * kTx3KC >= k_end) break;
    stash(0, k0 + 2 * kTx3KC, r1);
    fetch(k0 + 4 * kTx3KC, r1);
    lds_barrier_tn();
  }
  if (do_sum) {  // (uniform over the workgroup) the eight threads of a column add up in LDS: one atomic per column and workgroup
    lds_barrier_tn();
    float* cs = (float*)smx;
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[lrow * 128 + lc + 32 * j] = csum[j];
    lds_barrier_tn();
    if (threadIdx.x < 128 && (!RAGGED || mB + (int)threadIdx.x < M)) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += cs[r * 128 + threadIdx.x];
      __hip_atomic_fetch_add((GW_AS1 float*)(colsum_a + mB + threadIdx.x), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tune & 1) return;  // (tuning builds: the cost of the atomics)
  // D layout of the 16x16 MFMAs (fp32 and bf16 alike): column = lane & 15, rows 4 (lane >> 4) + r.  Here the A operand carries the
  // M index in its row slot (lane & 15 of the A fragment = m) and B the N index: D[row = m][col = n]
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = nB + wn + 16 * tn + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = mB + wm + 16 * tm + 4 * kq + r;
        if (!RAGGED || (mm < M && n < N))
          __hip_atomic_fetch_add((GW_AS1 float*)(C + (size_t)mm * ldc + n), acc[tm][tn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
}

// ---- ReLU backward + bias gradient ----------------------------------------------------------------------------------
// thread t owns column t of a strip of rows: dz = dh * (h > 0) (h == nullptr: no mask), db[t] += sum of dz over the strip
__global__ __launch_bounds__(256) void relu_bwd_kernel(int64_t rows, int width, const float* __restrict__ dh, int ld_dh,
                                                       const float* __restrict__ h, int ld_h, float* __restrict__ dz, int ld_dz,
                                                       float* __restrict__ db, int strip) {
  const int col = threadIdx.x;
  if (col >= width) return;
  const int64_t r0 = (int64_t)blockIdx.x * strip;
  const int64_t r1 = r0 + strip < rows ? r0 + strip : rows;
  float s = 0.f;
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {  // 8 independent row loads in flight per thread
    float g[8], hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      g[u] = dh[(r + u) * ld_dh + col];
      hv[u] = h != nullptr ? h[(r + u) * ld_h + col] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (!(hv[u] > 0.f)) g[u] = 0.f;
      if (dz != nullptr) dz[(r + u) * ld_dz + col] = g[u];
      s += g[u];
    }
  }
  for (; r < r1; ++r) {
    float g = dh[r * ld_dh + col];
    if (h != nullptr && !(h[r * ld_h + col] > 0.f)) g = 0.f;
    if (dz != nullptr) dz[r * ld_dz + col] = g;
    s += g;
  }
  if (db != nullptr) __hip_atomic_fetch_add(db + col, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- LayerNorm backward (width 256, eps 1e-5, biased variance) --------------------------------------------------------
// one wave per row (lane l owns columns 4l..4l+3); a block walks a strip of rows, 4 at a time
__global__ __launch_bounds__(256) void ln_bwd_kernel(int64_t rows, const float* __restrict__ dn, int ld_dn,
                                                     const float* __restrict__ y, int ld_y, const float* __restrict__ gamma,
                                                     float* __restrict__ dy, int ld_dy, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int strip) {
  constexpr int U = 4;  // rows per wave in flight: the three row reductions are shuffle chains, so interleave rows for ILP
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * strip;
  const int64_t r1 = r0 + strip < rows ? r0 + strip : rows;
  const f32x4 gm = *(const f32x4*)(gamma + 4 * lane);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, dbt = {0.f, 0.f, 0.f, 0.f};
  for (int64_t rb = r0 + wave * U; rb < r1; rb += 4 * U) {
    f32x4 yv[U], dv[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = rb + u < r1;
      const int64_t r = ok[u] ? rb + u : rb;
      yv[u] = *(const f32x4*)(y + r * ld_y + 4 * lane);
      dv[u] = *(const f32x4*)(dn + r * ld_dn + 4 * lane);
    }
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = (yv[u].x + yv[u].y) + (yv[u].z + yv[u].w);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) s[u] += __shfl_xor(s[u], off);
    f32x4 d[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d[u] = yv[u] - s[u] * (1.0f / 256.0f);
      v[u] = d[u].x * d[u].x + d[u].y * d[u].y + d[u].z * d[u].z + d[u].w * d[u].w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] += __shfl_xor(v[u], off);
    f32x4 xh[U], g[U];
    float rstd[U], sg[U], sgx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rstd[u] = 1.0f / sqrtf(v[u] * (1.0f / 256.0f) + 1e-5f);
      xh[u] = d[u] * rstd[u];
      g[u] = dv[u] * gm;
      sg[u] = (g[u].x + g[u].y) + (g[u].z + g[u].w);
      sgx[u] = g[u].x * xh[u].x + g[u].y * xh[u].y + g[u].z * xh[u].z + g[u].w * xh[u].w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        sg[u] += __shfl_xor(sg[u], off);
        sgx[u] += __shfl_xor(sgx[u], off);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) {
        const float mg = sg[u] * (1.0f / 256.0f), mgx = sgx[u] * (1.0f / 256.0f);
        *(f32x4*)(dy + (rb + u) * ld_dy + 4 * lane) = (g[u] - mg - xh[u] * mgx) * rstd[u];
        dg += dv[u] * xh[u];
        dbt += dv[u];
      }
    }
  }
  // combine the four waves through LDS: one set of atomics per block (all blocks hit the same 512 addresses)
  __shared__ f32x4 red[2][4][64];
  red[0][wave][lane] = dg;
  red[1][wave][lane] = dbt;
  __syncthreads();
  if (wave == 0) {
    dg = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    dbt = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (dgamma) __hip_atomic_fetch_add(dgamma + 4 * lane + c, dg[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dbeta) __hip_atomic_fetch_add(dbeta + 4 * lane + c, dbt[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// LayerNorm backward for widths other than 256 (output heads: LayerNorm(n_out), regional_forecast.py:223-230): one wave per
// row, lane l owns columns l, l+64, l+128, l+192 (scalar accesses: no alignment requirement on the row strides).
__global__ __launch_bounds__(256) void ln_bwd_narrow_kernel(int64_t rows, int width, const float* __restrict__ dn, int ld_dn,
                                                            const float* __restrict__ y, int ld_y, const float* __restrict__ gamma,
                                                            float* __restrict__ dy, int ld_dy, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int strip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * strip;
  const int64_t r1 = r0 + strip < rows ? r0 + strip : rows;
  const float inv_n = 1.0f / (float)width;
  float gm[4], dg[4], dbt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    gm[j] = c < width ? gamma[c] : 0.f;
    dg[j] = dbt[j] = 0.f;
  }
  for (int64_t r = r0 + wave; r < r1; r += 4) {
    float yv[4], dv[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      yv[j] = c < width ? y[r * ld_y + c] : 0.f;
      dv[j] = c < width ? dn[r * ld_dn + c] : 0.f;
      s += yv[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * inv_n;
    float d[4], v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d[j] = (lane + 64 * j < width) ? yv[j] - mean : 0.f;
      v += d[j] * d[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const float rstd = 1.0f / sqrtf(v * inv_n + 1e-5f);
    float xh[4], g[4], sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = d[j] * rstd;
      g[j] = dv[j] * gm[j];
      sg += g[j];
      sgx += g[j] * xh[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sg += __shfl_xor(sg, off);
      sgx += __shfl_xor(sgx, off);
    }
    const float mg = sg * inv_n, mgx = sgx * inv_n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      if (c < width) dy[r * ld_dy + c] = (g[j] - mg - xh[j] * mgx) * rstd;
      dg[j] += dv[j] * xh[j];
      dbt[j] += dv[j];
    }
  }
  __shared__ float red[2][4][256];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][wave][lane + 64 * j] = dg[j];
    red[1][wave][lane + 64 * j] = dbt[j];
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < width) {
    const float a = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    const float b = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    if (dgamma) __hip_atomic_fetch_add(dgamma + c, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dbeta) __hip_atomic_fetch_add(dbeta + c, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- gather / segment sum of 256-float rows --------------------------------------------------------------------------
// out[b * n_idx + k] = table[(b * rows_pb + idx[k])] (+ add[b * n_idx + k]);  one wave per output row
__global__ __launch_bounds__(256) void gather_rows_kernel(int batch, int n_idx, const float* __restrict__ table, int rows_pb,
                                                          const int* __restrict__ idx, const float* __restrict__ add,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= (int64_t)batch * n_idx) return;
  const int b = (int)(c / n_idx), k = (int)(c - (int64_t)b * n_idx);
  const int r = idx ? idx[k] : k;
  f32x4 v = *(const f32x4*)(table + ((size_t)b * rows_pb + r) * 256 + 4 * lane);
  if (add) v += *(const f32x4*)(add + (size_t)c * 256 + 4 * lane);
  *(f32x4*)(out + (size_t)c * 256 + 4 * lane) = v;
}

// out[bo * n_seg + n] (+)= sum_{b in group(bo)} sum_{i = ptr[n] .. ptr[n+1]-1} rows[(b * rows_pb_in + perm[i])]
// batch_out == batch: per-sample sums;  batch_out == 1: also summed over the batch (tables shared by the batch).
// W waves share one segment (entries strided by W, partial sums combined through LDS): W = 1 for the short segments of
// the mesh (6-7 edges per node), 4 or 16 for the long ones (77 grid edges per mesh node, 3 000 grid points in a polar cell)
// so that a segment's rows stream in from many waves instead of one wave's dependent loop.
template <int W>
__global__ __launch_bounds__(W == 1 ? 256 : 64 * W) void segment_sum_kernel(int batch, int batch_out, int n_seg, const float* __restrict__ rows,
                                                             int rows_pb_in, const int* __restrict__ perm,
                                                             const int* __restrict__ ptr, float* __restrict__ out, int accumulate) {
  constexpr int SEGS = W == 1 ? 4 : 1;  // segments per block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t o = W == 1 ? (int64_t)blockIdx.x * SEGS + wave : (int64_t)blockIdx.x;
  __shared__ f32x4 part[W == 1 ? 1 : W][64];
  const bool live = o < (int64_t)batch_out * n_seg;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const int bo = (int)(o / n_seg), n = (int)(o - (int64_t)bo * n_seg);
    const int i0 = ptr[n], i1 = ptr[n + 1];
    const int b_lo = batch_out == 1 ? 0 : bo, b_hi = batch_out == 1 ? batch : bo + 1;
    const int w0 = W == 1 ? 0 : wave;
    for (int b = b_lo; b < b_hi; ++b) {
      const float* base = rows + (size_t)b * rows_pb_in * 256 + 4 * lane;
      int i = i0 + w0;
      for (; i + 3 * W < i1; i += 4 * W) {  // four independent row loads in flight
        const int ra = perm ? perm[i] : i, rb = perm ? perm[i + W] : i + W;
        const int rc = perm ? perm[i + 2 * W] : i + 2 * W, rd = perm ? perm[i + 3 * W] : i + 3 * W;
        const f32x4 va = *(const f32x4*)(base + (size_t)ra * 256), vb = *(const f32x4*)(base + (size_t)rb * 256);
        const f32x4 vc = *(const f32x4*)(base + (size_t)rc * 256), vd = *(const f32x4*)(base + (size_t)rd * 256);
        s += (va + vb) + (vc + vd);
      }
      for (; i < i1; i += W) {
        const int r = perm ? perm[i] : i;
        s += *(const f32x4*)(base + (size_t)r * 256);
      }
    }
  }
  if (W > 1) {
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || !live) return;
#pragma unroll
    for (int w = 1; w < W; ++w) s += part[w][lane];
  } else if (!live) {
    return;
  }
  float* p = out + (size_t)o * 256 + 4 * lane;
  if (accumulate) s += *(const f32x4*)p;
  *(f32x4*)p = s;
}

// ---- loss backward, optimiser ---------------------------------------------------------------------------------------
__global__ void nmse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                const float* __restrict__ inv_var, int inv_var_full, const float* __restrict__ lat_w, int num_lon, int nodes,
                                int channels, size_t total, float scale, const float* __restrict__ dloss,
                                float* __restrict__ dpred) {
  const float g = dloss[0] * scale;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / channels;
    const int ch = (int)(i - row * channels);
    const int n = (int)(row % nodes);
    float d = 2.0f * (pred[i] - target[i]) * lat_w[n / num_lon] * g;
    if (inv_var) d *= inv_var_full ? inv_var[i] : inv_var[ch];
    dpred[i] = d;
  }
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected, amsgrad off): one fused elementwise pass
__global__ void adamw_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                             float bc2_sqrt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

int fail(int code, const char* msg) { return set_error(code, msg); }

// ---- boundary nudging (regional_forecast.py:44-132) -------------------------------------------------------------------
// alpha = clamp(prior + W2.relu(W1.[regional | context | prior] + b1) + b2, 0, 1); out = (1 - alpha) regional + alpha context.
// One wave per row; lane l owns hidden units l, l+64, l+128, l+192 (hidden <= 256) and feature columns l, l+64, ...
__device__ __forceinline__ float nudging_alpha_raw(const float* __restrict__ x, int kin, int h, const float* __restrict__ w1t,
                                                   const float* __restrict__ b1, const float* __restrict__ w2,
                                                   const float* __restrict__ b2, int lane, float (&z)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) z[j] = (lane + 64 * j < h) ? b1[lane + 64 * j] : 0.f;
  for (int i = 0; i < kin; ++i) {
    const float xi = x[i];  // same address in every lane: one broadcast load
    const float* wrow = w1t + (size_t)i * h;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < h) z[j] = fmaf(wrow[lane + 64 * j], xi, z[j]);
  }
  float corr = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lane + 64 * j < h) corr = fmaf(w2[lane + 64 * j], fmaxf(z[j], 0.f), corr);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) corr += __shfl_xor(corr, off);
  return x[kin - 1] + (corr + b2[0]);
}

__global__ __launch_bounds__(256) void nudging_fwd_kernel(int64_t rows, int f, int h, const float* __restrict__ in, int ld_in,
                                                          const float* __restrict__ w1t, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;  // no barrier below
  const float* x = in + row * ld_in;
  float z[4];
  const float raw = nudging_alpha_raw(x, 2 * f + 1, h, w1t, b1, w2, b2, lane, z);
  const float alpha = fminf(fmaxf(raw, 0.f), 1.f);
  for (int c = lane; c < f; c += 64) out[row * f + c] = (1.f - alpha) * x[c] + alpha * x[f + c];
}

// writes d_in[:, 0:f] (gradient of the regional columns), dz = gradient at the hidden pre-activations, hid = relu(hidden),
// dcorr = gradient at the MLP output; the weight gradients are dz^T.in and dcorr^T.hid (gw_gemm_f32 TN).
__global__ __launch_bounds__(256) void nudging_bwd_kernel(int64_t rows, int f, int h, const float* __restrict__ in, int ld_in,
                                                          const float* __restrict__ w1, const float* __restrict__ w1t,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, const float* __restrict__ dout,
                                                          float* __restrict__ d_in, int ld_din, float* __restrict__ dz,
                                                          float* __restrict__ hid, float* __restrict__ dcorr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;  // whole waves leave together: the shuffles below always see 64 active lanes
  const int kin = 2 * f + 1;
  const float* x = in + row * ld_in;
  float z[4];
  const float raw = nudging_alpha_raw(x, kin, h, w1t, b1, w2, b2, lane, z);
  const float alpha = fminf(fmaxf(raw, 0.f), 1.f);
  float da = 0.f;
  for (int c = lane; c < f; c += 64) da = fmaf(dout[row * f + c], x[f + c] - x[c], da);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) da += __shfl_xor(da, off);
  const float dc = (raw >= 0.f && raw <= 1.f) ? da : 0.f;  // clamp passes the gradient on [min, max] (ATen clamp_backward)
  if (lane == 0) dcorr[row] = dc;
  float dzj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int hj = lane + 64 * j;
    dzj[j] = 0.f;
    if (hj < h) {
      dzj[j] = z[j] > 0.f ? dc * w2[hj] : 0.f;
      hid[row * h + hj] = fmaxf(z[j], 0.f);
      dz[row * h + hj] = dzj[j];
    }
  }
  for (int c0 = 0; c0 < f; c0 += 64) {  // uniform trip count: every lane takes part in the shuffles
    const int c = c0 + lane;
    const bool ok = c < f;
    float g = ok ? (1.f - alpha) * dout[row * f + c] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (64 * j >= h) break;
      const int lim = h - 64 * j < 64 ? h - 64 * j : 64;
      for (int l = 0; l < lim; ++l) {
        const float dzv = __shfl(dzj[j], l);
        if (ok) g = fmaf(dzv, w1[(size_t)(64 * j + l) * kin + c], g);
      }
    }
    if (ok) d_in[row * ld_din + c] = g;
  }
}

}  // namespace

extern "C" {

int gw_gemm_f32(int32_t mode, int64_t m, int32_t n, int64_t k, const float* a, int32_t lda, const float* b, int32_t ldb,
                float* c, int32_t ldc, float* colsum_a, void* stream) {
  if (!a || !b || !c || m < 0 || n < 0 || k < 0 || (mode != GW_GEMM_NN && mode != GW_GEMM_TN && mode != GW_GEMM_TN_BF16X3))
    return fail(GW_E_BADARG, "gw_gemm_f32: bad arguments");
  const bool x3 = mode == GW_GEMM_TN_BF16X3;
  if (x3) mode = GW_GEMM_TN;
  if (m == 0 || n == 0) return GW_OK;
  if (m >= ((int64_t)1 << 31) || k >= ((int64_t)1 << 31)) return fail(GW_E_UNSUPPORTED, "gw_gemm_f32: dimension exceeds int32");
  if (colsum_a && mode != GW_GEMM_TN) return fail(GW_E_UNSUPPORTED, "gw_gemm_f32: colsum_a is a TN-mode extra");
  const dim3 block(256);
  if (mode == GW_GEMM_TN) {
    if (k == 0) return GW_OK;  // nothing to add
    // slab of rows per block: enough blocks to fill the chip, each at least 256 rows deep.  Every slab ends in 64 K atomics per
    // 128 x 128 tile (0.4 T atomics/s measured: 12 of the 25 us at the 11 764 mesh-node rows, 33 of 83 us at a processor block's
    // 82 324 edge rows), so the split kernel - whose loads run two stages ahead and need fewer workgroups to cover the latency -
    // aims at one resident round (512 workgroups, two per CU) for the shorter tables: 83 -> 62 us at 82 324 rows, the same time at
    // 11 764 (scripts/probes/gemm_tn_x3_probe.py, scripts/gpu_ab_gemm_tn_x3.sh)
    const int tiles = (int)((m + 127) / 128) * ((n + 127) / 128);
    static const int tn_target = GW_TUNE("GW_TN_TARGET", 0), tn_min = GW_TUNE("GW_TN_MIN", 256), tn_cap = GW_TUNE("GW_TN_CAP", 0);
    // (one round only while the slabs of two rounds would be short: at 453 600 rows two rounds of 1 792-row slabs - workgroups out
    // of step, the atomics of one under the loads of the next - measured 267 us against 310 us for one round of 3 584-row slabs)
    const bool one_round = x3 && k * tiles < (int64_t)1024 * 1024;
    const int target = tn_target > 0 ? tn_target : (one_round ? 512 : 1024), cap = tn_cap > 0 ? tn_cap : (x3 ? 16384 : 4096);
    int64_t k_slab = (k * tiles + target - 1) / target;
    k_slab = ((k_slab + 63) / 64) * 64;
    if (k_slab < tn_min) k_slab = tn_min;
    if (k_slab > cap) k_slab = cap;
    const dim3 grid((unsigned)((m + 127) / 128), (unsigned)((n + 127) / 128), (unsigned)((k + k_slab - 1) / k_slab));
    if (x3) {  // split-operand products: any shape (4-byte loads, ragged tiles guarded)
      constexpr int lds = 2 * kTx3Stage;
      const int tiles_m = (int)((m + 127) / 128), slabs8 = ((int)grid.z + 7) / 8 * 8;
      static const int x3_tune = GW_TUNE("GW_TN_X3_TUNE", 0);
      const dim3 grid1((unsigned)(tiles * slabs8));
      if (m % 128 == 0 && n % 128 == 0) {
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)gemm_tn_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(gemm_tn_x3_kernel<false>, grid1, block, lds, (hipStream_t)stream, (int)m, n, (int)k, a, lda, b, ldb, c, ldc, (int)k_slab,
                           colsum_a, tiles_m, tiles, x3_tune);
      } else {
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)gemm_tn_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(gemm_tn_x3_kernel<true>, grid1, block, lds, (hipStream_t)stream, (int)m, n, (int)k, a, lda, b, ldb, c, ldc, (int)k_slab,
                           colsum_a, tiles_m, tiles, x3_tune);
      }
      return check_launch("gemm_tn_x3_kernel launch");
    }
    if (m % 128 == 0 && n % 128 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0) {
      hipLaunchKernelGGL(gemm_tn_lds_kernel, grid, block, 0, (hipStream_t)stream, (int)k, a, lda, b, ldb, c, ldc, (int)k_slab, colsum_a);
      return check_launch("gemm_tn_lds_kernel launch");
    }
    hipLaunchKernelGGL(gemm_tn_kernel, grid, block, 0, (hipStream_t)stream, (int)m, n, (int)k, a, lda, b, ldb, c, ldc, (int)k_slab, colsum_a);
  } else {
    const dim3 grid((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64), 1);
    hipLaunchKernelGGL(gemm_kernel<false>, grid, block, 0, (hipStream_t)stream, (int)m, n, (int)k, a, lda, b, ldb, c, ldc, 0);
  }
  return check_launch("gemm_kernel launch");
}

int gw_relu_backward(int64_t rows, int32_t width, const float* dh, int32_t ld_dh, const float* h, int32_t ld_h, float* dz,
                     int32_t ld_dz, float* db, void* stream) {
  if (!dh || rows < 0 || width <= 0) return fail(GW_E_BADARG, "gw_relu_backward: bad arguments");
  if (rows == 0) return GW_OK;
  if (width > 256) return relu_mask_wide_launch(rows, width, dh, ld_dh, h, ld_h, dz, ld_dz, db, stream);  // wide models (gw_wide.hip)
  // few blocks per column - the bias-gradient atomics of all blocks hit the same 256 addresses - but enough of them to fill the
  // chip: mesh-sized inputs (11 764 rows at 1 degree, batch 2) ran on 46 blocks with a fixed strip of 256 rows
  int strip = (int)((rows + 1023) / 1024);
  strip = (strip + 15) / 16 * 16;
  if (strip > 256) strip = 256;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((rows + strip - 1) / strip)), dim3(256), 0, (hipStream_t)stream, rows, width,
                     dh, ld_dh, h, ld_h, dz, ld_dz, db, strip);
  return check_launch("relu_bwd_kernel launch");
}

int gw_layernorm_backward(int64_t rows, int32_t width, const float* dn, int32_t ld_dn, const float* y, int32_t ld_y,
                          const float* gamma, float* dy, int32_t ld_dy, float* dgamma, float* dbeta, void* stream) {
  if (!dn || !y || !gamma || !dy || rows < 0 || width <= 0 || ld_dn < width || ld_y < width || ld_dy < width)
    return fail(GW_E_BADARG, "gw_layernorm_backward: bad arguments");
  if (rows == 0) return GW_OK;
  if (width > 256) return ln_bwd_wide_launch(rows, width, dn, ld_dn, y, ld_y, gamma, dy, ld_dy, dgamma, dbeta, stream);
  // rows per block: 512 for large inputs (one set of dgamma / dbeta atomics per block), down to 16 so that mesh-sized inputs
  // still spread over ~1 000 blocks (they ran on 23 blocks - 87 us for 36 MB - with the fixed strip)
  int strip = (int)((rows + 1023) / 1024);
  strip = (strip + 15) / 16 * 16;
  if (strip > 512) strip = 512;
  const dim3 grid((unsigned)((rows + strip - 1) / strip));
  if (width == 256 && (ld_dn | ld_y | ld_dy) % 4 == 0) {
    hipLaunchKernelGGL(ln_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, dn, ld_dn, y, ld_y, gamma, dy, ld_dy, dgamma,
                       dbeta, strip);
    return check_launch("ln_bwd_kernel launch");
  }
  hipLaunchKernelGGL(ln_bwd_narrow_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, width, dn, ld_dn, y, ld_y, gamma, dy, ld_dy,
                     dgamma, dbeta, strip);
  return check_launch("ln_bwd_narrow_kernel launch");
}

int gw_gather_rows(int32_t batch, int32_t n_idx, const float* table, int32_t rows_per_batch, const int32_t* idx,
                   const float* add, float* out, void* stream) {
  if (!table || !out || batch <= 0 || n_idx < 0) return fail(GW_E_BADARG, "gw_gather_rows: bad arguments");
  const int64_t total = (int64_t)batch * n_idx;
  if (total == 0) return GW_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, batch, n_idx, table,
                     rows_per_batch, idx, add, out);
  return check_launch("gather_rows_kernel launch");
}

int gw_segment_sum_rows(int32_t batch, int32_t batch_out, int32_t n_seg, const float* rows, int32_t rows_per_batch_in,
                        const int32_t* perm, const int32_t* ptr, float* out, int32_t accumulate, void* stream) {
  if (!rows || !ptr || !out || batch <= 0 || n_seg < 0 || (batch_out != batch && batch_out != 1))
    return fail(GW_E_BADARG, "gw_segment_sum_rows: bad arguments");
  const int64_t total = (int64_t)batch_out * n_seg;
  if (total == 0) return GW_OK;
  // average rows per output segment decides how many waves share a segment (n_rows_hint: rows of one batch element)
  const int64_t per_seg = n_seg > 0 ? ((int64_t)(rows_per_batch_in > 0 ? rows_per_batch_in : 1) * (batch_out == 1 ? batch : 1)) / n_seg : 0;
  if (per_seg <= 12)
    hipLaunchKernelGGL(segment_sum_kernel<1>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, batch, batch_out,
                       n_seg, rows, rows_per_batch_in, perm, ptr, out, accumulate);
  else if (per_seg <= 128)
    hipLaunchKernelGGL(segment_sum_kernel<4>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, batch, batch_out, n_seg, rows,
                       rows_per_batch_in, perm, ptr, out, accumulate);
  else
    hipLaunchKernelGGL(segment_sum_kernel<16>, dim3((unsigned)total), dim3(1024), 0, (hipStream_t)stream, batch, batch_out, n_seg, rows,
                       rows_per_batch_in, perm, ptr, out, accumulate);
  return check_launch("segment_sum_kernel launch");
}

int gw_normalized_mse_backward(const float* pred, const float* target, const float* inv_var, int32_t inv_var_full,
                               const float* lat_weights, int32_t num_unique_lat, int32_t batch, int32_t nodes, int32_t channels, const float* dloss,
                               float* dpred, void* stream) {
  if (!pred || !target || !lat_weights || !dloss || !dpred || num_unique_lat <= 0 || batch <= 0 || nodes <= 0 || channels <= 0)
    return fail(GW_E_BADARG, "gw_normalized_mse_backward: bad arguments");
  const int num_lon = nodes / num_unique_lat;
  if (num_lon <= 0) return fail(GW_E_BADARG, "gw_normalized_mse_backward: nodes must equal num_unique_lat * num_lon");
  const size_t total = (size_t)batch * nodes * channels;
  const float scale = 1.0f / ((float)channels * (float)batch * (float)nodes);
  int grid = (int)((total + 1023) / 1024);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(nmse_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, target, inv_var, inv_var_full, lat_weights,
                     num_lon, nodes, channels, total, scale, dloss, dpred);
  return check_launch("nmse_bwd_kernel launch");
}

int gw_adamw_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step <= 0) return fail(GW_E_BADARG, "gw_adamw_step: bad arguments");
  if (n == 0) return GW_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  int grid = (int)((n + 1023) / 1024);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (size_t)n, param, grad, exp_avg, exp_avg_sq, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
  return check_launch("adamw_kernel launch");
}

int gw_nudging_forward(int64_t rows, int32_t feat, int32_t hidden, const float* in, int32_t ld_in, const float* w1t, const float* b1,
                       const float* w2, const float* b2, float* out, void* stream) {
  if (!in || !w1t || !b1 || !w2 || !b2 || !out || rows < 0 || feat <= 0 || feat > 256 || hidden <= 0 || hidden > 256 ||
      ld_in < 2 * feat + 1)
    return fail(GW_E_BADARG, "gw_nudging_forward: bad arguments (feat, hidden in 1..256; ld_in >= 2 feat + 1)");
  if (rows == 0) return GW_OK;
  hipLaunchKernelGGL(nudging_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, feat, hidden, in,
                     ld_in, w1t, b1, w2, b2, out);
  return check_launch("nudging_fwd_kernel launch");
}

int gw_nudging_backward(int64_t rows, int32_t feat, int32_t hidden, const float* in, int32_t ld_in, const float* w1, const float* w1t,
                        const float* b1, const float* w2, const float* b2, const float* dout, float* d_in, int32_t ld_din, float* dz,
                        float* hid, float* dcorr, void* stream) {
  if (!in || !w1 || !w1t || !b1 || !w2 || !b2 || !dout || !d_in || !dz || !hid || !dcorr || rows < 0 || feat <= 0 || feat > 256 ||
      hidden <= 0 || hidden > 256 || ld_in < 2 * feat + 1 || ld_din < feat)
    return fail(GW_E_BADARG, "gw_nudging_backward: bad arguments");
  if (rows == 0) return GW_OK;
  hipLaunchKernelGGL(nudging_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, feat, hidden, in,
                     ld_in, w1, w1t, b1, w2, b2, dout, d_in, ld_din, dz, hid, dcorr);
  return check_launch("nudging_bwd_kernel launch");
}

}  // extern "C"
