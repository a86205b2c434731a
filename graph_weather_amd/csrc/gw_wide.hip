// gw_wide.hip - the message-passing path for models WIDER than the fused kernels' 256 features (the reference's own training
// script builds node / edge / hidden widths of 1024: train/run.py:493-497).  Nothing here is fused across layers: every
// nn.Linear (graph_net_block.py:45-61) is one fp32-MFMA GEMM with bias and ReLU in its epilogue, LayerNorm, the x[row] / x[col]
// gathers (MetaLayer, :221-228) and the scatter_sum (:188) are one HBM-bound kernel each, at any width.  The fused kernels
// (gw_kernels.hip, gw_edge.hip, gw_edge16.hip) remain the path for widths <= 256 - this file is correctness and coverage,
// with matrix products on the matrix cores, not the tuned hot path.
//
//   gw_linear_forward          out = act(x . W^T + b)            W is nn.Linear.weight as it lies in memory ([n, k] row-major)
//   gw_layernorm_forward       out = LayerNorm(y; gamma, beta, eps 1e-5, biased variance) (+ residual)
//   gw_add_rows                out = a + b                        (the residual add of an MLP without norm)
//   gw_gather_rows_wide        out[b, i, :] = table[b, idx[i], :]
//   gw_segment_sum_rows_wide   out[bo, n, :] = sum_b sum_{i in seg(n)} rows[b, perm[i], :]   (CSR walk: no atomics, one order)
//   ln_bwd_wide_launch / relu_mask_wide_launch: widths > 256 of gw_layernorm_backward / gw_relu_backward (gw_train.hip)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/gw_amd.h"
#include "gw_device.hpp"
#include "gw_internal.hpp"

using namespace gw;

namespace {

int failw(int code, const char* msg) { return set_error(code, msg); }

// ---- C[m][n] = act(sum_k A[m][k] * W[n][k] + bias[n]) ------------------------------------------------------------------
// Block = 4 waves, tile 128 (m) x 128 (n); wave (wm, wn) owns 64 x 64 = 4 x 4 MFMA tiles (v_mfma_f32_16x16x4_f32: fp32 in,
// fp32 accumulate - an fmaf chain like the fused kernels).  Both operands are K-contiguous, so a 16-deep K chunk of each tile
// (128 rows x 16 floats) is staged through LDS by coalesced 16-byte loads, double buffered through registers; the LDS row
// stride of 20 floats keeps the operand reads (one ds_read_b128 per 16-row tile and chunk) spread over the banks.
constexpr int kNtKC = 16;
constexpr int kNtLd = 20;

// Row tables added to the product before the activation: out[m] += table_i[b * rows_pb_i + idx_i[k]] with (b, k) = (m / rpb, m % rpb)
// (idx NULL: k itself; rows_pb 0: one table shared by the batch).  This is how the layer-1 split of the fused kernels
// (cat[x_s, x_d, e] . W1^T = x_s.Ws^T[src] + x_d.Wd^T[dst] + e.We^T) reaches the wide path: node products are made once per
// node and gathered per edge here, in the epilogue of the edge-level product.
struct Addends {
  int n;                 // 0..3
  int rows_per_batch;    // rpb
  const float* table[3];
  const int* idx[3];
  int ld[3];
  int rows_pb[3];
};

template <bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_nt_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                      const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                      int relu, float* __restrict__ C, int ldc, const Addends ga) {
  __shared__ __attribute__((aligned(16))) float As[2][128 * kNtLd];
  __shared__ __attribute__((aligned(16))) float Ws[2][128 * kNtLd];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * 128;
  const int n0 = blockIdx.y * 128;
  // staging: thread t copies 8 consecutive k of row (t >> 1) of each tile
  const int srow = threadIdx.x >> 1, sk = 8 * (threadIdx.x & 1);
  const bool a_ok = m0 + srow < M, w_ok = n0 + srow < N;
  const float* arow = A + (size_t)(a_ok ? m0 + srow : 0) * lda;
  const float* wrow = W + (size_t)(w_ok ? n0 + srow : 0) * ldw;
  f32x4 ra[2], rw[2];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + sk + 4 * h;
      if (ALIGNED) {
        ra[h] = (a_ok && k < K) ? ldg4(arow + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        rw[h] = (w_ok && k < K) ? ldg4(wrow + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ra[h][r] = (a_ok && k + r < K) ? ldg1(arow + k + r) : 0.f;
          rw[h][r] = (w_ok && k + r < K) ? ldg1(wrow + k + r) : 0.f;
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *(f32x4*)(&As[buf][srow * kNtLd + sk + 4 * h]) = ra[h];
      *(f32x4*)(&Ws[buf][srow * kNtLd + sk + 4 * h]) = rw[h];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nch = (K + kNtKC - 1) / kNtKC;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    if (c + 1 < nch) load_chunk((c + 1) * kNtKC);
    // The k a lane feeds into K-step ks is 4 kq + ks (any assignment of the chunk's 16 k to (K-step, lane group) pairs is a
    // valid order of the sum as long as both operands agree): its four K-steps are 16 contiguous bytes, one ds_read_b128.
    const float* as = &As[buf][(64 * wm + i) * kNtLd + 4 * kq];
    const float* ws = &Ws[buf][(64 * wn + i) * kNtLd + 4 * kq];
    f32x4 a4[4], b4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a4[t] = *(const f32x4*)(as + 16 * t * kNtLd);
      b4[t] = *(const f32x4*)(ws + 16 * t * kNtLd);
    }
#pragma unroll
    for (int ks = 0; ks < kNtKC / 4; ++ks) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[tm][ks], b4[tn][ks], acc[tm][tn], 0, 0, 0);
    }
    if (c + 1 < nch) store_chunk(buf ^ 1);  // last read in iteration c - 1, before its barrier
    __syncthreads();
  }
  // D layout: column (n) = lane & 15, rows (m) = 4 * (lane >> 4) + r
  if (ga.n > 0) {  // gathered row tables: 16 lanes read 64 contiguous bytes of a table row
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t mm = m0 + 64 * wm + 16 * tm + 4 * kq + r;
        if (mm >= M) continue;
        const int b = (int)(mm / ga.rows_per_batch), k = (int)(mm - (int64_t)b * ga.rows_per_batch);
        for (int a = 0; a < ga.n; ++a) {
          const int tr = ga.idx[a] != nullptr ? ldgi(ga.idx[a] + k) : k;
          const float* row = ga.table[a] + ((size_t)b * ga.rows_pb[a] + tr) * ga.ld[a];
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + 64 * wn + 16 * tn + i;
            if (n < N) acc[tm][tn][r] += ldg1(row + n);
          }
        }
      }
  }
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int n = n0 + 64 * wn + 16 * tn + i;
    if (n >= N) continue;
    const float bv = bias != nullptr ? ldg1(bias + n) : 0.f;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t mm = m0 + 64 * wm + 16 * tm + 4 * kq + r;
        if (mm < M) {
          float v = acc[tm][tn][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          stg1(C + (size_t)mm * ldc + n, v);
        }
      }
  }
}

// out[m] = act(bias + sum_i table_i[row_i(m)]): layer 1 of an edge MLP whose every operand is a projected table (edge features
// shared by the batch: first processor block, encoder, decoder) - no matrix work per edge at all.
__global__ __launch_bounds__(256) void gather_sum_kernel(int64_t rows, int width, const float* __restrict__ bias, int relu,
                                                         float* __restrict__ out, int ldo, const Addends ga) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= width) return;
  const float bv = bias != nullptr ? ldg1(bias + c) : 0.f;
  for (int64_t m = blockIdx.x; m < rows; m += gridDim.x) {
    const int b = (int)(m / ga.rows_per_batch), k = (int)(m - (int64_t)b * ga.rows_per_batch);
    float v = bv;
    for (int a = 0; a < ga.n; ++a) {
      const int tr = ga.idx[a] != nullptr ? ldgi(ga.idx[a] + k) : k;
      v += ldg1(ga.table[a] + ((size_t)b * ga.rows_pb[a] + tr) * ga.ld[a] + c);
    }
    if (relu) v = fmaxf(v, 0.f);
    stg1(out + (size_t)m * ldo + c, v);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- LayerNorm forward: one wave per row, lane l owns columns l + 64 j ---------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(int64_t rows, int width, const float* __restrict__ y, int ld_y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ res, int ld_res, int64_t res_period,
                                                          float* __restrict__ out, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t rr = res_period > 0 ? r % res_period : r;  // residual rows shared by the batch repeat with this period
  float v[NJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < width ? ldg1(y + (size_t)r * ld_y + c) : 0.f;
    s += v[j];
  }
  const float inv_n = 1.0f / (float)width;
  const float mean = wave_sum(s) * inv_n;
  float s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float d = (lane + 64 * j < width) ? v[j] - mean : 0.f;
    s2 = fmaf(d, d, s2);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_n + 1e-5f);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < width) {
      float o = (v[j] - mean) * rstd * ldg1(gamma + c) + ldg1(beta + c);
      if (res != nullptr) o += ldg1(res + (size_t)rr * ld_res + c);
      stg1(out + (size_t)r * ld_out + c, o);
    }
  }
}

// ---- LayerNorm backward: dy = rstd (g - mean(g) - xhat mean(g xhat)), g = dn gamma; dgamma += dn xhat; dbeta += dn --------
template <int NJ>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(int64_t rows, int width, const float* __restrict__ dn, int ld_dn,
                                                          const float* __restrict__ y, int ld_y, const float* __restrict__ gamma,
                                                          float* __restrict__ dy, int ld_dy, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int strip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * strip;
  const int64_t r1 = r0 + strip < rows ? r0 + strip : rows;
  const float inv_n = 1.0f / (float)width;
  float gm[NJ], dg[NJ], db[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    gm[j] = c < width ? ldg1(gamma + c) : 0.f;
    dg[j] = 0.f;
    db[j] = 0.f;
  }
  for (int64_t r = r0 + wave; r < r1; r += 4) {
    float yv[NJ], dv[NJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      yv[j] = c < width ? ldg1(y + (size_t)r * ld_y + c) : 0.f;
      dv[j] = c < width ? ldg1(dn + (size_t)r * ld_dn + c) : 0.f;
      s += yv[j];
    }
    const float mean = wave_sum(s) * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      yv[j] = (lane + 64 * j < width) ? yv[j] - mean : 0.f;
      s2 = fmaf(yv[j], yv[j], s2);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_n + 1e-5f);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      yv[j] *= rstd;  // xhat
      const float g = dv[j] * gm[j];
      sg += g;
      sgx = fmaf(g, yv[j], sgx);
      dg[j] = fmaf(dv[j], yv[j], dg[j]);
      db[j] += dv[j];
    }
    const float mg = wave_sum(sg) * inv_n, mgx = wave_sum(sgx) * inv_n;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < width) stg1(dy + (size_t)r * ld_dy + c, (dv[j] * gm[j] - mg - yv[j] * mgx) * rstd);
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < width) {
      if (dgamma != nullptr) __hip_atomic_fetch_add((GW_AS1 float*)(dgamma + c), dg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dbeta != nullptr) __hip_atomic_fetch_add((GW_AS1 float*)(dbeta + c), db[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- elementwise: dz = dh * (h > 0);  out = a + b -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_mask_wide_kernel(int64_t rows, int width, const float* __restrict__ dh, int ld_dh,
                                                             const float* __restrict__ h, int ld_h, float* __restrict__ dz, int ld_dz,
                                                             float* __restrict__ db) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= width) return;
  float s = 0.f;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    float g = ldg1(dh + (size_t)r * ld_dh + c);
    if (h != nullptr && !(ldg1(h + (size_t)r * ld_h + c) > 0.f)) g = 0.f;
    if (dz != nullptr) stg1(dz + (size_t)r * ld_dz + c, g);
    s += g;
  }
  if (db != nullptr) __hip_atomic_fetch_add((GW_AS1 float*)(db + c), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void add_rows_kernel(int64_t rows, int width, const float* __restrict__ a, int lda,
                                                       const float* __restrict__ b, int ldb, float* __restrict__ out, int ldo) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= width) return;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
    stg1(out + (size_t)r * ldo + c, ldg1(a + (size_t)r * lda + c) + ldg1(b + (size_t)r * ldb + c));
}

// ---- gather / segment sum at any width (thread = column, block.y = 256-column slab) ----------------------------------------
__global__ __launch_bounds__(256) void gather_wide_kernel(int batch, int n_idx, int width, const float* __restrict__ table, int ld,
                                                          int rows_pb, const int* __restrict__ idx, float* __restrict__ out, int ldo) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= width) return;
  const int64_t total = (int64_t)batch * n_idx;
  for (int64_t o = blockIdx.x; o < total; o += gridDim.x) {
    const int b = (int)(o / n_idx), k = (int)(o - (int64_t)b * n_idx);
    const int r = idx ? ldgi(idx + k) : k;
    stg1(out + (size_t)o * ldo + c, ldg1(table + ((size_t)b * rows_pb + r) * ld + c));
  }
}

__global__ __launch_bounds__(256) void segment_sum_wide_kernel(int batch, int batch_out, int n_seg, int width,
                                                               const float* __restrict__ rows, int ld, int rows_pb_in,
                                                               const int* __restrict__ perm, const int* __restrict__ ptr,
                                                               float* __restrict__ out, int ldo) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= width) return;
  const int64_t total = (int64_t)batch_out * n_seg;
  for (int64_t o = blockIdx.x; o < total; o += gridDim.x) {
    const int bo = (int)(o / n_seg), n = (int)(o - (int64_t)bo * n_seg);
    const int i0 = ldgi(ptr + n), i1 = ldgi(ptr + n + 1);
    const int b_lo = batch_out == 1 ? 0 : bo, b_hi = batch_out == 1 ? batch : bo + 1;
    float s = 0.f;
    for (int b = b_lo; b < b_hi; ++b) {
      const float* base = rows + (size_t)b * rows_pb_in * ld + c;
      int i = i0;
      for (; i + 3 < i1; i += 4) {  // four independent row loads in flight
        const int ra = perm ? ldgi(perm + i) : i, rb = perm ? ldgi(perm + i + 1) : i + 1;
        const int rc = perm ? ldgi(perm + i + 2) : i + 2, rd = perm ? ldgi(perm + i + 3) : i + 3;
        const float va = ldg1(base + (size_t)ra * ld), vb = ldg1(base + (size_t)rb * ld);
        const float vc = ldg1(base + (size_t)rc * ld), vd = ldg1(base + (size_t)rd * ld);
        s += (va + vb) + (vc + vd);
      }
      for (; i < i1; ++i) s += ldg1(base + (size_t)(perm ? ldgi(perm + i) : i) * ld);
    }
    stg1(out + (size_t)o * ldo + c, s);
  }
}

unsigned row_blocks(int64_t rows) { return (unsigned)(rows < 65536 ? (rows > 0 ? rows : 1) : 65536); }

template <typename F>
int by_width(int width, F f) {  // NJ = columns per lane (64 lanes per row)
  if (width <= 512) return f(std::integral_constant<int, 8>{});
  if (width <= 1024) return f(std::integral_constant<int, 16>{});
  if (width <= 2048) return f(std::integral_constant<int, 32>{});
  return f(std::integral_constant<int, 64>{});
}

}  // namespace

namespace gw {

int ln_bwd_wide_launch(int64_t rows, int32_t width, const float* dn, int32_t ld_dn, const float* y, int32_t ld_y, const float* gamma,
                       float* dy, int32_t ld_dy, float* dgamma, float* dbeta, void* stream) {
  if (width > 4096) return failw(GW_E_UNSUPPORTED, "gw_layernorm_backward: widths above 4096 are not implemented");
  const int strip = 256;
  const dim3 grid((unsigned)((rows + strip - 1) / strip));
  by_width(width, [&](auto nj) {
    hipLaunchKernelGGL(ln_bwd_wide_kernel<decltype(nj)::value>, grid, dim3(256), 0, (hipStream_t)stream, rows, width, dn, ld_dn, y, ld_y,
                       gamma, dy, ld_dy, dgamma, dbeta, strip);
    return 0;
  });
  return check_launch("ln_bwd_wide_kernel launch");
}

int relu_mask_wide_launch(int64_t rows, int32_t width, const float* dh, int32_t ld_dh, const float* h, int32_t ld_h, float* dz,
                          int32_t ld_dz, float* db, void* stream) {
  // with a bias gradient: few row blocks, so that the atomics of all blocks on the same `width` addresses stay cheap
  const unsigned rb = db != nullptr ? (unsigned)(rows < 2048 ? (rows > 0 ? rows : 1) : 2048) : row_blocks(rows);
  hipLaunchKernelGGL(relu_mask_wide_kernel, dim3(rb, (unsigned)((width + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     rows, width, dh, ld_dh, h, ld_h, dz, ld_dz, db);
  return check_launch("relu_mask_wide_kernel launch");
}

}  // namespace gw

extern "C" {

static int linear_launch(int64_t rows, int32_t k, int32_t n, const float* x, int32_t ldx, const float* w, int32_t ldw, const float* bias,
                         int32_t relu, float* out, int32_t ldo, const Addends& ga, void* stream) {
  if (rows >= ((int64_t)1 << 31)) return failw(GW_E_UNSUPPORTED, "gw_linear_forward: row count exceeds int32");
  if (k == 0) {  // no matrix product: bias + gathered rows
    hipLaunchKernelGGL(gather_sum_kernel, dim3(row_blocks(rows), (unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n,
                       bias, relu, out, ldo, ga);
    return check_launch("gather_sum_kernel launch");
  }
  const dim3 grid((unsigned)((rows + 127) / 128), (unsigned)((n + 127) / 128));
  const bool aligned = k % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0;
  if (aligned)
    hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (int)rows, n, k, x, ldx, w, ldw, bias, relu, out, ldo, ga);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (int)rows, n, k, x, ldx, w, ldw, bias, relu, out, ldo, ga);
  return check_launch("gemm_nt_kernel launch");
}

int gw_linear_forward(int64_t rows, int32_t k, int32_t n, const float* x, int32_t ldx, const float* w, int32_t ldw, const float* bias,
                      int32_t relu, float* out, int32_t ldo, void* stream) {
  if (!x || !w || !out || rows < 0 || k <= 0 || n <= 0 || ldx < k || ldw < k || ldo < n)
    return failw(GW_E_BADARG, "gw_linear_forward: bad arguments");
  if (rows == 0) return GW_OK;
  Addends ga = {};
  ga.rows_per_batch = 1;
  return linear_launch(rows, k, n, x, ldx, w, ldw, bias, relu, out, ldo, ga, stream);
}

int gw_linear_gather_forward(int64_t rows, int32_t rows_per_batch, int32_t k, int32_t n, const float* x, int32_t ldx, const float* w,
                             int32_t ldw, const float* bias, int32_t n_add, const float* const* add_table, const int32_t* const* add_idx,
                             const int32_t* add_ld, const int32_t* add_rows_pb, int32_t relu, float* out, int32_t ldo, void* stream) {
  if (!out || rows < 0 || k < 0 || n <= 0 || ldo < n || n_add < 0 || n_add > 3 || rows_per_batch <= 0 || (k == 0 && n_add == 0) ||
      (k > 0 && (!x || !w || ldx < k || ldw < k)) || (n_add > 0 && (!add_table || !add_idx || !add_ld || !add_rows_pb)))
    return failw(GW_E_BADARG, "gw_linear_gather_forward: bad arguments");
  if (rows == 0) return GW_OK;
  Addends ga = {};
  ga.n = n_add;
  ga.rows_per_batch = rows_per_batch;
  for (int a = 0; a < n_add; ++a) {
    if (!add_table[a] || add_ld[a] < n || add_rows_pb[a] < 0) return failw(GW_E_BADARG, "gw_linear_gather_forward: bad addend");
    ga.table[a] = add_table[a];
    ga.idx[a] = add_idx[a];
    ga.ld[a] = add_ld[a];
    ga.rows_pb[a] = add_rows_pb[a];
  }
  return linear_launch(rows, k, n, x, ldx, w, ldw, bias, relu, out, ldo, ga, stream);
}

int gw_layernorm_forward(int64_t rows, int32_t width, const float* y, int32_t ld_y, const float* gamma, const float* beta,
                         const float* res, int32_t ld_res, int64_t res_period, float* out, int32_t ld_out, void* stream) {
  if (!y || !gamma || !beta || !out || rows < 0 || width <= 0 || ld_y < width || ld_out < width || (res && ld_res < width) || res_period < 0)
    return failw(GW_E_BADARG, "gw_layernorm_forward: bad arguments");
  if (width > 4096) return failw(GW_E_UNSUPPORTED, "gw_layernorm_forward: widths above 4096 are not implemented");
  if (rows == 0) return GW_OK;
  const dim3 grid((unsigned)((rows + 3) / 4));
  by_width(width, [&](auto nj) {
    hipLaunchKernelGGL(ln_fwd_wide_kernel<decltype(nj)::value>, grid, dim3(256), 0, (hipStream_t)stream, rows, width, y, ld_y, gamma, beta,
                       res, ld_res, res_period, out, ld_out);
    return 0;
  });
  return check_launch("ln_fwd_wide_kernel launch");
}

int gw_add_rows(int64_t rows, int32_t width, const float* a, int32_t lda, const float* b, int32_t ldb, float* out, int32_t ldo,
                void* stream) {
  if (!a || !b || !out || rows < 0 || width <= 0 || lda < width || ldb < width || ldo < width)
    return failw(GW_E_BADARG, "gw_add_rows: bad arguments");
  if (rows == 0) return GW_OK;
  hipLaunchKernelGGL(add_rows_kernel, dim3(row_blocks(rows), (unsigned)((width + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows,
                     width, a, lda, b, ldb, out, ldo);
  return check_launch("add_rows_kernel launch");
}

int gw_gather_rows_wide(int32_t batch, int32_t n_idx, int32_t width, const float* table, int32_t ld, int32_t rows_per_batch,
                        const int32_t* idx, float* out, int32_t ldo, void* stream) {
  if (!table || !out || batch <= 0 || n_idx < 0 || width <= 0 || ld < width || ldo < width || rows_per_batch < 0)
    return failw(GW_E_BADARG, "gw_gather_rows_wide: bad arguments");
  const int64_t total = (int64_t)batch * n_idx;
  if (total == 0) return GW_OK;
  hipLaunchKernelGGL(gather_wide_kernel, dim3(row_blocks(total), (unsigned)((width + 255) / 256)), dim3(256), 0, (hipStream_t)stream, batch,
                     n_idx, width, table, ld, rows_per_batch, idx, out, ldo);
  return check_launch("gather_wide_kernel launch");
}

int gw_segment_sum_rows_wide(int32_t batch, int32_t batch_out, int32_t n_seg, int32_t width, const float* rows, int32_t ld,
                             int32_t rows_per_batch_in, const int32_t* perm, const int32_t* ptr, float* out, int32_t ldo, void* stream) {
  if (!rows || !ptr || !out || batch <= 0 || n_seg < 0 || width <= 0 || ld < width || ldo < width || (batch_out != batch && batch_out != 1))
    return failw(GW_E_BADARG, "gw_segment_sum_rows_wide: bad arguments");
  const int64_t total = (int64_t)batch_out * n_seg;
  if (total == 0) return GW_OK;
  hipLaunchKernelGGL(segment_sum_wide_kernel, dim3(row_blocks(total), (unsigned)((width + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     batch, batch_out, n_seg, width, rows, ld, rows_per_batch_in, perm, ptr, out, ldo);
  return check_launch("segment_sum_wide_kernel launch");
}

}  // extern "C"
