// gw_internal.hpp - host-side declarations shared between the translation units of libgw_amd.so (not part of the ABI).
#ifndef GW_INTERNAL_HPP
#define GW_INTERNAL_HPP

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/gw_amd.h"

namespace gw {

// Arguments of the general fused-MLP kernels (fp32: chain_kernel in gw_kernels.hip, bf16: chain16_kernel in gw_bf16.hip).
// Weight pointers are typed float* but address the packed stream of the launch's weight dtype.
enum { EPI_ROWS = 0, EPI_EDGE = 1, EPI_DEC = 2 };

struct ChainArgs {
  int n_cols;          // total columns (batch * cols_per_batch)
  int cols_per_batch;
  unsigned long long* dbg;  // optional per-workgroup timestamp records (16 x u64 each), debug only
  int dbg_cap;
  int stagger;         // start-up delay (x 8k cycles) of every second wave of workgroups, see chain_kernel
  // layer-1 operands
  const float* seg_ptr[3];
  const int* seg_idx[3];
  int seg_rows_pb[3];
  int seg_ld[3];
  int seg_k[3];
  int seg_proj[3];     // operand is already multiplied by its layer-1 weight slice: rows are [hidden] wide
  int seg_half[3];     // ... and stored as fp16 rows (GW_LAYOUT_ROWS_F16; seg_ld counts halves): bf16 kernels only
  int seg_bf16k[3];    // raw operand stored as bf16 rows in K order (GW_LAYOUT_ROWS_BF16K; seg_ld counts bf16 values): bf16 kernels
  // single-layer projection mode: blockIdx.y selects the weight slice / output table.
  // POST mode (node update): after LayerNorm + residual the new rows x' are multiplied, still in registers, by n_post packed
  // [256, 256] slices - the layer-1 products of the NEXT block's edge MLP (P_s = x' Ws^T, P_d = x' Wd^T) - and written to
  // proj_out[s]: the separate projection launch of every block and its re-read of x' disappear.
  const float* proj_w[4];
  float* proj_out[4];
  int n_post;
  // HEAD mode (bf16 node update of the decoder): the new rows never leave the registers - they feed the output head
  // (AssimilatorDecoder.node_decoder, 256 -> 128 -> 128 -> <= 80 features, no norm) and only its output (+ residual) is stored
  const float* hd_w1;  // packed [128, 256]
  const float* hd_b1;
  const float* hd_w2;  // packed [128, 128]
  const float* hd_b2;
  const float* hd_w3;  // packed [<= 80, 128] (5 row tiles)
  const float* hd_b3;
  int tune16;          // tuning builds only (GW_CHAIN16_TUNE): timing experiments of the bf16 chain kernel (wrong results)
  int proj_half;       // bf16 launches: the products are stored as fp16 rows (256 halves per row, GW_LAYOUT_ROWS_F16), clamped to
                       // the fp16 range - half the bytes for the per-edge gathers that consume them
  // weights
  const float* w1[3];
  const float* b1;
  const float* w_mid;
  const float* b_mid;
  const float* w_out;
  const float* b_out;
  const float* gamma;
  const float* beta;
  int n_mid;
  // residual
  const float* res_ptr;
  const int* res_idx;
  int res_rows_pb;
  int res_ld;
  // outputs
  float* out;
  int out_ld;
  int out_cols;
  int ln_width;  // features the LayerNorm statistics span (<= tile width; the rest is zero padding)
  float* agg;
  const int* agg_idx;
  int agg_rows_pb;
  // single-layer mode: rows [n_cols, 256] whose sign masks the output (ReLU backward fused into an input-gradient product)
  const float* relu_mask;
  // single-layer mode: rows [n_cols, 256] to fill with zeros on the side (the aggregate buffer of the edge update that
  // consumes these products: saves a separate fill launch per block)
  float* zero_rows;
  float* carry;  // bf16 edge update: deterministic segment sums (per-tile carry records, see below); NULL = atomics
  // training: activations saved for the backward (gw_activation_save), NULL in inference
  float* save_h;
  long long save_stride;
  int save_ld;
  float* save_y;
};

// bf16-weight launches (gw_bf16.hip): kind 0 mlp, 1 edge update, 2 node update, 3 project (grid_y slices), 4 node update + POST,
// 5 mlp + POST, 6 node update + output head.
int chain16_launch(int kind, ChainArgs& a, int k_in, int hidden, int n_out, int grid_y, void* stream);
// split-operand launches (gw_split.hip, GW_DTYPE_BF16X3): the same kinds; operands and outputs are fp32 rows only
int chainx3_launch(int kind, ChainArgs& a, int k_in, int hidden, int n_out, int grid_y, void* stream);
// gw_mlp_chain_backward: d, then n_chain masked products (w[i], mask[i] -> out[i]), then n_fan products of the last gradient
// (w[n_chain + i] -> out[n_chain + i]); all rows x 256.  fp32 streams: gw_kernels.hip; split streams (bf16x3): gw_split.hip
struct BwdChainArgs {
  const float* d;
  const void* w[5];
  const float* mask[2];
  float* out[5];
  long long n_rows;
  int d_ld, n_chain, n_fan;
  // gw_mlp_ln_chain_backward: d is the gradient at the OUTPUT of the MLP's LayerNorm; the kernel walks back through the norm first
  const float* ln_y;      // pre-LayerNorm rows [n_rows, 256] (the forward's activation save); NULL: no LayerNorm in front
  const float* ln_gamma;  // [256]
  float* ln_dgamma;       // [256] +=
  float* ln_dbeta;        // [256] +=
  float* ln_dy;           // [n_rows, 256]: gradient at the LayerNorm input (the last Linear's weight-gradient GEMM reads it)
  const float* add[5];    // fan products: rows added to the product before it is stored (NULL: none), leading dimension add_ld
  int add_ld;
  float* colsum;          // [256] += column sums of the last chain gradient (Linear_0's bias gradient); NULL: not wanted
  // LN launches may gather their input gradient: row c reads d[(c / d_idx_n) * d_tab_rows_pb + d_idx[c % d_idx_n]] (+ d_add[c])
  const int* d_idx;       // NULL: row c of d
  int d_idx_n, d_tab_rows_pb;
  const float* d_add;     // rows [n_rows, 256 (ld d_add_ld)] added to the gathered rows; NULL: none
  int d_add_ld;
  unsigned add_d_mask;    // bit p: product p (a fan product) adds the launch's own input gradient row (as gathered) before the store
};
int bwd_chainx3_launch(const BwdChainArgs& a, void* stream);
// one matrix item of gw_pack_many into the split stream (strides in floats)
void pack_x3_item(const float* w, long long stride_f, long long stride_k, int n_out, int kseg, int ntp, int nsteps, void* out, void* stream);

// row-split node update for mesh-sized launches (gw_noders.hip): CG column groups x 4 row quarters per workgroup
int node_rs_groups(int64_t n_cols);             // column groups per workgroup, 0 = not mesh-sized
bool node_rs_eligible(const ChainArgs& a);      // aggregate raw fp32 rows, node operand raw / projected / absent, one middle layer, inference
int node_rs_launch(ChainArgs& a, bool x3 /* GW_DTYPE_BF16X3 packs, else fp32 */, void* stream);

// debug timestamp hook (gw_debug_timestamps) and tuning overrides, defined in gw_kernels.hip
extern unsigned long long* g_dbg;
extern int g_dbg_cap;
extern int g_dbg_kind;
// Tuning / A-B knobs read from the environment (GW_EDGE_SKIP, GW_EDGE_IMPL, GW_STAGGER, GW_XCD_MAP, GW_EDGE_LDS_PAD, ...)
// exist only in builds made with -DGW_TUNING (scripts/gpu_tune.sh); the shipped library has the defaults compiled in and
// no code path that can produce wrong results.
#ifdef GW_TUNING
int env_int(const char* name, int fallback);
#define GW_TUNE(name, fallback) gw::env_int(name, fallback)
#define GW_SKIP(a) ((a).skip)
#define GW_TUNE_ARG(a) ((a).tune)
#else
#define GW_TUNE(name, fallback) (fallback)
#define GW_SKIP(a) 0
#define GW_TUNE_ARG(a) 0
#endif

// hipFuncSetAttribute is per device: one flag per (kernel instantiation, device) instead of one per process, so a second
// GPU driven from the same process gets its dynamic-LDS limit raised too.
struct DeviceOnce {
  bool done[64] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};
int set_error(int code, const char* msg);
int check_launch(const char* what);

// Deterministic segment sums (GW_EDGE_DETERMINISTIC): runs of a 64-column tile that continue in a neighbouring tile are not
// added to agg with atomics (whose order is not fixed) but parked in a carry record per tile; segment_fixup_launch then adds
// the records of every multi-tile segment in tile order and stores the total.  Record of tile t (kCarryFloats floats):
//   [0, 256)    partial sum of the tile's FIRST run if it continues from tile t - 1 ("open low")
//   [256, 512)  partial sum of the tile's LAST run if it continues in tile t + 1 and is not that same open-low run
//   [512]       destination row (global) of slot 0 or -1        [513] destination row of slot 1 or -1
//   [514]       1 if the tile is ONE run that is open at both ends (slot 0 holds it, the chain goes on through it)
constexpr int kCarryFloats = 528;  // 2 x 256 + header, 16-byte aligned
size_t segment_carry_bytes(int64_t n_tiles);
int segment_fixup_launch(int64_t n_tiles, const float* carry, float* agg, void* stream);

// Fast path of gw_edge_update_forward (gw_edge.hip): at most one raw operand, the others pre-projected or zero.
// Returns GW_E_UNSUPPORTED (without touching the error string) when the operand combination is not eligible.
bool edge_fast_eligible(const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in, const gw_mlp_weights* w);
int edge_fast_launch(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst, const gw_operand* x_src,
                     const gw_operand* x_dst, const gw_operand* e_in, const gw_operand* e_res, const gw_mlp_weights* w,
                     float* e_out, float* agg, int32_t n_dst, const gw_activation_save* save, float* carry /* deterministic mode */,
                     void* stream);
size_t edge_fast_carry_bytes(int32_t batch, int32_t n_edges);

// bf16 edge update with register-resident weights (gw_edge16.hip)
bool edge16_eligible(const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in, const gw_mlp_weights* w);
size_t edge16_workspace_bytes(int32_t batch, int32_t n_edges);
int edge16_launch(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst, const gw_operand* x_src,
                  const gw_operand* x_dst, const gw_operand* e_in, const gw_operand* e_res, const gw_mlp_weights* w,
                  float* e_out, void* e_out_tiles, float* agg, int32_t n_dst, void* workspace, int32_t flags /* GW_EDGE_* */,
                  void* stream);
size_t edge16_workspace_needed(int32_t batch, int32_t n_edges, const gw_operand* e_in, bool deterministic);  // layer-1 tiles (if a
                                                                                  // separate launch makes them) + carry records
// team-pipelined form of the resident kernel (gw_edge16t.hip): residual as bf16 tiles, atomics mode; gather = layer 1 is a
// gather-add done inside the kernel (every operand projected), else the layer-1 tiles come from the workspace
int edge16t_launch(const void* edge16_args /* gw16::Edge16Args */, bool gather, int n_wg, void* stream);
// processor-block form on segment-aligned tiles (gw_edge16p.hip): layer-1 tiles from the workspace, residual tiles, agg += in place
int edge16p_launch(const void* edge16_args /* gw16::Edge16Args */, int n_wg, void* stream);
int edge16_rows_to_tiles(int32_t batch, int32_t n_edges, const float* rows, int32_t rows_per_batch, int32_t ld, void* tiles,
                         void* stream);

// widths above 256 of gw_layernorm_backward / gw_relu_backward (gw_wide.hip)
int ln_bwd_wide_launch(int64_t rows, int32_t width, const float* dn, int32_t ld_dn, const float* y, int32_t ld_y, const float* gamma,
                       float* dy, int32_t ld_dy, float* dgamma, float* dbeta, void* stream);
int relu_mask_wide_launch(int64_t rows, int32_t width, const float* dh, int32_t ld_dh, const float* h, int32_t ld_h, float* dz,
                          int32_t ld_dz, float* db, void* stream);

}  // namespace gw

#endif  // GW_INTERNAL_HPP
