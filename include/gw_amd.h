/*
 * gw_amd.h - C ABI of libgw_amd.so: the MI355X (gfx950) implementation of the message-passing hot path of
 * openclimatefix/graph_weather's GraphWeatherForecaster.
 *
 * The reference has no FFI: its boundary is the Python nn.Module API.  Each entry point below names the
 * reference statements it replaces (paths relative to the reference tree).  All functions
 *   - take plain device pointers + sizes + a hipStream_t passed as void* (no torch types),
 *   - never allocate, free or synchronise; they enqueue kernels on `stream` and return,
 *   - return 0 on success, a negative GW_E_* code otherwise (message via gw_last_error()).
 *
 * Data layout: activations row-major fp32 [rows, ld]; "tables" are [batch, rows_per_batch, ld] stacked along
 * rows (rows_per_batch == 0 means the table is shared by every batch element).  Edge lists are destination-
 * sorted int32 arrays shared by all batch elements ("shared graph" semantics, the reference's own
 * efficient_batching equivalence: tests/models/layers/test_efficient_batching.py:145).
 *
 * Weights are consumed in a packed MFMA-operand order produced by gw_pack_linear() from the reference's
 * nn.Linear layout ([out_features, in_features] row-major).
 */
#ifndef GW_AMD_H
#define GW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GW_OK 0
#define GW_E_BADARG (-1)
#define GW_E_UNSUPPORTED (-2)
#define GW_E_LAUNCH (-3)

#define GW_ABI_VERSION 19

/* dtype of the packed weight stream of an MLP (activations in HBM, accumulation, LayerNorm, residuals and segment
 * sums are always fp32). GW_DTYPE_BF16: matrix products on v_mfma_f32_16x16x32_bf16, operands rounded to bf16 (RNE)
 * in registers - BASELINE.json configs[2]. */
#define GW_DTYPE_F32 0
#define GW_DTYPE_BF16 1
/* GW_DTYPE_BF16X3 (v16): split-operand products.  Both operands of every Linear (graph_net_block.py:45-61: fp32 in the reference)
 * are carried as a pair of bf16 values, v = hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant bits, fp32 exponent
 * range), and x . w is evaluated as x_hi . w_hi + x_hi . w_lo + x_lo . w_hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation
 * (csrc/gw_split.hip): ~1e-5 per product - inside BASELINE.json's 1e-3 - at 3 bf16 MFMAs per product.  Streams from
 * gw_pack_linear_bf16x3 (4 bytes per weight).  Every table this mode reads or writes is fp32 rows (GW_LAYOUT_ROWS_F32): none of
 * the bf16 mode's 16-bit formats (edge tiles, fp16 product rows, bf16 K-order aggregates, segment-aligned tiles) applies.
 * Trains: the forward entry points accept gw_activation_save with these weights (fp32 saves: relu outputs of every Linear,
 * pre-LayerNorm rows), the backward's masked input-gradient products run through gw_mlp_ln_chain_backward /
 * gw_mlp_chain_backward_bf16x3 (gw_project_forward with relu_mask for single products) on transposed split packs and its
 * weight-gradient GEMMs through gw_gemm_f32 with GW_GEMM_TN_BF16X3; only GW_DTYPE_BF16 is inference-only.
 * Mesh-sized node updates (gw_node_update_forward with at most 12 288 rows, fp32 and bf16x3 weights alike) run on the row-split
 * kernels of csrc/gw_noders.hip - same arguments, bitwise the rows of the 64-column kernels. */
#define GW_DTYPE_BF16X3 2

/* Memory layout of a per-edge table handed to / produced by gw_edge_update_forward.
 * GW_LAYOUT_ROWS_F32: row-major fp32 rows (the reference's layout, graph_net_block.py:279-301 carries [E, D] tensors).
 * GW_LAYOUT_EDGE_TILES_BF16: the per-sample edge features of a processor stack between two blocks (bf16 mode,
 * BASELINE.json configs[2]; SURVEY.md 8(d): "halve for bf16 storage"): tiles of 64 consecutive destination-sorted edges of
 * one batch element, each tile 4 groups x 8 K-steps x 64 lanes x 8 bf16 = 32 KiB in the MFMA B-operand order
 *     byte offset ((b * ceil(E/64) + k/64) * 4 + (k%64)/16) * 8192 + s * 1024 + (16 q + k%16) * 16 + 2 i
 *     holds feature 32 s + 16 (i >> 2) + 4 q + (i & 3) of edge k of batch element b    (s < 8, q < 4, i < 8).
 * Block n writes e' in this form, block n+1 consumes it as the B operand of its layer-1 product and as its residual: half the
 * bytes of fp32 rows and every access a coalesced 1 KiB per wave instruction.  gw_edge_tiles_bytes() sizes such a buffer,
 * gw_edge_rows_to_tiles() converts rows a caller hands over. */
#define GW_LAYOUT_ROWS_F32 0
#define GW_LAYOUT_EDGE_TILES_BF16 1
/* GW_LAYOUT_ROWS_F16: layer-1 NODE PRODUCTS (X . W1_slice^T, gw_project_forward / the post products of gw_node_update_forward)
 * as fp16 rows of 256 halves (ld counts halves), values clamped to the fp16 range - bf16 mode only: the products are gathered
 * once per incident edge (~7 times on the mesh, ~77 times for a decoder source), so their bytes, not the edge features', are
 * most of what an edge update reads; fp16 keeps 11 significant bits in front of the bf16 rounding of the layer-1 activation.
 * Accepted as a projected x_src / x_dst operand by the bf16 edge update with resident weights (csrc/gw_edge16*.hip). */
#define GW_LAYOUT_ROWS_F16 2
/* GW_LAYOUT_ROWS_BF16K (v14): an AGGREGATE table (the scatter_sum of graph_net_block.py:188) as bf16 rows of 256 values in the K
 * order of the packed weight streams - position 32 s + 8 q + i of a row holds feature 32 s + 16 (i >> 2) + 4 q + (i & 3)
 * (s < 8, q < 4, i < 8; ld counts bf16 values) - i.e. exactly the 16 bytes lane q loads as its B-operand fragment of K-step s.
 * bf16 mode only: the node update rounds the aggregate to bf16 for its layer-1 product anyway (graph_net_block.py:189-190 on
 * v_mfma_f32_16x16x32_bf16), so the producer (gw_edge_update_forward with GW_EDGE_AGG_BF16K) stores what the consumer would
 * compute from fp32 rows - a quarter of the bytes written and read back (1.06 GB -> 0.27 GB per direction for the 1 degree
 * decoder at batch 16) and no conversion in the consumer.  Accepted as the raw `agg` operand of gw_node_update_forward /
 * gw_node_update_head_forward with bf16 weights. */
#define GW_LAYOUT_ROWS_BF16K 3

/* flags of gw_edge_update_forward.  GW_EDGE_DETERMINISTIC: the segment sums (scatter_sum, graph_net_block.py:188) are
 * bitwise reproducible from run to run - partial sums of segments that cross a 64-edge tile are parked in per-tile carry
 * records and added in tile order by a second small launch, instead of meeting in fp32 atomics whose order is not fixed
 * (torch_scatter's scatter_add_ on a GPU is order-nondeterministic too; this is an extra).  Needs the workspace of
 * gw_edge_update_workspace_bytes(..., flags). */
#define GW_EDGE_DETERMINISTIC 1
/* GW_EDGE_SEGMENT_TILES (v14; bf16 weights with resident kernels, every operand projected, e_res.k == 0, no e_out): `src` / `dst`
 * are a PADDED edge list of n_edges = 64 T entries in which no destination's run of edges crosses a multiple of 64 ("segment-
 * aligned tiles": assimilator_decoder.py:92-103 gives every grid node 7 or 6 consecutive edges, so 9 nodes = 63 columns fill a
 * tile).  dst[k] < 0 marks a padding column (src[k] must still be a valid row; rows of batch-shared per-edge tables are indexed by
 * the padded position k; edge tiles cover the padded list).  Every destination segment is then complete inside one tile:
 *   - e_res.k == 0 (no residual, no e_out; decoder): agg rows of destinations that have edges are WRITTEN (=, plain stores - no
 *     atomics, bitwise reproducible), rows of destinations without edges are not touched;
 *   - e_in = per-sample bf16 edge tiles (raw) with e_res = the same tiles (a processor block, graph_net_block.py:293-301): agg rows
 *     are UPDATED in place, agg[dst] += sum of LayerNorm(.) over the destination's edges WITHOUT the residual - the caller hands
 *     over the previous block's aggregate, whose rows are the segment sums of this block's residual (e of block n = e' of block
 *     n - 1), so the buffer always holds sum(e') (plain load / add / store; at most 16 destinations per tile);
 *   - every operand projected with a batch-shared residual (first processor block): as without the flag (agg += by atomics on a
 *     zero fill), the padding columns are skipped.
 * GW_EDGE_AGG_BF16K (with GW_EDGE_SEGMENT_TILES, no residual): `agg` points to bf16 rows in K order (GW_LAYOUT_ROWS_BF16K).
 * GW_EDGE_SEGMENT_SPLIT (with GW_EDGE_SEGMENT_TILES, no residual, fp32 agg): a destination with more than 64 edges (a polar
 * mesh cell of the encoder graph, encoder.py:75-104) is allowed: its run starts a fresh tile and continues over whole tiles; the
 * pieces are partial sums that are ADDED with fp32 atomics, all other rows are written - agg must be zero-filled by the caller. */
#define GW_EDGE_SEGMENT_TILES 2
#define GW_EDGE_AGG_BF16K 4
#define GW_EDGE_SEGMENT_SPLIT 8

/* Library / ABI version and last error text (thread local). */
int gw_version(void);
const char* gw_last_error(void);
/* Debug/profiling aid (no reference counterpart): while `buffer` != NULL, launches of family `kind` (0 mlp, 1 edge
 * update, 2 node update, 3 project) write 16 x uint64 per workgroup: s_memtime stamps of the kernel phases [0..6],
 * HW_ID [8], XCC_ID [9], block index [10].  buffer: device memory, capacity_workgroups * 16 * 8 bytes. */
int gw_debug_timestamps(void* buffer, int capacity_workgroups, int kind);

/* ---- weight packing ------------------------------------------------------------------------------------
 * Packed size in floats of the [k_lo, k_hi) column slice of an nn.Linear weight with `n_out` rows, padded to
 * 16-row tiles / 16-column groups (graph_net_block.py:45-49 creates the Linear layers being packed). */
size_t gw_packed_floats(int n_out, int k_lo, int k_hi);
/* w: device pointer to [n_out, k_total] fp32 (nn.Linear.weight); out: gw_packed_floats() floats. */
int gw_pack_linear(const float* w, int n_out, int k_total, int k_lo, int k_hi, float* out, void* stream);
/* bf16 form of the packed stream (byte size / packing); same arguments, out receives gw_packed_bytes_bf16() bytes. */
size_t gw_packed_bytes_bf16(int n_out, int k_lo, int k_hi);
int gw_pack_linear_bf16(const float* w, int n_out, int k_total, int k_lo, int k_hi, void* out, void* stream);
/* (v16) split form (GW_DTYPE_BF16X3): per 32-wide K-step the bf16 hi fragments of all row tiles, then their lo fragments;
 * out receives gw_packed_bytes_bf16x3() bytes (twice the bf16 stream). */
size_t gw_packed_bytes_bf16x3(int n_out, int k_lo, int k_hi);
int gw_pack_linear_bf16x3(const float* w, int n_out, int k_total, int k_lo, int k_hi, void* out, void* stream);
/* Zero-pad a vector (bias / LayerNorm gamma, beta) to a multiple of 32 floats. out has gw_padded_n(n). */
int gw_padded_n(int n);
int gw_pad_vector(const float* v, int n, float* out, void* stream);
/* (v13) Every matrix slice and vector of one MLP (or any other set) packed by ONE launch: what MLP.__init__ creates
 * (graph_net_block.py:45-61) is 3-5 Linear layers and a LayerNorm, i.e. ~10 gw_pack_linear / gw_pad_vector launches of a few
 * microseconds each per MLP and weight version - 27 MLPs per forecaster, every training step.  A matrix item is addressed by
 * strides, element (f, k) of the slice = w[f * stride_f + k * stride_k], so a column slice of nn.Linear.weight is
 * {w + k_lo, k_total, 1} and the transposed block the backward's input-gradient products stream ("Linear" that maps
 * gradients back, W[:, lo:hi]^T) is {w + lo, 1, k_total} - no transposed copy is materialised.  Rows n_out.. of the last
 * 16-row tile quad and input features kseg.. of the last K-step are zero.  The item arrays are HOST memory (copied into the
 * launch); at most GW_PACK_MAX_ITEMS matrices and GW_PACK_MAX_ITEMS vectors per call. */
#define GW_PACK_MAX_ITEMS 16
typedef struct gw_pack_item {
  const float* w;      /* device pointer to element (0, 0) of the slice */
  int64_t stride_f;    /* floats between consecutive output features (rows of nn.Linear.weight: k_total) */
  int64_t stride_k;    /* floats between consecutive input features (1) */
  int32_t n_out;       /* output features of the slice */
  int32_t kseg;        /* input features of the slice */
  int32_t rows;        /* rows of the packed stream, >= n_out (0 = n_out): an output head with n_out < 80 features is packed as
                          80 rows, the tile count its kernel variant walks; rows n_out.. are zero */
  int32_t reserved;
  void* out;           /* gw_packed_floats(rows, 0, kseg) floats, or gw_packed_bytes_bf16 / _bf16x3(rows, 0, kseg) bytes */
} gw_pack_item;
typedef struct gw_pad_item {
  const float* v;      /* device pointer, n floats */
  int32_t n;
  int32_t n_out;       /* entries written, >= n (0 = n), rounded up to gw_padded_n(); entries n.. are zero */
  float* out;          /* gw_padded_n(max(n, n_out)) floats */
} gw_pad_item;
int gw_pack_many(int32_t weight_dtype, int32_t n_mats, const gw_pack_item* mats, int32_t n_vecs, const gw_pad_item* vecs,
                 void* stream);

/* One input operand of a fused MLP ("segment" of the concatenated input). */
typedef struct gw_operand {
  const float* ptr;     /* table base                                                          */
  const int32_t* index; /* per-column row index within a batch element, NULL = identity         */
  int32_t rows_per_batch; /* rows of the table per batch element, 0 = table shared by the batch */
  int32_t ld;           /* row stride in floats                                                 */
  int32_t k;            /* valid input features taken from each row (0 = operand is all zeros:  */
                        /* its weight slice is skipped, exact since 0*W == 0)                    */
  int32_t projected;    /* 1 = rows already hold X . W1_slice^T (from gw_project_forward): they */
                        /* are gather-added into the layer-1 accumulator, no MFMA pass           */
  int32_t layout;       /* GW_LAYOUT_*: ROWS_F32 everywhere except the e_in / e_res operands of  */
                        /* gw_edge_update_forward, which may be EDGE_TILES_BF16 (ptr = tile      */
                        /* buffer; rows_per_batch = n_edges: one tile set per batch element, 0:  */
                        /* one set shared by the batch; ld / index unused, k = 256)              */
} gw_operand;

/* A 3+ layer MLP in packed form: Linear(k_in,h) ReLU [Linear(h,h) ReLU]*n_mid Linear(h,n_out) [LayerNorm]. */
typedef struct gw_mlp_weights {
  const float* w1[3];  /* packed slices of layer-1 weight, one per operand (NULL if operand k == 0) */
  const float* b1;     /* padded bias [h]                                                           */
  const float* w_mid;  /* n_mid packed [h,h] matrices, contiguous                                   */
  const float* b_mid;  /* n_mid padded biases                                                       */
  const float* w_out;  /* packed [n_out, h]                                                         */
  const float* b_out;  /* padded bias                                                               */
  const float* ln_gamma; /* padded, NULL = no LayerNorm (eps = 1e-5, biased variance)               */
  const float* ln_beta;
  int32_t hidden;      /* 128 or 256 */
  int32_t n_mid;       /* hidden_layers - 1 */
  int32_t n_out;       /* 256, or <= 80 for the decoder head */
  int32_t weight_dtype; /* GW_DTYPE_F32: streams from gw_pack_linear; GW_DTYPE_BF16: gw_pack_linear_bf16; GW_DTYPE_BF16X3: gw_pack_linear_bf16x3 */
  int32_t ln_width;    /* features LayerNorm normalises over; 0 = n_out.  Narrower models (node/edge width < 256,
                          graph_net_block.py:234-244 defaults to 128) run zero-padded to 256: their statistics then span
                          the first ln_width features only (fp32 weights) */
  int32_t k_in;        /* (v15) input features the layer-1 slices were packed for, summed over the operands; 0 = not recorded (MANDATORY - as is
                          out_rows - for the `head` argument of gw_node_update_head_forward, which rejects a zeroed struct).
                          Entry points that stream a FIXED number of K-steps from w1 (gw_node_update_head_forward: the
                          head's 256 inputs) refuse anything else instead of reading past the packed buffer */
  int32_t out_rows;    /* (v15) rows w_out / b_out were packed with (gw_pack_linear rows > n_out: zero rows of an output
                          head); 0 = gw_padded_n(n_out).  gw_node_update_head_forward requires 80 */
} gw_mlp_weights;

struct gw_activation_save; /* training only, defined with the backward entry points below; NULL in inference */

/* ---- MLP.forward (graph_net_block.py:63-77) applied to rows --------------------------------------------
 * y[c, :] = MLP(x[c, :k]) (+ residual[c, :n_out]);   c in [0, n_rows).
 * Used for Encoder.node_encoder (encoder.py:205), the three edge encoders (encoder.py:206-208,235-241,
 * assimilator_decoder.py:175-177) and AssimilatorDecoder.node_decoder + the Decoder residual
 * (assimilator_decoder.py:197, decoder.py:93).  Supported k: <= 16, <= 112, or exactly 256. */
int gw_mlp_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_mlp_weights* w,
                   const gw_operand* residual /* may be NULL */, float* out, int32_t out_ld,
                   const struct gw_activation_save* save /* may be NULL */, void* stream);

/* The same MLP whose output rows y are, while still in registers, multiplied by n_post (1..4) packed [256, 256] slices:
 * post_out[s][c] = y[c] . post_w[s]^T (rows of 256; post_layout GW_LAYOUT_ROWS_F32 / _F16) - Encoder.node_encoder on the grid
 * rows (encoder.py:205) followed by the x[row] slice of the encoder block's edge MLP layer 1 (graph_net_block.py:131-134: one
 * edge per grid node, so the product costs what the raw operand would cost inside the edge kernel).  out may be NULL when only
 * the products are wanted (the encoder drops the grid rows, encoder.py:219-223).  bf16 / bf16x3 weights, hidden 256, 256 outputs with
 * LayerNorm, 33..128 input features. */
int gw_mlp_post_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_mlp_weights* w,
                        float* out /* may be NULL */, int32_t out_ld, int32_t n_post, const float* const* post_w,
                        void* const* post_out, int32_t post_layout, void* stream);

/* (v13) Backward of the Linear / ReLU chain of an MLP (graph_net_block.py:45-61; what autograd runs for loss.backward(),
 * train/run.py:517-519, as one addmm-backward + threshold_backward pair per layer) in one launch, register-resident like the
 * forward: d_0 = d [n_rows, 256 (ld d_ld)];  d_{i+1} = (d_i . W_i) * (mask_i > 0) for i < n_chain (<= 2), each stored to
 * chain_out[i] [n_rows, 256] (the weight-gradient products read them); then fan_out[s] = d_{n_chain} . Wf_s for s < n_fan (<= 3):
 * the input gradients of the first Linear's operand blocks.  chain_w / fan_w: packed TRANSPOSED 256 x 256 blocks
 * (gw_pack_many with {w + lo, 1, k_total}); chain_mask[i]: the ReLU output that fed layer i [n_rows, 256] (gw_activation_save).
 * fp32 streams. */
int gw_mlp_chain_backward(int64_t n_rows, const float* d, int32_t d_ld, int32_t n_chain, const float* const* chain_w,
                          const float* const* chain_mask, float* const* chain_out, int32_t n_fan, const float* const* fan_w,
                          float* const* fan_out, void* stream);
/* (v18) The same chain on split operands (mixed-precision training, GW_DTYPE_BF16X3): chain_w / fan_w are the packed transposed
 * blocks in the split stream (gw_pack_many with GW_DTYPE_BF16X3); every product is the forward's three bf16 MFMAs on (hi, lo)
 * pairs, the gradient rows stay in registers between the products.  d, masks and outputs are fp32 rows as above. */
int gw_mlp_chain_backward_bf16x3(int64_t n_rows, const float* d, int32_t d_ld, int32_t n_chain, const void* const* chain_w,
                                 const float* const* chain_mask, float* const* chain_out, int32_t n_fan, const void* const* fan_w,
                                 float* const* fan_out, void* stream);

/* (v19) native_layer_norm_backward + that chain in one launch (what autograd runs behind MLP.forward's LayerNorm,
 * graph_net_block.py:59-61, under loss.backward()): dn [n_rows, 256 (ld dn_ld)] is the gradient at the OUTPUT of the LayerNorm
 * (width 256, eps 1e-5, biased variance), y the saved pre-norm rows [n_rows, 256]; dy [n_rows, 256] receives the gradient at the
 * norm's input (the last Linear's weight-gradient product reads it) and is the chain's d_0 without being read back;
 * dgamma / dbeta [256] are accumulated (+=).  y == NULL: no LayerNorm, dn is d_0 (gamma, dgamma, dbeta, dy unused).
 * With the LayerNorm the input gradient may be GATHERED by the launch (the index_select backward of scatter_sum,
 * graph_net_block.py:188: every edge row receives its destination's row of the aggregate's gradient): dn_idx [dn_idx_n] (NULL:
 * row c reads row c of dn) - row c = b * dn_idx_n + k reads dn[b * dn_table_rows_pb + dn_idx[k]] - plus row c of dn_add
 * [n_rows, 256 (ld dn_add_ld)] (may be NULL: the gradient of e' where the block exposes it); the [n_rows, 256] table of gathered
 * rows is then never written.
 * The chain arguments are those of gw_mlp_chain_backward_bf16x3, plus three things autograd does with separate kernels:
 * dz_colsum [256] (may be NULL) += column sums of the last chain gradient - the first Linear's bias gradient
 * (addmm backward's sum over rows); fan_add[s] (array or entries may be NULL; rows [n_rows, 256 (ld fan_add_ld)]) is added
 * to fan_out[s] before it is stored - the gradient of a tensor that is both an operand of the first Linear and the block's
 * residual (EdgeProcessor: graph_net_block.py:131-137) arrives as one row; bit s of fan_add_dn_mask: that joining gradient is
 * the launch's own input gradient row (dn as gathered).
 * weight_dtype: GW_DTYPE_F32 or GW_DTYPE_BF16X3 - the dtype of the packed streams. */
int gw_mlp_ln_chain_backward(int32_t weight_dtype, int64_t n_rows, const float* dn, int32_t dn_ld, const int32_t* dn_idx,
                             int32_t dn_idx_n, int32_t dn_table_rows_pb, const float* dn_add, int32_t dn_add_ld, const float* y,
                             const float* gamma, float* dgamma, float* dbeta, float* dy, int32_t n_chain, const void* const* chain_w,
                             const float* const* chain_mask, float* const* chain_out, float* dz_colsum, int32_t n_fan,
                             const void* const* fan_w, float* const* fan_out, const float* const* fan_add, int32_t fan_add_ld,
                             uint32_t fan_add_dn_mask, void* stream);

/* ---- layer-1 split: cat[x_s, x_d, e] . W1^T == x_s . Ws^T + x_d . Wd^T + e . We^T ------------------------
 * (graph_net_block.py:131-134 concatenates and multiplies; the products over node tables are shared by the ~7
 * edges incident to a node, and the ones over batch-independent tables are cacheable.)
 * out_s[c, :] = x[c, :256] . W_s^T for s < n_slices (<= 4); W_s = packed [256, 256] slices from gw_pack_linear. */
int gw_project_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, int32_t n_slices,
                       const float* const* w_slices, void* const* outs, int32_t out_ld,
                       int32_t out_layout /* GW_LAYOUT_ROWS_F32, or GW_LAYOUT_ROWS_F16 (bf16 slices; out_ld in halves) */,
                       int32_t weight_dtype /* GW_DTYPE_* of the slices */,
                       const float* relu_mask /* NULL, or [n_rows, 256]: out *= (relu_mask > 0) - ReLU backward fused into an
                                                  input-gradient product (backward pass; single fp32 slice) */,
                       float* zero_rows /* NULL, or [n_rows, 256] filled with zeros on the side: the aggregate buffer of the
                                           edge update that consumes these products (saves a fill launch; fp32) */,
                       void* stream);

/* ---- EdgeProcessor.forward + scatter_sum (graph_net_block.py:131-137 and :188) --------------------------
 * For every batch element b and edge e (dst-sorted):
 *   e_new = LN(MLP(cat[x_src[b, src[e]], x_dst[b, dst[e]], e_in[b, e]])) + e_res[b, e]
 *   agg[b, dst[e], :] += e_new                         (agg must be zero-filled by the caller)
 * and, if e_out != NULL, e_out[b, e, :] = e_new.  Feature width is 256.  Each of x_src / x_dst / e_in may be
 * raw rows, pre-projected rows (operand.projected) or zeros (k == 0); e_res is the raw edge feature row (the residual of
 * graph_net_block.py:135) - or k == 0 for "none": a caller that wants only the aggregate of batch-shared edge features (the
 * decoder, assimilator_decoder.py:195 drops e') may add their per-destination sums into agg beforehand instead,
 * sum(LN(.) + e) = sum(LN(.)) + sum(e)  (bf16 weights with resident kernels in atomics mode, or bf16x3 weights; e_out == NULL, no save). */
int gw_edge_update_forward(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst,
                           const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in,
                           const gw_operand* e_res, const gw_mlp_weights* w,
                           void* e_out /* NULL, or fp32 rows [batch*n_edges,256], or bf16 edge tiles (e_out_layout) */,
                           int32_t e_out_layout /* GW_LAYOUT_* of e_out */, float* agg /* [batch*n_dst,256] (GW_EDGE_AGG_BF16K: bf16) */,
                           int32_t n_dst, const struct gw_activation_save* save /* may be NULL */,
                           void* workspace /* may be NULL */, size_t workspace_bytes, int32_t flags /* GW_EDGE_* */, void* stream);
/* Edge tiles (GW_LAYOUT_EDGE_TILES_BF16) are consumed and produced by the bf16 path with register-resident weights only:
 * bf16 weights, one middle layer, x_src / x_dst pre-projected or zero, and the workspace of gw_edge_update_workspace_bytes. */
size_t gw_edge_tiles_bytes(int32_t batch, int32_t n_edges);
/* rows [batch (rows_per_batch > 0) or shared (0)][n_edges, ld] fp32 -> tiles (bf16, round to nearest even; padding edges 0). */
int gw_edge_rows_to_tiles(int32_t batch, int32_t n_edges, const float* rows, int32_t rows_per_batch, int32_t ld, void* tiles,
                          void* stream);
/* Scratch device memory (16-byte aligned, contents irrelevant) gw_edge_update_forward needs for these operands and flags:
 * the bf16 path with a raw edge operand (bf16 tiles) stages the layer-1 activations of all tiles between its two launches
 * (32 KiB per 64 edges and batch element); deterministic segment sums park one carry record per tile.  0 = none needed
 * (e.g. the bf16 path whose operands are all projected gathers them inside its one persistent launch); the library never
 * allocates (the caller's allocator owns all device memory). */
size_t gw_edge_update_workspace_bytes(int32_t batch, int32_t n_edges, const gw_operand* x_src, const gw_operand* x_dst,
                                      const gw_operand* e_in, const gw_mlp_weights* w, int32_t flags);

/* ---- NodeProcessor.forward after aggregation (graph_net_block.py:189-191) -------------------------------
 *   x_new[b, j] = LN(MLP(cat[x[b, j], agg[b, j]])) + x_res[b, j]
 * x may be raw, pre-projected or zeros (k == 0: the decoder's lat/lon rows are zeros, assimilator_decoder.py:84,
 * 190 - the x-slice of layer 1 is skipped); x_res = raw node rows for the residual (NULL or k == 0: none).  With bf16
 * weights a pre-projected x may be fp16 product rows (GW_LAYOUT_ROWS_F16, ld in halves); every other operand of the row-wise
 * entry points is fp32 rows - any other layout is rejected; agg may also be bf16 rows in K order (GW_LAYOUT_ROWS_BF16K, bf16
 * weights, ld in bf16 values, no index). */
int gw_node_update_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_operand* x_res,
                           const gw_operand* agg, const gw_mlp_weights* w, float* x_out, int32_t out_ld,
                           const struct gw_activation_save* save /* may be NULL */,
                           int32_t n_post /* 0..4 */, const float* const* post_w, void* const* post_out,
                           int32_t post_layout /* GW_LAYOUT_ROWS_F32, or GW_LAYOUT_ROWS_F16 (bf16 weights) */,
                           float* zero_rows /* may be NULL */, void* stream);
/* n_post > 0: while the new rows are still in registers they are multiplied by n_post packed [256, 256] slices (packing and
 * dtype of w) and the products written to post_out[s] [n_rows, 256]: post_out[s][j] = x_new[j] . post_w[s]^T - the layer-1
 * products of the NEXT block's edge MLP over the node table (see gw_project_forward), so that GraphProcessor's loop
 * (graph_net_block.py:293-301) needs no projection launch between blocks.  zero_rows: [n_rows, 256] filled with zeros on the
 * side (the next block's aggregate buffer).  Inference only (save must be NULL), out_ld 256. */

/* (v17) Which kernel form gw_node_update_forward gives a launch of n_rows rows with fp32 / bf16x3 weights (fp32 row tables,
 * one middle layer, LayerNorm over 256 features or none, no activation saves): 1, 2 or 3 = the row-split kernels of
 * csrc/gw_noders.hip with that many 16-column groups per workgroup (mesh-sized launches: at most one round of 48-column
 * workgroups on the 256 CUs), 0 = the 64-column kernels.  Pure host logic (no GPU needed): graph_weather_amd/routes.py keeps the
 * same table and tests/test_routes.py compares the two. */
int gw_node_update_row_split_groups(int64_t n_rows);

/* ---- NodeProcessor.forward followed by the output head, one launch (fp32 since v17; bf16; bf16x3) ---------------------
 * AssimilatorDecoder.forward after its edge update (assimilator_decoder.py:195-200) + the Decoder residual (decoder.py:93):
 *   x_new[j] = LN(MLP_node(cat[x[j], agg[j]]))            (graph_net_block.py:189-191; the decoder's rows are zeros: no x_res)
 *   out[j, :n] = MLP_head(x_new[j]) + residual[j, :n]     (node_decoder 256 -> 128 -> 128 -> n <= 80 features, no norm)
 * x_new stays in registers: the [rows, 256] table between the two MLPs is neither written nor read.
 * Packed sizes the kernel streams unconditionally - checked through head->k_in == 256 and head->out_rows == 80 (v15):
 * head->w1[0] = [128, 256] slice (8 K-steps x 8 row tiles), head->w_mid = [128, 128], head->w_out / b_out packed with rows = 80.
 * Both MLPs in the same weight dtype.  fp32: the same arithmetic in the same order as gw_node_update_forward followed by
 * gw_mlp_forward with a residual - bitwise their result (tests/test_gpu_round6.py). */
int gw_node_update_head_forward(int64_t n_rows, int32_t rows_per_batch, const gw_operand* x, const gw_operand* agg,
                                const gw_mlp_weights* w, const gw_mlp_weights* head, const gw_operand* residual /* may be NULL */,
                                float* out, int32_t out_ld, void* stream);

/* ---- NormalizedMSELoss.forward (losses.py:66-94, normalize on/off) ---------------------------------------
 * loss = mean_{b,n}( w_lat[n / num_lon] * mean_c( (pred-target)^2 [/ var_c] ) ); *loss_out must be zeroed. */
int gw_normalized_mse_forward(const float* pred, const float* target,
                              const float* inv_var /* NULL, [c] (inv_var_full 0) or [batch, nodes, c] (inv_var_full 1) */,
                              int32_t inv_var_full, const float* lat_weights, int32_t num_unique_lat, int32_t batch, int32_t nodes,
                              int32_t channels, float* loss_out, void* stream);

/* =====================================================================================================================
 * Backward / training step (reference: autograd through the same modules + torch.optim.AdamW, train/run.py:509-521;
 * SURVEY.md 8f row 1).  The forward entry points above take an optional gw_activation_save: when given, the fused
 * kernels also write the activations autograd would have saved; the backward is then composed from the generic
 * kernels below (all fp32, matrix products on the fp32 MFMA like the forward).
 * ===================================================================================================================== */

/* Activations written by a forward call for its backward (all row-major fp32, one row per column of the launch):
 * hidden[l] = relu output of Linear l (l = 0 .. n_mid), hidden_ld floats per row (>= hidden width);
 * pre_norm = output of the last Linear before LayerNorm (NULL when the MLP has no norm). */
typedef struct gw_activation_save {
  float* hidden;        /* [(n_mid + 1), n_rows, hidden_ld] */
  int64_t hidden_stride; /* floats between consecutive hidden layers */
  int32_t hidden_ld;
  float* pre_norm;      /* [n_rows, 256] ([n_rows, 80] for an output head with n_out <= 80, zero padded) or NULL */
} gw_activation_save;

#define GW_GEMM_NN 0 /* C[m][n]  = sum_k A[m][k] * B[k][n]   (input gradients:  dX = dZ . W)                         */
#define GW_GEMM_TN 1 /* C[m][n] += sum_k A[k][m] * B[k][n]   (weight gradients: dW += dZ^T . X; C must hold the sum) */
#define GW_GEMM_TN_BF16X3 2 /* (v16) GW_GEMM_TN with split-operand products (both operands as bf16 hi / lo pairs, three bf16 MFMAs per
                               product, fp32 accumulate: the weight-gradient GEMMs of the mixed-precision training step); any
                               shape, leading dimension and alignment (v19; before: multiples of 128 only, other shapes on
                               the fp32 kernel) */
int gw_gemm_f32(int32_t mode, int64_t m, int32_t n, int64_t k, const float* a, int32_t lda, const float* b, int32_t ldb,
                float* c, int32_t ldc, float* colsum_a /* TN only, may be NULL: colsum_a[m] += sum_k A[k][m] (bias gradient) */,
                void* stream);
/* nn.ReLU backward fused with the nn.Linear bias gradient: dz = dh * (h > 0) (h NULL: dz = dh), db[c] += sum_r dz[r][c].
 * dz may alias dh or be NULL (bias gradient only); db may be NULL.  Any width (above 256: wide models, csrc/gw_wide.hip). */
int gw_relu_backward(int64_t rows, int32_t width, const float* dh, int32_t ld_dh, const float* h, int32_t ld_h, float* dz,
                     int32_t ld_dz, float* db, void* stream);
/* nn.LayerNorm(width, eps 1e-5) backward from the saved pre-norm rows y: dy; dgamma += , dbeta += (may be NULL).
 * width 256: the message-passing MLPs; other widths <= 256: LayerNorm on an output head (regional_forecast.py:223-230);
 * 257..4096: wide models. */
int gw_layernorm_backward(int64_t rows, int32_t width, const float* dn, int32_t ld_dn, const float* y, int32_t ld_y,
                          const float* gamma, float* dy, int32_t ld_dy, float* dgamma, float* dbeta, void* stream);
/* Dual of the segment sum (graph_net_block.py:188): out[b, k, :] = table[b, idx[k], :] (+ add[b, k, :]); 256-float rows;
 * idx NULL = identity; rows_per_batch 0 = table shared by the batch. */
int gw_gather_rows(int32_t batch, int32_t n_idx, const float* table, int32_t rows_per_batch, const int32_t* idx,
                   const float* add, float* out, void* stream);
/* Dual of the x[row] / x[col] gathers (MetaLayer, graph_net_block.py:221-228):
 * out[bo, n, :] (+)= sum_{b} sum_{i in [ptr[n], ptr[n+1])} rows[b, perm[i], :]; perm NULL = identity;
 * batch_out == batch: per sample; batch_out == 1: summed over the batch too (gradient of a batch-shared table). */
int gw_segment_sum_rows(int32_t batch, int32_t batch_out, int32_t n_seg, const float* rows, int32_t rows_per_batch_in,
                        const int32_t* perm, const int32_t* ptr, float* out, int32_t accumulate, void* stream);
/* NormalizedMSELoss backward (losses.py:66-94): dpred = dloss * d loss / d pred. */
int gw_normalized_mse_backward(const float* pred, const float* target, const float* inv_var, int32_t inv_var_full,
                               const float* lat_weights,
                               int32_t num_unique_lat, int32_t batch, int32_t nodes, int32_t channels, const float* dloss,
                               float* dpred, void* stream);
/* torch.optim.AdamW step (decoupled weight decay, bias correction with `step` >= 1) on a flat fp32 buffer. */
int gw_adamw_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* BoundaryNudgingLayer (regional_forecast.py:44-132) on rows of in = [regional (feat) | global context (feat) | prior (1)]:
 * alpha = clamp(prior + w2 . relu(W1 . in + b1) + b2, 0, 1); out = (1 - alpha) regional + alpha context.
 * w1t = W1^T ([2 feat + 1, hidden] row-major), w2 [hidden], b2 [1]; feat, hidden <= 256. */
int gw_nudging_forward(int64_t rows, int32_t feat, int32_t hidden, const float* in, int32_t ld_in, const float* w1t, const float* b1,
                       const float* w2, const float* b2, float* out, void* stream);
/* Its backward for dout [rows, feat]: d_in[:, 0:feat] = gradient of the regional columns (other columns untouched),
 * dz [rows, hidden] = gradient at the hidden pre-activations, hid [rows, hidden] = relu(hidden), dcorr [rows] = gradient at the
 * MLP output; the parameter gradients are then dW1 += dz^T in, db1 += colsum dz, dw2 += dcorr^T hid, db2 += sum dcorr
 * (gw_gemm_f32 TN).  w1 = W1 ([hidden, 2 feat + 1] row-major). */
int gw_nudging_backward(int64_t rows, int32_t feat, int32_t hidden, const float* in, int32_t ld_in, const float* w1, const float* w1t,
                        const float* b1, const float* w2, const float* b2, const float* dout, float* d_in, int32_t ld_din, float* dz,
                        float* hid, float* dcorr, void* stream);

/* =====================================================================================================================
 * Models wider than 256 features (the reference's training script builds 1024-wide ones, train/run.py:493-497) run layer
 * by layer on the generic kernels below (csrc/gw_wide.hip) instead of the fused ones: same arithmetic as graph_net_block.py,
 * nothing fused across layers.  All fp32 row-major, any width.
 * ===================================================================================================================== */
/* nn.Linear (+ nn.ReLU): out[r, :n] = act(x[r, :k] . w^T + bias); w = nn.Linear.weight as it lies in memory ([n, k], row
 * stride ldw); bias may be NULL; relu 0 / 1.  fp32 MFMA, fp32 accumulate. */
int gw_linear_forward(int64_t rows, int32_t k, int32_t n, const float* x, int32_t ldx, const float* w, int32_t ldw, const float* bias,
                      int32_t relu, float* out, int32_t ldo, void* stream);
/* The same Linear with up to three row tables added before the activation:
 *   out[m] = act(x[m] . w^T + bias + sum_i table_i[b * rows_pb_i + idx_i[k]]),  (b, k) = (m / rows_per_batch, m % rows_per_batch)
 * (idx_i NULL: k itself; rows_pb_i 0: table shared by the batch).  This is the layer-1 split of the fused kernels for wide
 * models - cat[x_s, x_d, e] . W1^T = (x_s . Ws^T)[src] + (x_d . Wd^T)[dst] + e . We^T: node products are made once per node and
 * gathered per edge in the epilogue of the edge-level product.  k == 0 (x, w NULL): no product, only bias + gathered rows. */
int gw_linear_gather_forward(int64_t rows, int32_t rows_per_batch, int32_t k, int32_t n, const float* x, int32_t ldx, const float* w,
                             int32_t ldw, const float* bias, int32_t n_add, const float* const* add_table, const int32_t* const* add_idx,
                             const int32_t* add_ld, const int32_t* add_rows_pb, int32_t relu, float* out, int32_t ldo, void* stream);
/* nn.LayerNorm(width, eps 1e-5, affine) (+ res, the residual add of graph_net_block.py:135/:191; res_period > 0: residual rows
 * shared by the batch, row m reads res[m % res_period]); width <= 4096. */
int gw_layernorm_forward(int64_t rows, int32_t width, const float* y, int32_t ld_y, const float* gamma, const float* beta,
                         const float* res, int32_t ld_res, int64_t res_period, float* out, int32_t ld_out, void* stream);
/* out = a + b (residual add behind an MLP without norm). */
int gw_add_rows(int64_t rows, int32_t width, const float* a, int32_t lda, const float* b, int32_t ldb, float* out, int32_t ldo,
                void* stream);
/* gw_gather_rows / gw_segment_sum_rows for rows of any width (row strides ld / ldo); the segment sum walks the CSR in one
 * fixed order (no atomics: bitwise reproducible). */
int gw_gather_rows_wide(int32_t batch, int32_t n_idx, int32_t width, const float* table, int32_t ld, int32_t rows_per_batch,
                        const int32_t* idx, float* out, int32_t ldo, void* stream);
int gw_segment_sum_rows_wide(int32_t batch, int32_t batch_out, int32_t n_seg, int32_t width, const float* rows, int32_t ld,
                             int32_t rows_per_batch_in, const int32_t* perm, const int32_t* ptr, float* out, int32_t ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GW_AMD_H */
