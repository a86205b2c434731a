// gw_internal.hpp - host-side declarations shared between the translation units of libgw_amd.so (not part of the ABI).
#ifndef GW_INTERNAL_HPP
#define GW_INTERNAL_HPP

#include <stdint.h>

#include "../../include/gw_amd.h"

namespace gw {

// debug timestamp hook (gw_debug_timestamps) and tuning overrides, defined in gw_kernels.hip
extern unsigned long long* g_dbg;
extern int g_dbg_cap;
extern int g_dbg_kind;
int env_int(const char* name, int fallback);
int set_error(int code, const char* msg);
int check_launch(const char* what);

// Fast path of gw_edge_update_forward (gw_edge.hip): at most one raw operand, the others pre-projected or zero.
// Returns GW_E_UNSUPPORTED (without touching the error string) when the operand combination is not eligible.
bool edge_fast_eligible(const gw_operand* x_src, const gw_operand* x_dst, const gw_operand* e_in, const gw_mlp_weights* w);
int edge_fast_launch(int32_t batch, int32_t n_edges, const int32_t* src, const int32_t* dst, const gw_operand* x_src,
                     const gw_operand* x_dst, const gw_operand* e_in, const gw_operand* e_res, const gw_mlp_weights* w,
                     float* e_out, float* agg, int32_t n_dst, void* stream);

}  // namespace gw

#endif  // GW_INTERNAL_HPP
