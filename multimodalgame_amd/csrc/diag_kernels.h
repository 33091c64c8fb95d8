// diag_kernels.h -- diagnosis only, compiled with -DMMG_ROLE_DIAG (scripts/isa_stats.py), never part of libmmg.so:
// the workgroup roles of the persistent launches as kernels of their own, so that register counts / spills / ISA size can be
// attributed to a role.
#pragma once
#include "kernels_tile.h"
#include "kernels_rc.h"

namespace mmg {
// diagnosis only (scripts/isa_stats.py -D MMG_ROLE_DIAG --kernel k_diag): every role of k_conv_persist as a kernel of its own,
// so that register counts / spills can be attributed to a role
template <int NT> __global__ __launch_bounds__(NT) void k_diag_sa(Dims dm, Params P, Tape tp, ConvArgs ar) { sa_role<NT>(dm, P, tp, ar, blockIdx.x, 0); }
template <int NT> __global__ __launch_bounds__(NT) void k_diag_sb(Dims dm, Params P, Tape tp, ConvArgs ar) { sb_role<NT>(dm, P, tp, ar, blockIdx.x, 0); }
template <int NT> __global__ __launch_bounds__(NT) void k_diag_s1(Dims dm, Params P, Tape tp, ConvArgs ar) { s1_role<NT>(dm, P, tp, ar, blockIdx.x, 0); }
template <int NT> __global__ __launch_bounds__(NT) void k_diag_s2(Dims dm, Params P, Tape tp, ConvArgs ar) { s2_role<NT>(dm, P, tp, ar, blockIdx.x, 0); }
template <int NT> __global__ __launch_bounds__(NT) void k_diag_rs256(Dims dm, Params P, Tape tp, ConvArgs ar) { rs_role<NT, 64, 100, 30, 256>(dm, P, tp, ar, blockIdx.x); }
template <int NT> __global__ __launch_bounds__(NT) void k_diag_rs0(Dims dm, Params P, Tape tp, ConvArgs ar) { rs_role<NT, 64, 100, 30, 0>(dm, P, tp, ar, blockIdx.x); }
template <int NT> __global__ __launch_bounds__(NT) void k_diag_body(Dims dm, Params P, Tape tp, ConvArgs ar) { conv_tile_body<NT, true>(dm, P, tp, ar, blockIdx.x); }
template __global__ void k_diag_sa<512>(Dims, Params, Tape, ConvArgs);
template __global__ void k_diag_sb<512>(Dims, Params, Tape, ConvArgs);
template __global__ void k_diag_s1<512>(Dims, Params, Tape, ConvArgs);
template __global__ void k_diag_s2<512>(Dims, Params, Tape, ConvArgs);
template __global__ void k_diag_rs256<512>(Dims, Params, Tape, ConvArgs);
template __global__ void k_diag_rs0<512>(Dims, Params, Tape, ConvArgs);
template __global__ void k_diag_body<512>(Dims, Params, Tape, ConvArgs);

// diagnosis only (scripts/isa_stats.py -D MMG_ROLE_DIAG --kernel k_diag_rc): the roles / bodies of k_rc_persist as kernels of their own,
// so that register counts and spills can be attributed
__global__ __launch_bounds__(256) void k_diag_rc_s1(Dims dm, Params P, Tape tp, ConvArgs ar) { rc_s1_role(dm, P, tp, ar, blockIdx.x, 0, 16); }
__global__ __launch_bounds__(256) void k_diag_rc_s2(Dims dm, Params P, Tape tp, ConvArgs ar) { rc_s2_role(dm, P, tp, ar, blockIdx.x, 0, 16); }
__global__ __launch_bounds__(256) void k_diag_rc_gru(Dims dm, Params P, Tape tp, ConvArgs ar) { RcGruW w; rc_gru_w(w, dm, P, 0); for (int t = 0; t < dm.T; ++t) rc_gru_body<true>(dm, P, tp, ar, t, blockIdx.x, 0, w); }
__global__ __launch_bounds__(256) void k_diag_rc_heads(Dims dm, Params P, Tape tp, ConvArgs ar) { RcHeadsW w; rc_heads_w(w, dm, P, tp, 0, true); for (int t = 0; t < dm.T; ++t) rc_heads_body<true>(dm, P, tp, ar, t, blockIdx.x, 0, w); }
__global__ __launch_bounds__(256) void k_diag_rc_query(Dims dm, Params P, Tape tp, ConvArgs ar) { RcQueryW w; rc_query_w(w, dm, P, 0); for (int t = 0; t < dm.T; ++t) rc_query_body<true>(dm, P, tp, ar, t, blockIdx.x, 0, 0, w); }

}  // namespace mmg
