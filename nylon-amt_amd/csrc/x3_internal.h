// Internal hand-off to the kernels of the split-operand ("x3") precision mode: the public entry points (hftt_attn_fwd / hftt_attn_bwd,
// hftt_strip_linear, ...) validate a descriptor and pass the npass 2 / 4 cases on to these.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hftt_hip.h"
int hftt_x3_attn_fwd(const hftt_attn_desc& d, hipStream_t st);      // x3_attn.hip
int hftt_x3_attn_bwd(const hftt_attn_desc& d, hipStream_t st);
int hftt_x3p_attn_bwd(const hftt_attn_desc& d, hipStream_t st);          // x3_attn_pl.hip: the backward on f16-pair planes
int hftt_x3p_attn_fwd_try(const hftt_attn_desc& d, hipStream_t st);      // x3_attn_pl.hip: -1 when the operands are not f16-pair planes
int hftt_x3_strip_linear(const hftt_strip_desc& d, hipStream_t st);  // x3_strip.hip
int hftt_x3_strip_mlp(const hftt_ffn_desc& d, hipStream_t st);
int hftt_x3_attn_out_ffn(const hftt_strip_desc& o, const hftt_ffn_desc& d, hipStream_t st);

// The "all-bf16 stream" form of the attention kernels (bf16 mode): q, k, v, out stored as bf16 with every row / sequence stride a multiple
// of 8 elements.  ONE predicate for attn_fwd.hip, attn_fwd8.hip and attn_bwd.hip: in this form lse[0] holds the RAW row maximum (the
// scaled maximum otherwise), so forward and backward must agree on it for every descriptor.
bool hftt_attn_hb_form(const hftt_attn_desc& d);
