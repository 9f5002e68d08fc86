// Internal hand-off to the kernels of the split-operand ("x3") precision mode: the public entry points (hftt_attn_fwd / hftt_attn_bwd,
// hftt_strip_linear, ...) validate a descriptor and pass the npass 2 / 4 cases on to these.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hftt_hip.h"
int hftt_x3_attn_fwd(const hftt_attn_desc& d, hipStream_t st);      // x3_attn.hip
int hftt_x3_attn_bwd(const hftt_attn_desc& d, hipStream_t st);
int hftt_x3_strip_linear(const hftt_strip_desc& d, hipStream_t st);  // x3_strip.hip
int hftt_x3_strip_mlp(const hftt_ffn_desc& d, hipStream_t st);
