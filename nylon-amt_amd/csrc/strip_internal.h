// Internal hand-off between the two strip kernel files: the entry points in strip_gemm.hip validate a descriptor, then offer it to
// the persistent software-pipelined kernels of strip_gemm2.hip; -1 = "not covered, launch the general kernel".
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hftt_hip.h"
int hftt_strip_linear2_try(const hftt_strip_desc& d, hipStream_t st);
int hftt_strip_mlp2_try(const hftt_ffn_desc& d, hipStream_t st);
// bs_strip.hip: the bf16 small-width family (K, N <= 192; fused block d = 64, p = 128): -1 = not one of its shapes / storages
int hftt_bs_strip_linear_try(const hftt_strip_desc& d, hipStream_t st);
int hftt_bs_strip_mlp_try(const hftt_ffn_desc& d, hipStream_t st);
