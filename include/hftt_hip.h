/* libhftt_hip.so -- C ABI of the MI355X (gfx950) hFT-Transformer hot path.
 *
 * The reference (d-f/nylon-amt, vendored Sony hFT-Transformer) has no FFI: its hot path is the Python
 * classes of hftt_code/model/model_spec2midi.py driven by hftt_code/training/train.py and
 * hftt_code/model/amt.py.  This header is the boundary a replacement binds instead of the stock
 * torch.nn ops those classes call; every entry point cites the reference lines whose arithmetic it
 * replaces (paths relative to /root/reference/hftt_code).  Rules of the ABI:
 *   - extern "C", plain device pointers + sizes (+ one POD descriptor per call), no torch types;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no allocation, no sync:
 *     the caller owns all buffers, including workspaces sized by the *_ws_bytes helpers;
 *   - return value 0 = ok, non-zero = error (message via hftt_last_error(), thread-local);
 *   - activations / gradients are fp32 row-major in HBM; in the bf16 mode (npass 1) tensors that are consumed ONLY as MFMA
 *     operands may be stored as bf16 (`io_flags` of each descriptor; leading dimensions and strides then count bf16
 *     elements) -- numerically identical to rounding at load time, half the traffic.  `npass` selects the MFMA arithmetic:
 *       1 = operands rounded to bf16, fp32 accumulate (v_mfma_f32_32x32x16_bf16): the "bf16" throughput mode,
 *       2 = "x3", split fp16: every fp32 operand is carried as fp16 hi + fp16 lo and a product as three MFMA passes
 *           (lo.hi + hi.lo + hi.hi, v_mfma_f32_32x32x16_f16, fp32 accumulate): ~22 significant bits, outputs within 1.3e-4 of the
 *           fp32 reference at paper size (budget 1e-3) at 3/16 of the cost of (3).  All tensors fp32 in HBM.  Forward products.
 *       3 = exact fp32 products and accumulation (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate): the round-1 parity mode,
 *       4 = "x3", split bf16 (bf16 hi + bf16 lo, three passes): ~16 significant bits with fp32's exponent range: products with a
 *           GRADIENT operand (1e-4 .. 1e-10 in magnitude: subnormal or zero as fp16).  On the forward it is not enough for the
 *           reference's first encoder layer, whose attention logits reach ~1e5 on raw log-mel input (velocity logits 1.2e-3 off).
 *     hftt_attn_bwd with npass 2 recomputes the scores in split fp16 (bit for bit the forward's) and forms the four gradient
 *     products in split bf16.
 */
#ifndef HFTT_HIP_H
#define HFTT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HFTT_ABI_VERSION 8

int hftt_abi_version(void);
/* bit 0: the library carries the opt-in gradient-rounding forms (HFTT_SL_X3_GRAD_HI, HFTT_TN_DY_HI, HFTT_NT_A_HI: a gradient operand enters a
 * backward product as its bf16 rounding -- faster, outside the 1e-3 the default keeps for gradients).  They are built only with
 * HFTT_BUILD_GRAD_HI=1 (nylon-amt_amd/build.py); the default library rejects those flags. */
int hftt_build_options(void);
const char* hftt_last_error(void);
/* ONE PROCESS DRIVES ONE DEVICE (one process per GPU, as torch.distributed launches them): the launchers keep per-kernel state -- dynamic-LDS
 * attributes, the CU count, the resident-workgroup grids of the persistent kernels -- in process-wide caches made for the device of the
 * first launch.  A launch from the same process with another device current returns 3 with a message instead of running with the first
 * device's cached state.  Every entry point may be called from any host thread; launches go to the stream that is passed in. */
/* number of compute units of the current device (for workspace sizing on the host side) */
int hftt_device_cus(void);

/* ---------------------------------------------------------------------------------------------
 * Dropout (nn.Dropout of model_spec2midi.py:95,236,242,372 and the attention probabilities :345).  Every kernel that drops takes
 * (drop_p, drop_site, drop_seed) and decides element `idx` of its site (idx = row*N + col of the tensor it produces; attention:
 * ((seq*H + head)*Lq + q)*Lk + k) with the counter-based generator of csrc/hftt_common.h:
 *     key  = splitmix64 round over (drop_seed + 0x9E3779B97F4A7C15 * (drop_site + 1))
 *     word = 32-bit two-multiply mix of (idx >> 2) with key          (one word per FOUR consecutive elements)
 *     keep = byte (idx & 3) of word  <  round((1 - drop_p) * 256)
 * with thr = round((1 - drop_p) * 256); kept elements are scaled by 256 / thr -- the reciprocal of the keep probability ACTUALLY applied
 * (p = 0.1: thr = 230, scale 1.11304, not 1/0.9 = 1.11111), so E[dropout(x)] = x exactly (csrc/hftt_common.h: hftt_keep_scale); drop_p
 * so small that thr = 256 drops nothing and scales by 1.  The backward of a site regenerates the same decisions from the same three numbers
 * (no mask tensor exists); tests/util.py::keep_mask is the numpy restatement the tests compare against.
 * --------------------------------------------------------------------------------------------- */

/* ---------------------------------------------------------------------------------------------
 * Weight preparation.  fp32 parameters -> the layouts the GEMMs consume (concatenated / transposed), as a bf16
 * plane `wbf` (npass 1) and/or an fp32 copy `wf32` (npass 3); either may be NULL.
 * (replaces nothing in the reference: it is the operand format of the MFMA kernels).
 * Each entry copies a [rows, cols] fp32 matrix (row stride src_ld) from `params + src_off` to
 * `dst + dst_off`, either as-is (dst[r*dst_ld + c]) or transposed (dst[c*dst_ld + r]).
 * Entries may also write fp32 vectors (bias concatenation): kind 2 copies `rows*cols` floats to
 * `fdst + dst_off`.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t src_off;   /* element offset into params */
  int64_t dst_off;   /* element offset into the destination plane(s) */
  int32_t rows, cols, src_ld, dst_ld;
  int32_t kind;      /* 0 = copy, 1 = transposed copy (both into wbf/wf32), 2 = fp32 copy into fdst */
  int32_t pad;
} hftt_prep_entry;
int hftt_prep_weights(const float* params, uint16_t* wbf, float* wf32, float* fdst,
                      const hftt_prep_entry* table_dev, int n_entries, void* stream);
/* The same table for the split ("x3") modes: every matrix entry is written as TWO 16-bit planes, whi[dst] + wlo[dst] ~ the fp32 value,
 * with the entry's `pad` field naming the element type (2 = fp16 for the forward matrices, 4 = bf16 for the transposed matrices of
 * the dX products); kind 2 entries copy fp32 vectors into fdst as above. */
int hftt_prep_weights_x3(const float* params, uint16_t* whi, uint16_t* wlo, float* fdst,
                         const hftt_prep_entry* table_dev, int n_entries, void* stream);

/* ---------------------------------------------------------------------------------------------
 * NT GEMM with fused epilogue:  C[M,N] = epi( A[M,K] . W[N,K]^T + bias )
 * Replaces nn.Linear in MultiHeadAttentionLayer.fc_q/k/v/o (model_spec2midi.py:328-330,357),
 * PositionwiseFeedforwardLayer.fc_1/fc_2 (:372,375), Encoder tok_embedding_freq (:85, after the
 * conv fold), the output heads (:172-175, :203-206) and, with transposed weights, every dX = dY.W of
 * loss.backward() (training/train.py:158).
 * epilogue, in this order:  v = acc + bias[col];  if act==1 v = max(v,0);  v *= out_scale;
 *   if add_table: v += add_table[(row % add_mod)*N + col]              (position embedding, :95)
 *   if gate:      v = gate[row*ldg+col] > 0 ? v*gate_scale : 0          (ReLU/dropout backward)
 *   if drop_p>0:  v = keep(seed,site,row*N+col) ? v*256/thr : 0         (nn.Dropout, :95,236,242,372; thr: see Dropout above)
 *   if residual:  v += residual[(row % res_mod)*ldr + col]              (:236,242)
 *   if ln_gamma:  pre_ln_out = v;  v = LayerNorm(v)*gamma+beta (eps 1e-5, over N; needs N in {64,128,256})
 *                 mean/rstd written per row                              (nn.LayerNorm, :225,236)
 *   C[row*ldc+col] = v
 * W is the prepared matrix [N_pad, K] (K % 32 == 0, N_pad % 64 == 0, rows >= N zero): bf16 when npass == 1,
 * fp32 when npass == 3 (passed through the same pointer); npass 2 / 4: the hi plane in W and the lo plane in W_lo
 * (hftt_prep_weights_x3), A / C / residual / gate fp32.
 * --------------------------------------------------------------------------------------------- */
#define HFTT_NT_A_BF16 1u
#define HFTT_NT_C_BF16 2u      /* not with LayerNorm */
#define HFTT_NT_GATE_BF16 4u
#define HFTT_NT_A_HI 16u       /* npass 4: A (a gradient) enters as its bf16 rounding only -- two MFMA passes against the weight pair */
#define HFTT_NT_RES_BF16 8u    /* residual stored as bf16 (A-stationary bf16-mode kernels: N % 256 == 0, M >= 256, K <= 768) */
typedef struct {
  int32_t M, N, K, npass;
  const float* A; int64_t lda;
  const void* W;                                  /* [N_pad, K] bf16 (npass 1) or fp32 (npass 3) */
  uint32_t io_flags;                              /* HFTT_NT_* (npass 1 only): A / C / gate stored as bf16 */
  uint32_t debug;
  const float* bias;                              /* [N] or NULL */
  float* C; int64_t ldc;
  int32_t act; float out_scale;
  const float* add_table; int32_t add_mod;
  const float* gate; int64_t ldg; float gate_scale;
  float drop_p; uint32_t drop_site; uint64_t drop_seed;
  const float* residual; int64_t ldr; int32_t res_mod;
  const float* ln_gamma; const float* ln_beta; float* pre_ln_out; float* ln_mean; float* ln_rstd;
  const void* W_lo;                               /* npass 2 / 4: lo plane of the prepared weights (same layout as W) */
} hftt_gemm_nt_desc;
int hftt_gemm_nt(const hftt_gemm_nt_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * "Strip" kernels (bf16 mode): the MI355X-first form of every nn.Linear on the path whose output width is a multiple of
 * 256 -- MultiHeadAttentionLayer.fc_q/k/v/o (model_spec2midi.py:328-330,357), PositionwiseFeedforwardLayer (:362-378),
 * the post-norm residual blocks of EncoderLayer / DecoderLayer (:236,242,262,268,289,295,301) and their dX in backward.
 *
 * Geometry: a workgroup of 4 waves owns 128 consecutive tokens; a wave owns a 32-token strip and keeps it in registers as the
 * MFMA *B* operand (lane = (token, half): for every 32-feature group it holds 16 consecutive features).  The weights are the
 * *A* operand: pre-packed once per step into 1 KB MFMA fragments in consumption order ("strip pack"), streamed L2 -> LDS by
 * LDS-DMA through a 4 x 16 KB ring shared by the four waves.  v_mfma_f32_32x32x16_bf16 then leaves C^T in the same
 * (token, half) x 16-consecutive-features layout, so bias / ReLU / dropout / gate / residual / LayerNorm are per-lane register
 * work (row statistics = 128 own values + one v_permlane32_swap), the result of one GEMM is directly the B operand of the
 * next (fused FFN: the p-wide hidden never leaves the registers), and every global access is 16 bytes per lane.
 *
 * strip pack: logical weight matrix Wl[N, K] (out x in features, i.e. the nn.Linear weight for a forward GEMM, its transpose
 * for a dX GEMM).  Fragment(tile, pt, u), 64 lanes x 8 bf16: lane (i = lane & 31, hk = lane >> 5) holds
 *   Wl[32*tile + c(i)][32*pt + 16*hk + 8*u + 0..7],   c(i) = 16*((i >> 2) & 1) + (i & 3) + 4*(i >> 3).
 * Slots of 16 fragments (16 KB), order 0 ("linear"): slot = pass*(K/32) + pt, fragment = u*8 + (tile & 7), pass = tile >> 3;
 * order 1 ("tile-major", K == 256): slot = tile, fragment = 2*pt + u.  The stream position of a slot is
 * slot_offset + slot_stride*slot (the fused FFN interleaves fc_1 tiles with fc_2 K-slices: stride 2, offsets 0 / 1).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t src_off;              /* element offset of the fp32 source matrix inside params */
  int64_t dst_off;              /* bf16-element offset of stream slot 0 inside wstrip */
  int32_t rows, cols, src_ld;   /* source [rows, cols], row stride src_ld */
  int32_t transpose;            /* 0: Wl[n0 + r][k0 + c] = src[r][c];  1: Wl[n0 + c][k0 + r] = src[r][c] */
  int32_t n0, k0;               /* position inside the logical matrix (k0 % 8 == 0) */
  int32_t K;                    /* logical K (multiple of 32) */
  int32_t order;                /* 0 linear, 1 tile-major */
  int32_t slot_stride, slot_offset;
} hftt_strip_pack_entry;
int hftt_strip_pack(const float* params, uint16_t* wstrip, const hftt_strip_pack_entry* table_dev, int n_entries, void* stream);
/* The split-operand form of the pack (same table type; elem = 2: fp16 halves, 4: bf16 halves).  Every fragment(tile, chunk) of the bf16
 * pack becomes a (hi, lo) pair of 1 KB fragments, chunk = 2*pt + u naming the 16-deep k step; a 16 KB slot holds the pairs of 8 tiles for
 * one chunk (order 0: slot = (tile >> 3)*(K/16) + chunk, fragment = 2*(tile & 7) + plane) or of 8 chunks for one tile (order 1, K == 256:
 * slot = 2*tile + (chunk >> 3), fragment = 2*(chunk & 7) + plane); slots come in pairs: stream position = slot_offset +
 * slot_stride*(slot >> 1) + (slot & 1) (plain stream: stride 2, offset 0; the fused block interleaves two pairs per hidden tile:
 * stride 4, offsets 0 / 2).  A packed matrix takes 2 * rows * cols 16-bit elements. */
int hftt_x3_strip_pack(const float* params, uint16_t* wstrip, const hftt_strip_pack_entry* table_dev, int n_entries, int elem, void* stream);

#define HFTT_SL_X_BF16 1u       /* x stored as bf16 (else fp32, rounded to bf16 on load) */
#define HFTT_SL_C_BF16 2u       /* C (and pre_ln_out) stored as bf16 */
#define HFTT_SL_RES_BF16 4u     /* residual stored as bf16 */
#define HFTT_SL_RELU 8u
/* split-operand ("x3") forms of the strip kernels: x, C, residual, pre_ln_out (and the fused block's h_out / gate) are fp32, `w` is an
 * hftt_x3_strip_pack stream of fp16 (HFTT_SL_X3_F16: forward products) or bf16 (HFTT_SL_X3_BF16: products with a gradient operand)
 * hi / lo fragment pairs; every product runs in three MFMA passes.  Shapes: K, N multiples of 256 with N/256 x K/256 in
 * {1x1, 2x1, 3x1, 1x2, 1x3} (LayerNorm form: N == 256, K in {256, 512}), M % 32 == 0, no gate; fused block: d == 256, p == 512.
 * Pack order: K == 256 without LayerNorm takes the TILE-MAJOR pack (order 1: the kernel walks one 32-column output tile at a time), every
 * other form the linear pack (order 0). */
#define HFTT_SL_X3_F16 16u
#define HFTT_SL_X3_BF16 32u
/* fused block in a split mode: h_out / gate are bf16 [M, p] instead of fp32.  The hidden feeds fc_2 from registers at full width either way;
 * the STORED copy is read only as the ReLU / dropout gate and as an operand of the weight-gradient products (hftt_gemm_tn, npass 4 with
 * HFTT_TN_X_BF16 / HFTT_TN_DY_BF16), where 8 mantissa bits of one factor leave the gradient's direction untouched. */
#define HFTT_SL_H_BF16 64u
/* with HFTT_SL_X3_BF16: the strip x (hftt_ffn_bwd_dx: dy and the dh it forms) is a gradient and enters as its bf16 rounding only */
#define HFTT_SL_X3_GRAD_HI 256u
/* split modes, LayerNorm forms (hftt_strip_linear with ln_gamma, hftt_ffn_res_ln_fwd): pre_ln_out is bf16 [M, 256] -- it is read only by
 * hftt_ln_bwd (HFTT_LNB_R_BF16), as xhat = (r - mean) * rstd against the fp32 statistics. */
#define HFTT_SL_PRE_BF16 128u
/* Small widths (split modes): K, N <= 192, multiples of 32, with K x N in {64x64, 64x128, 64x192, 128x64, 192x64} (LayerNorm form: 64x64) and
 * the fused block at d == 64, p == 128 -- the reference's default model (training/m_training.py:56-61).  `w` is then an
 * hftt_x3_strip_pack stream of order 2 ("compact": the (hi, lo) pair of (k chunk c, output tile t) at pair index slot_offset + c * NT + t with
 * NT = slot_stride = N / 32, 2 KB per pair; fused block: first matrix at pairs 0 .. 15, second at 16 .. 31): the whole matrix sits in LDS for the
 * launch (csrc/x3s_strip.h).
 * The same shapes WITHOUT a split flag and with all-bf16 storage (HFTT_SL_X_BF16 | HFTT_SL_C_BF16, a bf16 residual, M % 32 == 0) run the bf16
 * small-width family (csrc/bs_strip.hip, round 5: BASELINE config 2 -- the reference's default model in the bf16 mode): same compact pack, made
 * with bf16 halves (hftt_x3_strip_pack element type 4), of which these kernels read the hi fragment; one MFMA pass; every tensor bf16, statistics
 * fp32; results leave as whole 128-byte lines. */
/* HFTT_SL_X3_F16 without LayerNorm, K == 256 (the output-tile-major kernel): C is written as f16-pair planes per 32-column group (see
 * HFTT_ATTN_Q_F16PAIR) -- the q / k / v projections of the attention layers (model_spec2midi.py:328-330).  N = 256, 512, 768 and, since
 * round 6, 1024 / 1536: the cross-attention K / V projections of two / three decoder layers stacked along N (they share their input, the
 * encoder output: model_spec2midi.py:259,296), one launch that reads it once. */
#define HFTT_SL_C_F16PAIR 512u
/* ABI v7, HFTT_SL_X3_BF16 (backward products), N == K == 256 without LayerNorm / residual: the strip x is the gradient of a dropout OUTPUT of
 * width K and is masked with (drop_p, drop_site, drop_seed) while it is loaded (element (m, k) = element m * K + k of the site); no epilogue
 * dropout in this form.  hftt_ffn_bwd_dx does the same with site_o when drop_p > 0 (the gradient of the block's output dropout). */
#define HFTT_SL_X_DROP 2048u
/* C[M,N] = epi(x[M,K] . Wl[N,K]^T + bias): same epilogue order as hftt_gemm_nt (relu, out_scale, gate, dropout, residual,
 * LayerNorm over N == 256).  N % 256 == 0; K % 32 == 0 and (K <= 256 or K % 256 == 0); gate is bf16. */
typedef struct {
  int32_t M, N, K; uint32_t flags;
  const void* x; int64_t ldx;
  const uint16_t* w;                  /* strip pack, order 0 */
  const float* bias;                  /* [N] or NULL */
  void* C; int64_t ldc;
  float out_scale;
  float gate_scale;
  const uint16_t* gate; int64_t ldg;  /* bf16 [M, N] or NULL: v = gate > 0 ? v*gate_scale : 0 */
  float drop_p; uint32_t drop_site; uint64_t drop_seed;
  const void* residual; int64_t ldr; int32_t res_mod; int32_t pad;
  const float* ln_gamma; const float* ln_beta; void* pre_ln_out; float* ln_mean; float* ln_rstd;
} hftt_strip_desc;
int hftt_strip_linear(const hftt_strip_desc* d, void* stream);

/* Fused position-wise feed-forward block (model_spec2midi.py:369-378 + the post-norm of :242/:268/:301), d == 256:
 *   h = dropout_h(relu(x W1^T + b1))      [M, p] stays in registers (optionally also written as bf16 for backward)
 *   y = LayerNorm(x + dropout_o(h W2^T + b2)) * gamma + beta
 * and, with mode 1, the dX half of its backward:
 *   dh = gate(h) * (dy W2) * gate_scale   (written as bf16 for the weight-gradient GEMMs)
 *   dx = dh W1 + residual
 * w = ONE interleaved strip stream: slot 2t = tile t of the first matrix (order 1), slot 2t+1 = K-slice t of the second
 * (order 0).  p % 32 == 0. */
typedef struct {
  int32_t M, d, p; uint32_t flags;    /* HFTT_SL_X_BF16 / C_BF16 / RES_BF16 */
  int32_t mode; int32_t pad;
  const void* x; int64_t ldx;
  const uint16_t* w;
  const float* b1; const float* b2;   /* [p], [d] (mode 0) */
  uint16_t* h_out; int64_t ldh;       /* mode 0: post-dropout hidden (bf16) or NULL; mode 1: dh (bf16, required) */
  const uint16_t* gate; int64_t ldg;  /* mode 1: stored hidden (bf16) */
  float gate_scale;
  float drop_p; uint32_t site_h, site_o; uint64_t drop_seed;
  const void* residual; int64_t ldr;  /* mode 0: NULL = x itself; mode 1: added to dx (or NULL) */
  const float* ln_gamma; const float* ln_beta; void* pre_ln_out; float* ln_mean; float* ln_rstd;   /* mode 0 */
  void* y; int64_t ldy;
} hftt_ffn_desc;
int hftt_ffn_res_ln_fwd(const hftt_ffn_desc* d, void* stream);   /* mode 0 */
int hftt_ffn_bwd_dx(const hftt_ffn_desc* d, void* stream);       /* mode 1 */
/* ABI v8 (split mode, HFTT_SL_X3_F16; d == 256, p == 512): the second half of a layer behind its attention as ONE launch --
 *   x1 = LayerNorm(residual + dropout(ctx Wo^T + bo))          model_spec2midi.py:236-237 / :262-263 / :295-296   (descriptor o, exactly as for hftt_strip_linear)
 *   y  = LayerNorm(x1 + dropout(FFN(x1)))                       model_spec2midi.py:240-242 / :266-268 / :299-301   (descriptor f, exactly as for hftt_ffn_res_ln_fwd)
 * x1 stays in registers between the two halves: o->C may be NULL (inference plan: x1 is not written at all); when it is given it must be f->x.
 * f->w must be o->w + 16 slots (one weight stream of 80 slots per 128-token block: fc_o in order 0, then the FFN's interleaved pair).
 * Results are bit-identical to hftt_strip_linear(o) followed by hftt_ffn_res_ln_fwd(f). */
int hftt_attn_out_ffn_fwd(const hftt_strip_desc* o, const hftt_ffn_desc* f, void* stream);

/* ---------------------------------------------------------------------------------------------
 * TN GEMM (weight gradient):  dW[N,K] = out_scale * dY[M,N]^T . X[M,K],  db[N] = colsum(dY)
 * Replaces the weight/bias gradient of every nn.Linear in loss.backward() (training/train.py:158).
 * Two launches: partial products per M-split into `ws`, then a reduce that writes (beta=0) or
 * accumulates (beta=1) into up to 8 row segments of the destination (fused QKV / packed heads / stacked cross-attention K, V).
 * --------------------------------------------------------------------------------------------- */
#define HFTT_TN_DY_BF16 1u
#define HFTT_TN_X_BF16 2u
#define HFTT_TN_DY_HI 4u       /* npass 4: dY (the gradient) enters as its bf16 rounding only; X keeps its bf16 pair (two MFMA passes) */
/* ABI v7, npass 4, fp32 dY with N == lddy == the width of the dropped tensor: dY is the gradient of a dropout OUTPUT and the kernel applies
 * the mask of (drop_p, drop_site, drop_seed) while it loads it -- element (m, n) is element m * N + n of the site -- so that the LayerNorm
 * backward does not have to write a masked copy of its result for this product (hftt_ln_bwd with dr_drop == NULL); db = colsum of the
 * masked rows.  Tiles: N >= 256 and K >= 256; the loader indexes the site's hash quads in 32 bits: M < 2^24 and M * N < 2^34 (status 1 beyond). */
#define HFTT_TN_DY_DROP 8u
typedef struct {
  int32_t M, N, K, npass;
  const float* dY; int64_t lddy;
  const float* X; int64_t ldx;
  float out_scale; float beta;
  int32_t n_seg;             /* 1 .. 8 (ABI v7; 4 before: the cross-attention K / V gradients of three decoder layers are six segments of one product) */
  int32_t seg_row0[8]; int32_t seg_rows[8];
  float* seg_dw[8];          /* [seg_rows, K] row-major (ld = K_out) */
  float* seg_db[8];          /* [seg_rows] or NULL */
  int32_t K_out;             /* number of K columns to write (<= K), destination leading dim */
  uint32_t io_flags;         /* HFTT_TN_*: npass 1 (either / both), npass 4 (ONE of them: that operand is its own hi half, two MFMA passes) */
  void* ws; int64_t ws_bytes;
  float drop_p; uint32_t drop_site; uint64_t drop_seed;      /* HFTT_TN_DY_DROP (ABI v7) */
} hftt_gemm_tn_desc;          /* npass 4 (split bf16): dY and X fp32, split on their way into LDS (or one of them stored as bf16) */
int64_t hftt_gemm_tn_ws_bytes(int32_t M, int32_t N, int32_t K);
int hftt_gemm_tn(const hftt_gemm_tn_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-head attention, one workgroup per (sequence, head); Lq, Lk <= 256, dh in {32, 64}.
 * forward:  P = softmax(Q K^T / sqrt(dh));  out = dropout(P) V;  lse = per-row (max score, 1/sum exp) pairs;
 *           probs (optional) = P (pre-dropout)         -- model_spec2midi.py:335-354, returned P :360
 * backward: recomputes P from (Q,K,lse); dQ,dK,dV      -- autograd of the same lines
 * Q/K/V/out element (seq, row, head, c) lives at  base + seq*seq_stride + row*ld + head*dh + c.
 * --------------------------------------------------------------------------------------------- */
#define HFTT_ATTN_Q_BF16 1u       /* q */
#define HFTT_ATTN_KV_BF16 2u      /* k, v */
#define HFTT_ATTN_O_BF16 4u       /* out, dout */
#define HFTT_ATTN_DQ_BF16 8u      /* dq */
#define HFTT_ATTN_DKV_BF16 16u    /* dk, dv */
/* npass 2 only, dh == 64: q / k, v are "f16-pair planes" -- the 32 fp32 slots (128 bytes) of every aligned 32-column group of a row hold
 * the group's 32 fp16 hi halves followed by its 32 fp16 lo halves (x = hi + lo to 2^-22), written ONCE by the projection that produced them
 * (hftt_strip_linear with HFTT_SL_C_F16PAIR, hftt_x3_to_planes) instead of being split by every kernel that reads them; pointers and
 * strides keep their fp32 meaning.  Both flags or neither.  The forward then stages K / V by LDS-DMA (csrc/x3_attn_pl.hip: persistent
 * workgroups, ceil(Lq / 32) <= 8 at Lk > 128 and <= 4 below).  The FORWARD's results (output, row statistics, attention map) are bit-identical
 * to the fp32-operand form's.  The backward recomputes P bitwise as the forward did, but forms the bf16 (hi, lo) operands of its gradient
 * products from the stored fp16 pair (hi + lo, exact in fp32) instead of from the fp32 value: dq / dk / dv agree with the fp32-operand
 * backward to 4e-5 of each tensor's maximum (tests/test_x3_gpu.py::test_attention_on_planes_equals_attention_on_fp32_operands, 12 geometries
 * x dropout on / off). */
#define HFTT_ATTN_Q_F16PAIR 32u
#define HFTT_ATTN_KV_F16PAIR 64u
typedef struct {
  int32_t n_seq, n_heads, Lq, Lk, dh, npass;
  const float* q; int64_t q_seq_stride; int64_t ldq;
  const float* k; int64_t k_seq_stride; int64_t ldk;
  const float* v; int64_t v_seq_stride; int64_t ldv;
  float* out; int64_t o_seq_stride; int64_t ldo;
  float* lse;                 /* [n_seq, n_heads, Lq, 2]: (row max, 1 / sum(exp(score - max))); the max is of the scaled scores, except
                                 in the npass == 2 kernels and the npass == 1 kernels with q, k, v AND out stored as bf16, which keep the
                                 max of the RAW Q.K^T (before the 1/sqrt(dh)) -- an opaque pair between a forward and the backward of
                                 the SAME npass, storage flags AND strides (the all-bf16 form also needs every row / sequence stride of
                                 q, k, v, out to be a multiple of 8 elements; forward and backward test that with one shared predicate) */
  float* probs;               /* [n_seq, n_heads, Lq, Lk] or NULL */
  float drop_p; uint32_t drop_site; uint64_t drop_seed;
  /* backward only */
  const float* dout;          /* same layout as out */
  float* dq; int64_t dq_seq_stride; int64_t lddq;
  float* dk; int64_t dk_seq_stride; int64_t lddk;
  float* dv; int64_t dv_seq_stride; int64_t lddv;
  uint32_t io_flags;          /* HFTT_ATTN_* (the BF16 flags: npass 1 only; the F16PAIR flags: npass 2 only) */
  uint32_t pad;
} hftt_attn_desc;
int hftt_attn_fwd(const hftt_attn_desc* d, void* stream);
int hftt_attn_bwd(const hftt_attn_desc* d, void* stream);
/* fp32 [rows, cols] (row stride lds) -> f16-pair planes [rows, cols] (row stride ldd, both in fp32 slots), per 32-column group; cols % 32 == 0,
 * dst != src.  For operands no strip kernel produces (the shared query of DecoderLayer_Zero, model_spec2midi.py:154-155; tests). */
int hftt_x3_to_planes(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t rows, int32_t cols, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder front: fold of Conv2d(1,C,(1,k)) + window flatten + Linear(C*(n_proc-k+1), d)
 * (model_spec2midi.py:65-85) into one Linear over the n_proc-wide window:
 *   tok[j] = sum_u Weff[j,u] * spec[f, t+u] + beff[j],
 *   Weff[j,u] = sum_{c, kk, w+kk=u} Wtok[j, c*nw + w] * wconv[c,kk],  beff[j] = btok[j] + sum_c bconv[c] sum_w Wtok[j,c*nw+w]
 * fold_fwd writes Weff [d_pad, Kp] (Kp = n_proc rounded up to 32, zero padded) as bf16 and/or fp32, + beff.
 * fold_bwd maps (dWeff [d, Kp] fp32, dbeff [d]) back to the four reference parameters' gradients.
 * im2win materialises A[(b,t,f), u] = spec[b, f, t+u] (zero for u >= n_proc), row stride Kp.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t d, C, kw, n_proc, Kp, d_pad;
  const float* wconv;   /* [C, kw] */
  const float* bconv;   /* [C] */
  const float* wtok;    /* [d, C*(n_proc-kw+1)] */
  const float* btok;    /* [d] */
  uint16_t* weff_bf; float* weff_f32;     /* [d_pad, Kp]; either may be NULL */
  float* beff;          /* [d] */
  /* backward */
  const float* dweff;   /* [d, Kp] */
  const float* dbeff;   /* [d] */
  float* g_wconv; float* g_bconv; float* g_wtok; float* g_btok;
  uint16_t* weff_hi; uint16_t* weff_lo;   /* split fp16 planes of Weff for the x3 mode (both or neither) */
} hftt_fold_desc;
int hftt_embed_fold_fwd(const hftt_fold_desc* d, void* stream);
int hftt_embed_fold_bwd(const hftt_fold_desc* d, void* stream);
int hftt_im2win(const float* spec, float* win, int32_t B, int32_t F, int32_t T, int32_t n_proc, int32_t Kp, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm backward (shared gamma/beta per layer, model_spec2midi.py:225,250,277):
 *   dr = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dy*gamma,  xhat = (r-mean)*rstd
 *   dr_drop (optional) = dropout-masked dr for the residual branch (site/seed of the forward dropout)
 *   partial dgamma/dbeta per workgroup -> ws; hftt_ln_bwd_reduce sums them into the parameter grads.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t M, N;
  const float* dy; const float* r; const float* mean; const float* rstd; const float* gamma;
  float* dr; float* dr_drop;
  float drop_p; uint32_t drop_site; uint64_t drop_seed;
  float* ws;            /* [n_wg, 2, N] partial sums; n_wg = hftt_ln_bwd_wgs(M) */
  uint32_t drop_bf16;   /* dr_drop is stored as bf16 */
  uint32_t io_flags;    /* HFTT_LNB_DY_BF16: dy stored as bf16; HFTT_LNB_DR_BF16: dr stored as bf16 (bf16 gradient stream);
                           HFTT_LNB_R_BF16: the saved pre-LayerNorm sum r stored as bf16 (strip forward kernels) */
} hftt_ln_bwd_desc;
#define HFTT_LNB_DY_BF16 1u
#define HFTT_LNB_DR_BF16 2u
#define HFTT_LNB_R_BF16 4u
int32_t hftt_ln_bwd_wgs(int32_t M);
int hftt_ln_bwd(const hftt_ln_bwd_desc* d, void* stream);
int hftt_ln_bwd_reduce(const float* ws, int32_t n_wg, int32_t N, float* dgamma, float* dbeta, float beta, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Small data-movement kernels of the decoder (model_spec2midi.py:189-191, :172-175, :203-206).
 * --------------------------------------------------------------------------------------------- */
/* y[(b,n), t, :] = dropout( x[(b,t), n, :]*scale + pos[t, :] );  io_flags: HFTT_TE_X_BF16 (x), HFTT_TE_Y_BF16 (y) stored as bf16 */
#define HFTT_TE_X_BF16 1u
#define HFTT_TE_Y_BF16 2u
#define HFTT_TE_M_BF16 4u
int hftt_time_embed_fwd(const float* x, const float* pos, float* y, int32_t B, int32_t T, int32_t Nn, int32_t d,
                        float scale, float drop_p, uint32_t site, uint64_t seed, uint32_t io_flags, void* stream);
/* dx[(b,t), n, :] (+)= mask*dy[(b,n), t, :]*scale ; dym (optional, [(b,n),t,:]) = masked dy (for the pos-emb colsum);
 * io_flags: HFTT_TE_X_BF16 (dy), HFTT_TE_Y_BF16 (dx), HFTT_TE_M_BF16 (dym) stored as bf16 */
int hftt_time_embed_bwd(const float* dy, float* dx, float* dym, int32_t B, int32_t T, int32_t Nn, int32_t d,
                        float scale, float drop_p, uint32_t site, uint64_t seed, int32_t accumulate, uint32_t io_flags, void* stream);
/* in-place dropout backward: g *= mask*256/thr, thr = round((1-p)*256) (same indexing as the forward epilogue: idx = row*N+col); n % 4 == 0; bf16 != 0: g is bf16 */
int hftt_dropout_bwd(float* g, int64_t n, float drop_p, uint32_t site, uint64_t seed, uint32_t bf16, void* stream);
/* colsum: out[j] = beta*out[j] + sum_r x[r*ld + j], j < n (x fp32, or bf16 when x_bf16 != 0; out fp32) */
int hftt_colsum(const float* x, int64_t rows, int64_t n, int64_t ld, float* out, float beta, float* ws, uint32_t x_bf16, void* stream);
int64_t hftt_colsum_ws_bytes(int64_t rows, int64_t n);
/* heads: logits [S, ldl] with cols [0,V) velocity, V onset, V+1 offset, V+2 mpe ->
 *   velocity [.,V] raw, onset/offset/mpe sigmoid; time_major!=0: rows are (b,n,t) and outputs are (b,t,n). */
int hftt_heads_split(const float* logits, int64_t ldl, float* onset, float* offset, float* mpe, float* velocity,
                     int32_t B, int32_t T, int32_t Nn, int32_t V, int32_t time_major, void* stream);
/* backward of hftt_heads_split: dlogits[., ldl] from d(onset,offset,mpe probabilities) and d(velocity logits) */
int hftt_heads_split_bwd(const float* p_onset, const float* p_offset, const float* p_mpe,
                         const float* d_onset, const float* d_offset, const float* d_mpe, const float* d_velocity,
                         float* dlogits, int64_t ldl, int32_t B, int32_t T, int32_t Nn, int32_t V, int32_t time_major,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss (training/train.py:141-153): 6x BCELoss(mean) + 2x CrossEntropyLoss(mean), weighted sum.
 * loss_out[0] = total, [1..8] = the eight terms.  Gradients w.r.t. the eight model outputs are
 * written when the d_* pointers are non-NULL (scaled by grad_scale, normally 1).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t n;            /* B*T*Nn label elements */
  int32_t V; int32_t pad;
  const float* prob[6]; /* onset_A, offset_A, mpe_A, onset_B, offset_B, mpe_B : [n] */
  const float* vel[2];  /* velocity_A, velocity_B : [n, V] logits */
  const float* label_onset; const float* label_offset; const float* label_mpe; const int64_t* label_velocity;
  float weight_A, weight_B, grad_scale; float pad2;
  float* d_prob[6]; float* d_vel[2];
  float* loss_out;      /* [9] */
  float* ws;            /* hftt_loss_ws_bytes(n) */
} hftt_loss_desc;
int64_t hftt_loss_ws_bytes(int64_t n);
int hftt_loss(const hftt_loss_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused Adam over one flat parameter buffer (torch.optim.Adam defaults, m_training.py:146):
 *   m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * Hyper-parameters are doubles (python floats): 1-b1, 1-b2 and the bias corrections are formed in double and rounded once, like torch.
 * --------------------------------------------------------------------------------------------- */
int hftt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int32_t step,
                   double lr, double beta1, double beta2, double eps, double grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Log-mel front end (model/amt.py:55-63 after resampling): frames of n_fft with hop, centred,
 * zero padded, periodic hann, |rDFT|^2, sparse slaney/htk mel filterbank, log(. + offset).
 * feat [n_frames, n_mels], n_frames = 1 + n_samples/hop.  fb is given in CSR-by-mel form.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  const float* wave; int64_t n_samples;
  int32_t n_fft, hop, n_mels, n_frames;
  const float* window;       /* [n_fft] */
  const float* twiddle;      /* [n_fft/2] cos, then [n_fft/2] sin of 2*pi*k/n_fft */
  const int32_t* fb_start;   /* [n_mels] first frequency bin of each mel filter */
  const int32_t* fb_len;     /* [n_mels] number of bins */
  const int32_t* fb_off;     /* [n_mels] offset into fb_w */
  const float* fb_w;         /* packed weights */
  float log_offset; int32_t pad;
  float* feat;
} hftt_logmel_desc;
int hftt_logmel(const hftt_logmel_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Resampling in front of the log-mel (model/amt.py:57-58: torchaudio.transforms.Resample(sr, 16000) with its defaults -- Hann-windowed sinc,
 * lowpass_filter_width 6, rolloff 0.99; torchaudio is un-vendored and absent: published algorithm, parity unpinned).  With g = gcd(sr_in,
 * sr_out), down = sr_in / g, up = sr_out / g, width = ceil(6 * down / (min(down, up) * 0.99)): the host builds the kernel table
 * kernel[up, taps], taps = 2 * width + down (row p = the windowed sinc sampled for output phase p), and
 *     out[f * up + p] = sum_t wave[f * down - width + t] * kernel[p, t]          (samples outside [0, n_in) are zeros)
 * for the first n_out = ceil(n_in * up / down) outputs.  fp32 accumulate, four interleaved partial sums over the taps (t mod 4), added at the end.
 * The input window of 256 consecutive outputs, (255 / up + 1) * down + taps samples, is staged in LDS: it must fit 64 KB (it does for every
 * audio rate down to 16 kHz from <= 768 kHz).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  const float* wave; int64_t n_in;
  const float* kernel;       /* [up, taps] */
  int32_t up, down, width, taps;
  float* out; int64_t n_out;
} hftt_resample_desc;
int hftt_resample(const hftt_resample_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HFTT_HIP_H */
