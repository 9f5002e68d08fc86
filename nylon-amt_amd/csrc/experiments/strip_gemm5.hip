// Strip linear kernel, fifth form ("v5"): TWO 32-token strips per wave, 128 output columns per pass.
//
// Why.  The mover-wave experiment (strip_gemm4.hip) showed what bounds the persistent strip kernels: the bytes one CU's memory path moves,
// and 60 % of them are WEIGHTS -- every 128-token block streams the whole weight matrix L2 -> LDS again, and every wave reads every
// fragment LDS -> registers for a single MFMA.  The schedule is not the lever; fewer weight bytes per token are.
//
// Here a wave owns 64 tokens (two strips, both held as MFMA B operands: 2 x 64 registers) and a pass covers 128 output columns (four
// 32-column tiles: accumulators 2 x 4 x 16 = 128 registers, as before).  A ring slot still holds 16 weight fragments, now four tiles x
// four 16-deep k chunks (two 32-wide k tiles), and every fragment read from LDS feeds TWO MFMAs.  Per token that halves the L2 -> LDS
// stream and the LDS -> register reads; a workgroup walks 256-token blocks.  The weight pack is the one of the other forms: wave w moves
// the four adjacent fragments (u, tiles 4*th .. 4*th+3) of k tile 2q + kt, with kt = w >> 1, u = w & 1 -- a contiguous 4 KB of the packed stream.
//
// Accumulation order per output element is the same as in the second form (k chunks ascending, one accumulator per element), and the
// epilogue arithmetic is the same code: results are BIT-IDENTICAL to strip_gemm2.hip / strip_gemm.hip (tests/test_strip_gpu.py).
//
// Registers: there is no second activation set.  The next step's activations are loaded straight into the strip registers, the pieces of
// k tile 2q once the slot's first eight fragments are done with them, those of k tile 2q+1 behind the slot's last MFMA (ordinary loads:
// the compiler waits for them where they are next used, a whole step later).  Results wait in 16 pending pieces (64 registers) and leave
// four per slot; with a residual those registers then receive the residual rows the next epilogue needs, as in the second form.
// Covered: bf16 tensors, no LayerNorm (a LayerNorm row needs all 256 columns at once), K / 256 in {1, 2, 3}, N / 256 in {1, 2, 3}.
//
// RESULT (MI355X; tools/chk_forms.py V5).  Bit-identical on every shape, with and without residual.  Faster only for K = N = 256 at S_e (73.5 against
// 85.9 us, with residual 101 against 111 us); slower everywhere else: QKV at S_e 191 against 175 us, K = 768 213 against 161 us, and at
// S_n = 90,112 every shape loses 20-45 % (352 blocks of 256 tokens on 256 CUs: the second round is 3/8 full).  Halving the weight
// stream per token bought nothing: a slot now carries twice the HBM operations (8 activation loads, 4 stores, 4 residual loads) and takes
// more than twice as long -- the time of a slot follows the HBM operations it has to retire at its boundary, not its 32 MFMAs and not its
// ring fill.  OPT-IN (HFTT_STRIP_V5=1).  Together with strip_gemm4.hip: the second form is at a local optimum that neither the schedule,
// the queue depth, the role split nor the weight traffic moves; DESIGN.md section 4.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../hftt_common.h"
#include "../hftt_host.h"
#include "../strip_internal.h"
#include "../../../include/hftt_hip.h"

namespace {

#include "../strip_pipe.h"

// 16 fragments -> 32 MFMAs: mfma_f(f, a) multiplies fragment f into both strips; six reads up front, one behind every fragment, the
// slot's share of the memory work (side(f)) in program order between the MFMAs (see slot_mfmas_mix)
template <typename F, typename G>
__device__ __forceinline__ void slot_mfmas2_mix(const unsigned char* slot, F&& mfma_f, G&& side) {
  bf16x8 fr[16];
#pragma unroll
  for (int i = 0; i < 6; i++) fr[i] = *reinterpret_cast<const bf16x8*>(slot + i * 1024);
  static_for<16>([&](auto f_c) __attribute__((always_inline)) {
    constexpr int f = decltype(f_c)::value;
    mfma_f(f_c, fr[f]);
    if (f + 6 < 16) fr[f + 6] = *reinterpret_cast<const bf16x8*>(slot + (f + 6) * 1024);
    side(f_c);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// HP = N / 128 half-passes, KCH = K / 256 chunks, HR = residual rows are added
template <int HP, int KCH, bool HR>
__global__ __launch_bounds__(256, 1) void strip_linear5_kernel(const hftt_strip_desc g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  constexpr int S = HP * KCH * 4;                   // ring slots per 256-token block
  const long nblk = ((long)g.M + 255) / 256;
  float* prm = reinterpret_cast<float*>(smem + RING_BYTES);      // bias[N]
  const unsigned ring = (unsigned)(uintptr_t)HFTT_LDS_PTR(unsigned char, smem);
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(g.x);
  unsigned short* cb = reinterpret_cast<unsigned short*>(g.C);
  const unsigned short* rb = reinterpret_cast<const unsigned short*>(g.residual);
  const bool relu = g.flags & HFTT_SL_RELU;
  const bool plain = !relu && g.out_scale == 1.0f;

  // ring slot `pos` of a block's stream = (half-pass hp, chunk kc, k-tile pair q); this wave's four fragments of it in the packed stream
  const int kt_w = wave >> 1, u_w = wave & 1;
  auto src_of = [&](int pos) {
    const int q = pos & 3, st = pos >> 2;
    const int kc = st % KCH, hp = st / KCH;
    const long sidx = (long)((hp >> 1) * KCH + kc) * 8 + 2 * q + kt_w;
    return g.w + (sidx * 16 + u_w * 8 + 4 * (hp & 1)) * 512 + lane * 8;
  };
  int fill_pos = 0;
  auto advance = [&]() { fill_pos = (fill_pos + 1 == S) ? 0 : fill_pos + 1; };
  auto tok_s = [&](long blk, int s) { const long t = blk * 256 + wave * 64 + s * 32 + j; return t < g.M ? t : (long)g.M - 1; };

  // ---- prologue: bias to LDS, first block's activations, first three ring slots
  for (int i = tid; i < g.N; i += 256) prm[i] = g.bias != nullptr ? g.bias[i] : 0.f;
  u4v xf[2][16], pend[16];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const unsigned short* p0 = xb + tok_s(blockIdx.x, s) * g.ldx + 16 * h;
#pragma unroll
    for (int i = 0; i < 16; i++) pload16(xf[s][i], p0 + piece_off(i));
  }
#pragma unroll
  for (int b = 0; b < FILL_AHEAD; b++) { glds16x4(src_of(fill_pos), ring + (unsigned)b * SLOT_BYTES + (unsigned)wave * 4096u); advance(); }
  static_assert(FILL_AHEAD == 3 && NSLOT == 4, "ring geometry: one step = four slots = one turn of the ring");
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();

  const uint32_t thr = hftt_keep_thr(g.drop_p);
  const float inv_keep = hftt_keep_scale(g.drop_p);
  const unsigned char* abase = smem + lane * 16;
  unsigned short* pend_ptr[2] = {cb, cb};           // where the pending results of each strip go (this lane's view of the 128-column pass)
  bool pend_valid[2] = {false, false};

  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int hb = h;                                       // opaque per iteration (see strip_gemm2.hip)
    asm volatile("" : "+v"(hb));
    const long nxt = blk + gridDim.x;
    const bool has_next = nxt < nblk;
    long tok[2], tokc[2], rrow[2];
    bool wave_ok[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      tok[s] = blk * 256 + wave * 64 + s * 32 + j;
      wave_ok[s] = (blk * 256 + wave * 64 + s * 32) < g.M;        // M % 32 == 0 (host check): a strip is all-valid or all-invalid
      tokc[s] = tok_s(blk, s);
      rrow[s] = g.res_mod > 0 ? (long)((unsigned)tokc[s] % (unsigned)g.res_mod) : tokc[s];
    }
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const float* prm_b = prm + zero;
    for (int hp = 0; hp < HP; hp++) {
      const int col_hp = (hp >> 1) * 256 + (hp & 1) * 128;       // first output column of this half-pass
      f32x16 acc[2][4];
#pragma unroll
      for (int t = 0; t < 4; t++) {                   // accumulators start from the bias
        float b[16];
        lds16f(prm_b + col_hp + t * 32 + 16 * hb, b);
#pragma unroll
        for (int q = 0; q < 16; q++) { acc[0][t][q] = b[q]; acc[1][t][q] = b[q]; }
      }
      for (int kc = 0; kc < KCH; kc++) {
        const bool last_step = (hp == HP - 1) && (kc == KCH - 1);
        const bool refresh = (KCH > 1) ? true : last_step;       // the next step multiplies other activations (past the last block: a harmless re-read)
        const unsigned short* nx_src[2];
        const unsigned short* res_src[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
          nx_src[s] = last_step ? (xb + (has_next ? tok_s(nxt, s) : tokc[s]) * g.ldx + 16 * hb)
                                : (xb + tokc[s] * g.ldx + (kc + 1 == KCH ? 0 : kc + 1) * 256 + 16 * hb);
          res_src[s] = rb + (HR ? rrow[s] * g.ldr + col_hp + 16 * hb : 0);
        }
        const bool pf_res = HR && (kc == KCH - 1);
        static_for<4>([&](auto q_c) __attribute__((always_inline)) {
          constexpr int q = decltype(q_c)::value;
          constexpr int BUF = q;
          constexpr int ps = q >> 1;                  // the pending pieces 4q .. 4q+3 belong to strip ps, tiles 2 (q & 1) and 2 (q & 1) + 1
          HFTT_WAITVM(8);                             // this slot's fragments have landed in every wave (two later four-piece refills were issued since)
          __builtin_amdgcn_s_barrier();
          const unsigned char* slot = abase + BUF * SLOT_BYTES;
          slot_mfmas2_mix(slot,
            [&](auto f_c, bf16x8 a) __attribute__((always_inline)) {
              constexpr int f = decltype(f_c)::value;
              constexpr int kt = f >> 3, u = (f >> 2) & 1, t = f & 3;
              acc[0][t] = mfma32(a, as_frag(xf[0][2 * (2 * q + kt) + u]), acc[0][t]);
              acc[1][t] = mfma32(a, as_frag(xf[1][2 * (2 * q + kt) + u]), acc[1][t]);
            },
            [&](auto f_c) __attribute__((always_inline)) {
              constexpr int f = decltype(f_c)::value;
              if (f == 1) glds16x4(src_of(fill_pos), ring + (unsigned)((BUF + FILL_AHEAD) & (NSLOT - 1)) * SLOT_BYTES + (unsigned)wave * 4096u);
              if (f == 4) {
                if (pend_valid[ps]) {
                  astore16(pend_ptr[ps] + (2 * (q & 1)) * 32, pend[4 * q]);
                  astore16(pend_ptr[ps] + (2 * (q & 1)) * 32 + 8, pend[4 * q + 1]);
                }
              }
              if (f == 6) {
                if (pend_valid[ps]) {
                  astore16(pend_ptr[ps] + (2 * (q & 1) + 1) * 32, pend[4 * q + 2]);
                  astore16(pend_ptr[ps] + (2 * (q & 1) + 1) * 32 + 8, pend[4 * q + 3]);
                }
              }
              if (f == 8) {
                if (HR && pf_res) {
                  pload16(pend[4 * q], res_src[ps] + (2 * (q & 1)) * 32);
                  pload16(pend[4 * q + 1], res_src[ps] + (2 * (q & 1)) * 32 + 8);
                  pload16(pend[4 * q + 2], res_src[ps] + (2 * (q & 1) + 1) * 32);
                  pload16(pend[4 * q + 3], res_src[ps] + (2 * (q & 1) + 1) * 32 + 8);
                }
              }
              if (f == 10) {                          // k tile 2q is finished with (fragments 0-7): its pieces take the next step's values
                if (refresh) {
                  pload16(xf[0][4 * q], nx_src[0] + piece_off(4 * q)); pload16(xf[0][4 * q + 1], nx_src[0] + piece_off(4 * q + 1));
                  pload16(xf[1][4 * q], nx_src[1] + piece_off(4 * q)); pload16(xf[1][4 * q + 1], nx_src[1] + piece_off(4 * q + 1));
                }
              }
              if (f == 12) advance();
            });
          if (refresh) {                              // k tile 2q+1, behind the slot's last MFMA
            pload16(xf[0][4 * q + 2], nx_src[0] + piece_off(4 * q + 2)); pload16(xf[0][4 * q + 3], nx_src[0] + piece_off(4 * q + 3));
            pload16(xf[1][4 * q + 2], nx_src[1] + piece_off(4 * q + 2)); pload16(xf[1][4 * q + 3], nx_src[1] + piece_off(4 * q + 3));
          }
          if (q == 1) pend_valid[0] = false;
          if (q == 3) pend_valid[1] = false;
        });
      }
      // ---------------- epilogue of this half-pass: results into the pending registers ----------------
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const uint64_t rowq = ((uint64_t)tok[s] * (uint64_t)g.N) >> 2;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int col0 = col_hp + t * 32 + 16 * hb;
          float v[16];
          if (plain) {
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = acc[s][t][q];
          } else {
#pragma unroll
            for (int q = 0; q < 16; q++) {
              float x = acc[s][t][q];
              if (relu) x = fmaxf(x, 0.f);
              v[q] = x * g.out_scale;
            }
          }
          if (g.drop_p > 0.f) drop16(v, g.drop_seed, g.drop_site, rowq + (col0 >> 2), thr, inv_keep);
          if (HR) {
            float r[16];
            unpack8(pend[s * 8 + 2 * t], r); unpack8(pend[s * 8 + 2 * t + 1], r + 8);
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] += r[q];
          }
          pend[s * 8 + 2 * t] = pack8u(v);
          pend[s * 8 + 2 * t + 1] = pack8u(v + 8);
          __builtin_amdgcn_sched_barrier(0);
        }
        pend_ptr[s] = cb + tok[s] * g.ldc + col_hp + 16 * hb;
        pend_valid[s] = wave_ok[s];
      }
    }
  }
  // drain: the last half-pass's results
#pragma unroll
  for (int s = 0; s < 2; s++) {
    if (pend_valid[s]) {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        astore16(pend_ptr[s] + t * 32, pend[s * 8 + 2 * t]);
        astore16(pend_ptr[s] + t * 32 + 8, pend[s * 8 + 2 * t + 1]);
      }
    }
  }
  HFTT_WAITVM(0);                                     // nothing may still be on its way into LDS when the workgroup ends (slots fetched past the end)
}

int n_cus5() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n = prop.multiProcessorCount;
  }
  return n;
}

template <int HP, int KCH, bool HR>
int launch_linear5(const hftt_strip_desc& d, hipStream_t st) {
  const int lds = RING_BYTES + 4 * d.N;
  static int attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(strip_linear5_kernel<HP, KCH, HR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { hftt_set_error("strip_linear5: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e)); return 2; }
    attr = lds;
  }
  const int cus = n_cus5();
  if (cus <= 0) { hftt_set_error("strip_linear5: device query failed"); return 2; }
  const long nblk = ((long)d.M + 255) / 256;
  hipLaunchKernelGGL((strip_linear5_kernel<HP, KCH, HR>), dim3((unsigned)(nblk < cus ? nblk : cus)), dim3(256), lds, st, d);
  HFTT_CHECK_LAUNCH("strip_linear5");
  return 0;
}

}  // namespace

// -1: not covered (the caller tries the other forms); otherwise the launch status
int hftt_strip_linear5_try(const hftt_strip_desc& d, hipStream_t st) {
  const char* e = getenv("HFTT_STRIP_V5");          // (read at every call: the tests switch forms inside one process)
  const bool enabled = e && e[0] == '1';
  const uint32_t bf = HFTT_SL_X_BF16 | HFTT_SL_C_BF16;
  if (!enabled || (d.flags & bf) != bf || d.K % 256 != 0 || d.N % 256 != 0 || d.M % 32 != 0 || d.gate != nullptr || d.ln_gamma != nullptr) return -1;
  if (d.residual != nullptr && !(d.flags & HFTT_SL_RES_BF16)) return -1;
  const int hp = d.N / 128, kch = d.K / 256;
  const bool hr = d.residual != nullptr;
#define HFTT_L5(H_, K_) return hr ? launch_linear5<H_, K_, true>(d, st) : launch_linear5<H_, K_, false>(d, st)
  if (kch == 1 && hp == 2) HFTT_L5(2, 1);
  if (kch == 1 && hp == 4) HFTT_L5(4, 1);
  if (kch == 1 && hp == 6) HFTT_L5(6, 1);
  if (kch == 2 && hp == 2) HFTT_L5(2, 2);
  if (kch == 3 && hp == 2) HFTT_L5(2, 3);
#undef HFTT_L5
  return -1;
}
