// Log-mel front end (gfx950): one workgroup per STFT frame.
//   frame (centred, zero padded) * periodic hann -> real 2048-point FFT in LDS (1024-point complex radix-4 Stockham + split) -> |X|^2 (1025 bins)
//   -> sparse slaney/htk mel filterbank (CSR by mel: ~2k non-zeros of 1025x256) -> log(. + offset)
// Replaces torchaudio MelSpectrogram + log of AMT.wav2feature (model/amt.py:59-61); see include/hftt_hip.h.
#include "hftt_common.h"
#include "hftt_host.h"
#include "../../include/hftt_hip.h"

namespace {

// Round 6: the frame is REAL, so its 2048-point transform is one 1024-point COMPLEX transform of z[n] = x[2n] + i x[2n+1] plus a pass that
// separates the spectra of the even and odd samples (X[k] = E[k] + W^k O[k], E / O from Z[k] and conj(Z[N/2 - k])): half the butterflies.
// The 1024-point transform is radix-4 Stockham (autosort: natural order in, natural order out, no bit-reversal pass): 5 stages of ONE
// butterfly per thread and one barrier each (the round-1 kernel ran 11 radix-2 stages of four butterflies per thread on 2048 complex points
// with a zero imaginary half).  Points live in LDS as interleaved (re, im) pairs: a stage READS x[j + q N/4] -- consecutive lanes,
// consecutive 8-byte words: conflict-free -- and writes its four results Ns apart (stride-4 Ns writes: 4-way conflicts in the first three
// stages, none in the last two; 15 of the kernel's ~60 LDS instructions per thread).  Twiddles come from the table the host already passes
// (cos / sin of 2 pi k / 2048, k < 1024; the second half turn is a sign).
template <int NFFT>
__global__ __launch_bounds__(256) void logmel_kernel(const hftt_logmel_desc g) {
  static_assert(NFFT == 2048, "the stage count below is for 1024 complex points");
  constexpr int N = NFFT / 2, Q = N / 4;              // complex points, butterflies per stage (= threads)
  __shared__ float2 bufA[N];
  __shared__ float2 bufB[N];
  __shared__ float pw[N + 1];
  const int tid = threadIdx.x;
  const long frame = blockIdx.x;
  const long start = frame * g.hop - NFFT / 2;
  const float* tc = g.twiddle;
  const float* ts = g.twiddle + NFFT / 2;
  auto tw = [&](int m) {                              // e^(-2 pi i m / 2048), 0 <= m < 2048
    const int mm = m & (NFFT / 2 - 1);
    const float c = tc[mm], sn = ts[mm];
    return (m & (NFFT / 2)) ? make_float2(-c, sn) : make_float2(c, -sn);
  };
  auto cmul = [](float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); };
#pragma unroll
  for (int u = 0; u < N / 256; u++) {
    const int n = tid + 256 * u;
    const long s0 = start + 2 * n;
    const float xe = (s0 >= 0 && s0 < g.n_samples) ? g.wave[s0] * g.window[2 * n] : 0.f;
    const float xo = (s0 + 1 >= 0 && s0 + 1 < g.n_samples) ? g.wave[s0 + 1] * g.window[2 * n + 1] : 0.f;
    bufA[n] = make_float2(xe, xo);
  }
  __syncthreads();
  float2* src = bufA;
  float2* dst = bufB;
#pragma unroll
  for (int Ns = 1; Ns < N; Ns *= 4) {
    const int j = tid, k = j & (Ns - 1);
    const int step = NFFT / (4 * Ns);                 // twiddle index of exp(-2 pi i k / (4 Ns)) per unit k
    float2 v0 = src[j], v1 = src[j + Q], v2 = src[j + 2 * Q], v3 = src[j + 3 * Q];
    if (Ns > 1) { v1 = cmul(v1, tw(k * step)); v2 = cmul(v2, tw(2 * k * step)); v3 = cmul(v3, tw(3 * k * step)); }
    const float2 a02 = make_float2(v0.x + v2.x, v0.y + v2.y), s02 = make_float2(v0.x - v2.x, v0.y - v2.y);
    const float2 a13 = make_float2(v1.x + v3.x, v1.y + v3.y), s13 = make_float2(v1.x - v3.x, v1.y - v3.y);
    const int j0 = ((j - k) << 2) + k;                // (j / Ns) * 4 Ns + k
    dst[j0] = make_float2(a02.x + a13.x, a02.y + a13.y);
    dst[j0 + Ns] = make_float2(s02.x + s13.y, s02.y - s13.x);          // v0 - i v1 - v2 + i v3
    dst[j0 + 2 * Ns] = make_float2(a02.x - a13.x, a02.y - a13.y);
    dst[j0 + 3 * Ns] = make_float2(s02.x - s13.y, s02.y + s13.x);      // v0 + i v1 - v2 - i v3
    __syncthreads();
    float2* t = src; src = dst; dst = t;
  }
  // real spectrum: E = (Z[k] + conj Z[N-k]) / 2, O = (Z[k] - conj Z[N-k]) / (2i), X[k] = E + W^k O, k = 0 .. N (Z[N] = Z[0]); |X|^2
  for (int k = tid; k <= N; k += 256) {
    const float2 zk = src[k & (N - 1)], zn = src[(N - k) & (N - 1)];
    const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
    const float2 wo = cmul(o, tw(k));
    const float xr = e.x + wo.x, xi = e.y + wo.y;
    pw[k] = xr * xr + xi * xi;
  }
  __syncthreads();
  for (int m = tid; m < g.n_mels; m += 256) {
    const int st = g.fb_start[m], len = g.fb_len[m], off = g.fb_off[m];
    float acc = 0.f;
    for (int j = 0; j < len; j++) acc += g.fb_w[off + j] * pw[st + j];
    g.feat[frame * g.n_mels + m] = logf(acc + g.log_offset);
  }
}

// Polyphase band-limited resampling (the convolution of torchaudio.transforms.Resample, model/amt.py:57-58): output sample f * new + p is the
// dot product of taps input samples starting at f * orig - width with kernel row p.  One thread per output sample.  Round 6: a workgroup's 256
// consecutive outputs span at most 255 / up + 1 input frames, so their input window ((f1 - f0) * down + taps samples, zero-filled outside the
// signal) is staged in LDS once -- the tap loop has no bounds test and reads the window as LDS broadcasts -- and the loop runs four
// independent accumulators, sixteen kernel-row loads in flight per lane (the row of a lane is its own 4 x taps bytes of the [up, taps] table,
// L2-resident: <= 300 KB at 44.1 -> 16 kHz; the round-5 loop carried one dependent FMA and one bounds-tested load per tap: 1.3 ms per minute
// of 44.1 kHz audio).
__global__ __launch_bounds__(256) void resample_kernel(const hftt_resample_desc g) {
  extern __shared__ float win[];
  const int tid = threadIdx.x;
  const long o0 = (long)blockIdx.x * 256;
  const long olast = (o0 + 255 < g.n_out ? o0 + 255 : g.n_out - 1);
  const long f0 = o0 / g.up, f1 = olast / g.up;
  const long sbase = f0 * g.down - g.width;
  const int wlen = (int)(f1 - f0) * g.down + g.taps;
  for (int i = tid; i < wlen; i += 256) {
    const long s = sbase + i;
    win[i] = (s >= 0 && s < g.n_in) ? g.wave[s] : 0.f;
  }
  __syncthreads();
  const long o = o0 + tid;
  if (o >= g.n_out) return;
  const long f = o / g.up;
  const int ph = (int)(o - f * g.up);
  const float* k = g.kernel + (long)ph * g.taps;
  const float* w = win + (int)(f - f0) * g.down;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int t = 0;
#pragma unroll 4
  for (; t + 4 <= g.taps; t += 4) {
    a0 = fmaf(w[t], k[t], a0);
    a1 = fmaf(w[t + 1], k[t + 1], a1);
    a2 = fmaf(w[t + 2], k[t + 2], a2);
    a3 = fmaf(w[t + 3], k[t + 3], a3);
  }
  for (; t < g.taps; t++) a0 = fmaf(w[t], k[t], a0);
  g.out[o] = (a0 + a1) + (a2 + a3);
}

}  // namespace

extern "C" int hftt_resample(const hftt_resample_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->wave && d->kernel && d->out, "resample: null operand");
  HFTT_REQUIRE(d->up > 0 && d->down > 0 && d->taps > 0 && d->width >= 0 && d->n_in > 0 && d->n_out > 0, "resample: bad shape");
  HFTT_REQUIRE(d->taps == 2 * d->width + d->down, "resample: taps must be 2 * width + down (the kernel rows of Resample)");
  HFTT_REQUIRE(d->n_out <= (d->n_in * d->up + d->down - 1) / d->down, "resample: n_out exceeds ceil(n_in * up / down)");
  const long wlen = (long)(255 / d->up + 1) * d->down + d->taps;      // the input window of 256 consecutive outputs, staged in LDS
  HFTT_REQUIRE(wlen * 4 <= 64 * 1024, "resample: the input window of one workgroup (%ld samples at down = %d, taps = %d) exceeds 64 KB of LDS", wlen, d->down, d->taps);
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((d->n_out + 255) / 256)), dim3(256), (size_t)wlen * 4, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("resample");
  return 0;
}

extern "C" int hftt_logmel(const hftt_logmel_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->wave && d->window && d->twiddle && d->fb_start && d->fb_len && d->fb_off && d->fb_w && d->feat, "logmel: null operand");
  HFTT_REQUIRE(d->n_fft == 2048, "logmel: n_fft=%d unsupported (2048 only)", d->n_fft);
  HFTT_REQUIRE(d->hop > 0 && d->n_mels > 0 && d->n_frames > 0 && d->n_samples > 0, "logmel: bad shape");
  HFTT_REQUIRE(d->n_frames == 1 + d->n_samples / d->hop, "logmel: n_frames must be 1 + n_samples/hop");
  hipLaunchKernelGGL((logmel_kernel<2048>), dim3((unsigned)d->n_frames), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("logmel");
  return 0;
}
