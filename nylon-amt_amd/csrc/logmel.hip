// Log-mel front end (gfx950): one workgroup per STFT frame.
//   frame (centred, zero padded) * periodic hann -> 2048-point radix-2 FFT in LDS -> |X|^2 (1025 bins)
//   -> sparse slaney/htk mel filterbank (CSR by mel: ~2k non-zeros of 1025x256) -> log(. + offset)
// Replaces torchaudio MelSpectrogram + log of AMT.wav2feature (model/amt.py:59-61); see include/hftt_hip.h.
#include "hftt_common.h"
#include "hftt_host.h"
#include "../../include/hftt_hip.h"

namespace {

template <int NFFT, int LOG2N>
__global__ __launch_bounds__(256) void logmel_kernel(const hftt_logmel_desc g) {
  __shared__ float re[NFFT];
  __shared__ float im[NFFT];
  __shared__ float pw[NFFT / 2 + 1];
  const int tid = threadIdx.x;
  const long frame = blockIdx.x;
  const long start = frame * g.hop - NFFT / 2;
  for (int i = tid; i < NFFT; i += 256) {
    const long s = start + i;
    const float x = (s >= 0 && s < g.n_samples) ? g.wave[s] * g.window[i] : 0.f;
    const int j = (int)(__brev((unsigned)i) >> (32 - LOG2N));
    re[j] = x;
    im[j] = 0.f;
  }
  __syncthreads();
  const float* tc = g.twiddle;
  const float* ts = g.twiddle + NFFT / 2;
#pragma unroll 1
  for (int s = 1; s <= LOG2N; s++) {
    const int half = 1 << (s - 1);
    const int tstep = NFFT >> s;
    for (int b = tid; b < NFFT / 2; b += 256) {
      const int pos = b & (half - 1);
      const int i0 = ((b >> (s - 1)) << s) + pos;
      const int i1 = i0 + half;
      const float wr = tc[pos * tstep], wi = -ts[pos * tstep];
      const float xr = re[i1], xi = im[i1];
      const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
      const float ar = re[i0], ai = im[i0];
      re[i0] = ar + tr; im[i0] = ai + ti;
      re[i1] = ar - tr; im[i1] = ai - ti;
    }
    __syncthreads();
  }
  for (int k = tid; k <= NFFT / 2; k += 256) pw[k] = re[k] * re[k] + im[k] * im[k];
  __syncthreads();
  for (int m = tid; m < g.n_mels; m += 256) {
    const int st = g.fb_start[m], len = g.fb_len[m], off = g.fb_off[m];
    float acc = 0.f;
    for (int j = 0; j < len; j++) acc += g.fb_w[off + j] * pw[st + j];
    g.feat[frame * g.n_mels + m] = logf(acc + g.log_offset);
  }
}

// Polyphase band-limited resampling (the convolution of torchaudio.transforms.Resample, model/amt.py:57-58): output sample f * new + p is the
// dot product of taps input samples starting at f * orig - width with kernel row p.  One thread per output sample; a workgroup's 256
// consecutive outputs read overlapping input windows (L1 / L2 hits) and the kernel table ([new, taps] floats, <= 300 KB at 44.1 -> 16 kHz)
// stays in L2.  0.5 GFLOP per minute of audio: latency-bound, not tuned.
__global__ __launch_bounds__(256) void resample_kernel(const hftt_resample_desc g) {
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= g.n_out) return;
  const long f = o / g.up;
  const int ph = (int)(o - f * g.up);
  const long s0 = f * g.down - g.width;
  const float* k = g.kernel + (long)ph * g.taps;
  float acc = 0.f;
  for (int t = 0; t < g.taps; t++) {
    const long s = s0 + t;
    const float x = (s >= 0 && s < g.n_in) ? g.wave[s] : 0.f;
    acc = fmaf(x, k[t], acc);
  }
  g.out[o] = acc;
}

}  // namespace

extern "C" int hftt_resample(const hftt_resample_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->wave && d->kernel && d->out, "resample: null operand");
  HFTT_REQUIRE(d->up > 0 && d->down > 0 && d->taps > 0 && d->width >= 0 && d->n_in > 0 && d->n_out > 0, "resample: bad shape");
  HFTT_REQUIRE(d->taps == 2 * d->width + d->down, "resample: taps must be 2 * width + down (the kernel rows of Resample)");
  HFTT_REQUIRE(d->n_out <= (d->n_in * d->up + d->down - 1) / d->down, "resample: n_out exceeds ceil(n_in * up / down)");
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((d->n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("resample");
  return 0;
}

extern "C" int hftt_logmel(const hftt_logmel_desc* d, void* stream) {
  HFTT_REQUIRE(d && d->wave && d->window && d->twiddle && d->fb_start && d->fb_len && d->fb_off && d->fb_w && d->feat, "logmel: null operand");
  HFTT_REQUIRE(d->n_fft == 2048, "logmel: n_fft=%d unsupported (2048 only)", d->n_fft);
  HFTT_REQUIRE(d->hop > 0 && d->n_mels > 0 && d->n_frames > 0 && d->n_samples > 0, "logmel: bad shape");
  HFTT_REQUIRE(d->n_frames == 1 + d->n_samples / d->hop, "logmel: n_frames must be 1 + n_samples/hop");
  hipLaunchKernelGGL((logmel_kernel<2048, 11>), dim3((unsigned)d->n_frames), dim3(256), 0, (hipStream_t)stream, *d);
  HFTT_CHECK_LAUNCH("logmel");
  return 0;
}
