// GPU box probe: the operand maps of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, checked against a CPU product -- groundwork for the
// fp8 cross terms of DESIGN section 8 (1a).  The map, as found with one-hot operands (mfma_f8_discover.hip, mfma_f8_scales.hip): lane l holds, in
// eight registers = 32 bytes, row (A) / column (B) l % 32; byte p of lane l of A meets byte p of lane l' of B when l / 32 == l' / 32 (so any
// k labelling that A and B share multiplies correctly); the SCALE blocks are {bytes 0..15 of lanes r and r + 32} with the scale byte
// (E8M0, 127 = 1) of lane r and {bytes 16..31 of lanes r and r + 32} with the scale byte of lane r + 32 -- i.e. k = 32 (p / 16) + 16 (l / 32)
// + p % 16; the result is in the 32x32 accumulator map of the bf16 instruction (register 4 j + e: row 8 j + 4 (l / 32) + e, column l % 32).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f8_map.hip -o /tmp/mfma_f8_map && /tmp/mfma_f8_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void k_mfma(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* C) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int r = 0; r < 8; r++) { a[r] = reinterpret_cast<const int*>(A + l * 32)[r]; b[r] = reinterpret_cast<const int*>(B + l * 32)[r]; }
  f32x16 c;
  for (int r = 0; r < 16; r++) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; r++) C[l * 16 + r] = c[r];
}

static unsigned char e4m3(float v) {          // exact for the small integers / halves used here
  if (v == 0.f) return 0;
  unsigned char s = v < 0 ? 0x80 : 0; v = fabsf(v);
  int e; float m = frexpf(v, &e);             // v = m * 2^e, m in [0.5, 1)
  m *= 2.f; e -= 1;                           // m in [1, 2)
  int E = e + 7;
  if (E < 1) { int q = (int)lrintf(v * 512.f); return s | (unsigned char)q; }   // subnormal: multiples of 2^-9
  int M = (int)lrintf((m - 1.f) * 8.f);
  return s | (unsigned char)((E << 3) | M);
}

int main() {
  srand(7);
  std::vector<float> Af(32 * 64), Bf(64 * 32);
  for (auto& v : Af) v = (float)(rand() % 9 - 4) * 0.5f;
  for (auto& v : Bf) v = (float)(rand() % 9 - 4);
  // lane images under the hypothesis
  std::vector<unsigned char> Ab(64 * 32), Bb(64 * 32);
  for (int l = 0; l < 64; l++) for (int p = 0; p < 32; p++) { Ab[l * 32 + p] = e4m3(Af[(l % 32) * 64 + 32 * (l / 32) + p]); Bb[l * 32 + p] = e4m3(Bf[(32 * (l / 32) + p) * 32 + (l % 32)]); }
  // scales: A's upper k half (lanes 32..63) x 2, B's rows of odd columns x 4 (E8M0 exponent bytes 128 / 129)
  std::vector<int> sa(64), sb(64);
  for (int l = 0; l < 64; l++) { sa[l] = l >= 32 ? 128 : 127; sb[l] = (l & 1) ? 129 : 127; }
  unsigned char *dA, *dB; int *dsa, *dsb; float* dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 64 * 16 * 4);
  hipMemcpy(dA, Ab.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, Bb.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
  std::vector<float> C(64 * 16);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0; int bad = 0;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
    const int row = 8 * (r / 4) + 4 * (l / 32) + (r % 4), col = l % 32;
    double ref = 0;
    // (this file stores element k of a row in lane half k / 32, byte k % 32: its scale is lane (row + 32 ((k % 32) / 16))'s -- doubled for lanes >= 32)
    for (int k = 0; k < 64; k++) ref += (double)Af[row * 64 + k] * ((k % 32) >= 16 ? 2.0 : 1.0) * (double)Bf[k * 32 + col] * ((col & 1) ? 4.0 : 1.0);
    const double d = fabs(ref - (double)C[l * 16 + r]);
    if (d > worst) worst = d;
    if (d > 1e-3) bad++;
  }
  printf("v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3) against the maps above: %d of 1024 elements differ, worst |diff| %.3g\n", bad, worst);
  return 0;
}
