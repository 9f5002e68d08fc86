// Probe of gfx950 MFMA fragment layouts and ds_read_tr16_b64 semantics.
// Test infrastructure only (not part of the product path): prints PASS/FAIL per
// assumption plus raw dumps, so kernels in nylon-amt_amd/csrc can rely on
// verified lane maps.  Build: hipcc --offload-arch=gfx950 -O2 probe_mfma.hip -o probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

__device__ __host__ inline unsigned short f2bf(float f) {
  unsigned int u; memcpy(&u, &f, 4);
  u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
  return (unsigned short)u;
}
__device__ __host__ inline float bf2f(unsigned short h) {
  unsigned int u = ((unsigned int)h) << 16; float f; memcpy(&f, &u, 4); return f;
}

// ---- test 1: 32x32x16.  A[32][16] row-major floats, B[16][32] row-major. C[32][32].
__global__ void k_mfma32(const float* A, const float* B, float* C) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (short)f2bf(A[r * 16 + 8 * h + j]); b[j] = (short)f2bf(B[(8 * h + j) * 32 + r]); }
  f32x16 acc; for (int i = 0; i < 16; i++) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int g = 0; g < 16; g++) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; C[row * 32 + r] = acc[g]; }
}
// ---- test 2: 16x16x32. A[16][32], B[32][16], C[16][16]
__global__ void k_mfma16(const float* A, const float* B, float* C) {
  int l = threadIdx.x; int r = l & 15, q = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (short)f2bf(A[r * 32 + 8 * q + j]); b[j] = (short)f2bf(B[(8 * q + j) * 16 + r]); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int g = 0; g < 4; g++) { int row = q * 4 + g; C[row * 16 + r] = acc[g]; }
}
// ---- test 3: tr16 read raw dump. LDS matrix M[32 rows][64 cols] of shorts, value = row*64+col.
// lane 4q+p of each 16-lane group gi supplies address of row (4*gi + q), cols 4p..4p+3 (+16*0).
__global__ void k_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short M[32 * 64];
  for (int i = threadIdx.x; i < 32 * 64; i += 64) M[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x; int gi = l >> 4, li = l & 15; int q = li >> 2, p = li & 3;
  const short* addr = &M[(4 * gi + q) * 64 + 4 * p];
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)addr);
  for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}
// ---- test 4: acc tile as next operand.  X = A1[32][16] * B1[16][32] (32x32). Z = X^T * B2 where B2[32][32]
//      and Y = A2 * X where A2[32][32].  uses two k-steps with permuted k order.
__global__ void k_chain(const float* A1, const float* B1, const float* A2, const float* B2, float* Z, float* Y) {
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (short)f2bf(A1[r * 16 + 8 * h + j]); b[j] = (short)f2bf(B1[(8 * h + j) * 32 + r]); }
  f32x16 x; for (int i = 0; i < 16; i++) x[i] = 0.f;
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 z, y; for (int i = 0; i < 16; i++) { z[i] = 0.f; y[i] = 0.f; }
  for (int s = 0; s < 2; s++) {
    bf16x8 xf, b2, a2;
    for (int j = 0; j < 8; j++) {
      xf[j] = (short)f2bf(x[8 * s + j]);
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);   // row of X this element represents
      b2[j] = (short)f2bf(B2[k * 32 + r]);               // B2[k][col r]
      a2[j] = (short)f2bf(A2[r * 32 + k]);               // A2[row r][k]
    }
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, b2, z, 0, 0, 0);   // Z = X^T * B2
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xf, y, 0, 0, 0);   // Y = A2 * X
  }
  for (int g = 0; g < 16; g++) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; Z[row * 32 + r] = z[g]; Y[row * 32 + r] = y[g]; }
}
// ---- test 5: tr read feeding an MFMA A operand: Mem is T[m][n] (16 m-rows x 32 n-cols) bf16 in LDS;
//      want A[row=n][k=m] for 32x32x16 (A = T^T, 32x16).  C = A * B, B[16][32].
__global__ void k_tr_mfma(const float* T, const float* B, float* C) {
  __shared__ __attribute__((aligned(16))) short Ts[16 * 32];
  for (int i = threadIdx.x; i < 16 * 32; i += 64) Ts[i] = (short)f2bf(T[i]);
  __syncthreads();
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  // lane l needs A[row r][k=8h+j] = T[m=8h+j][n=r], j=0..7 -> two tr reads (j 0..3, 4..7).
  // 16-lane group gi = l>>4 covers rows n = 16*(gi&1) + (0..15), k-half h = gi>>1.
  int li = l & 15, gi = l >> 4; int q = li >> 2, p = li & 3; int nb = 16 * (gi & 1);
  bf16x8 a, b;
  for (int half = 0; half < 2; half++) {
    int m = 8 * h + 4 * half + q;               // block row q of this 4-row block
    const short* addr = &Ts[m * 32 + nb + 4 * p];
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)addr);
    for (int j = 0; j < 4; j++) a[4 * half + j] = v[j];
  }
  for (int j = 0; j < 8; j++) b[j] = (short)f2bf(B[(8 * h + j) * 32 + r]);
  f32x16 acc; for (int i = 0; i < 16; i++) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int g = 0; g < 16; g++) { int row = (g & 3) + 8 * (g >> 2) + 4 * h; C[row * 32 + r] = acc[g]; }
}

static float ival(int i, int j, int salt) { return (float)(((i * 7 + j * 13 + salt * 5) % 9) - 4); }

int main() {
  int dev = 0; CK(hipSetDevice(dev)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  printf("device %s arch %s CUs %d\n", p.name, p.gcnArchName, p.multiProcessorCount);
  float *dA, *dB, *dC, *dA2, *dB2, *dZ, *dY; short* dS;
  CK(hipMalloc(&dA, 4096 * 4)); CK(hipMalloc(&dB, 4096 * 4)); CK(hipMalloc(&dC, 4096 * 4));
  CK(hipMalloc(&dA2, 4096 * 4)); CK(hipMalloc(&dB2, 4096 * 4)); CK(hipMalloc(&dZ, 4096 * 4)); CK(hipMalloc(&dY, 4096 * 4));
  CK(hipMalloc(&dS, 4096 * 2));
  std::vector<float> A(4096), B(4096), C(4096), A2(4096), B2(4096), Z(4096), Y(4096);
  // test 1
  for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) A[i * 16 + k] = ival(i, k, 1);
  for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) B[k * 32 + j] = ival(k, j, 2);
  CK(hipMemcpy(dA, A.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 4096 * 4, hipMemcpyHostToDevice));
  k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 4096 * 4, hipMemcpyDeviceToHost));
  { double e = 0; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { double s = 0; for (int k = 0; k < 16; k++) s += A[i * 16 + k] * B[k * 32 + j]; e = fmax(e, fabs(s - C[i * 32 + j])); }
    printf("T1 mfma32x32x16 layout: max err %g -> %s\n", e, e == 0 ? "PASS" : "FAIL"); }
  // test 2
  for (int i = 0; i < 16; i++) for (int k = 0; k < 32; k++) A[i * 32 + k] = ival(i, k, 3);
  for (int k = 0; k < 32; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = ival(k, j, 4);
  CK(hipMemcpy(dA, A.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 4096 * 4, hipMemcpyHostToDevice));
  k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 4096 * 4, hipMemcpyDeviceToHost));
  { double e = 0; for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 32; k++) s += A[i * 32 + k] * B[k * 16 + j]; e = fmax(e, fabs(s - C[i * 16 + j])); }
    printf("T2 mfma16x16x32 layout: max err %g -> %s\n", e, e == 0 ? "PASS" : "FAIL"); }
  // test 3
  { std::vector<short> S(256); k_tr<<<1, 64>>>(dS); CK(hipDeviceSynchronize()); CK(hipMemcpy(S.data(), dS, 256 * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++) { int gi = l >> 4, li = l & 15; for (int j = 0; j < 4; j++) { int expect = (4 * gi + j) * 64 + li; if (S[l * 4 + j] != expect) bad++; } }
    printf("T3 ds_read_tr16_b64: %d mismatches -> %s\n", bad, bad == 0 ? "PASS" : "FAIL");
    printf("T3 raw (lane: 4 values as row.col):\n");
    for (int l = 0; l < 64; l++) { printf(" l%02d:", l); for (int j = 0; j < 4; j++) printf(" %d.%d", S[l * 4 + j] / 64, S[l * 4 + j] % 64); if ((l & 3) == 3) printf("\n"); } }
  // test 4
  for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) A[i * 16 + k] = ival(i, k, 5) * 0.5f;
  for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) B[k * 32 + j] = ival(k, j, 6) * 0.5f;
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { A2[i * 32 + j] = ival(i, j, 7); B2[i * 32 + j] = ival(i, j, 8); }
  CK(hipMemcpy(dA, A.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dA2, A2.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB2, B2.data(), 4096 * 4, hipMemcpyHostToDevice));
  k_chain<<<1, 64>>>(dA, dB, dA2, dB2, dZ, dY); CK(hipDeviceSynchronize());
  CK(hipMemcpy(Z.data(), dZ, 4096 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Y.data(), dY, 4096 * 4, hipMemcpyDeviceToHost));
  { std::vector<double> X(1024); for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { double s = 0; for (int k = 0; k < 16; k++) s += A[i * 16 + k] * B[k * 32 + j]; X[i * 32 + j] = s; }
    // X values are multiples of 0.25 with |X| <= 64 -> exactly representable in bf16? (8 bits mantissa) not always; use bf16-rounded X
    double ez = 0, ey = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { double sz = 0, sy = 0; for (int k = 0; k < 32; k++) { sz += (double)bf2f(f2bf((float)X[k * 32 + i])) * B2[k * 32 + j]; sy += (double)A2[i * 32 + k] * bf2f(f2bf((float)X[k * 32 + j])); } ez = fmax(ez, fabs(sz - Z[i * 32 + j])); ey = fmax(ey, fabs(sy - Y[i * 32 + j])); }
    printf("T4 acc-as-operand: Z=X^T*B2 max err %g -> %s ; Y=A2*X max err %g -> %s\n", ez, ez < 1e-3 ? "PASS" : "FAIL", ey, ey < 1e-3 ? "PASS" : "FAIL"); }
  // test 5
  { std::vector<float> T(512); for (int m = 0; m < 16; m++) for (int n = 0; n < 32; n++) T[m * 32 + n] = ival(m, n, 9);
    for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) B[k * 32 + j] = ival(k, j, 10);
    CK(hipMemcpy(dA, T.data(), 512 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 4096 * 4, hipMemcpyHostToDevice));
    k_tr_mfma<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 4096 * 4, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { double s = 0; for (int k = 0; k < 16; k++) s += T[k * 32 + i] * B[k * 32 + j]; e = fmax(e, fabs(s - C[i * 32 + j])); }
    printf("T5 tr-read as MFMA A operand (A=T^T): max err %g -> %s\n", e, e == 0 ? "PASS" : "FAIL"); }
  printf("probe done\n");
  return 0;
}
