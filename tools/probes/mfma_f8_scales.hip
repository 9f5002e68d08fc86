// GPU box probe: WHICH lane's scale byte applies to an element of A / B in v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3)?  One-hot product
// A(lane la, byte pa) x B(lane lb, byte pa) = 1; the scale of ONE lane X is doubled (E8M0 128) for A, then for B: the X at which the result
// becomes 2 is the lane whose scale that element uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__global__ void k_sc(int la, int lb, int pa, int cfg, float* out) {
  const int l = threadIdx.x;
  __shared__ float res;
  for (int side = 0; side < 2; side++) for (int X = 0; X < 64; X++) {
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == la) a[pa >> 2] = 0x38 << (8 * (pa & 3));
    if (l == lb) b[pa >> 2] = 0x38 << (8 * (pa & 3));
    const int sa = (side == 0 && l == X) ? 128 : 127, sb = (side == 1 && l == X) ? 128 : 127;
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    if (l == 0) res = 0.f;
    __syncthreads();
    for (int r = 0; r < 16; r++) if (c[r] != 0.f) res = c[r];
    __syncthreads();
    if (l == 0) out[(cfg * 2 + side) * 64 + X] = res;
    __syncthreads();
  }
}
int main() {
  float* d; hipMalloc(&d, 8 * 2 * 64 * 4);
  const int las[6] = {0, 5, 32, 37, 5, 37}, lbs[6] = {0, 3, 32, 35, 3, 35}, pas[6] = {0, 0, 0, 0, 17, 30};
  for (int c = 0; c < 6; c++) hipLaunchKernelGGL(k_sc, dim3(1), dim3(64), 0, 0, las[c], lbs[c], pas[c], c, d);
  std::vector<float> h(6 * 2 * 64);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int c = 0; c < 6; c++) for (int side = 0; side < 2; side++) {
    printf("A lane %2d x B lane %2d byte %2d, %s scale doubled in lane X -> result 2 at X =", las[c], lbs[c], pas[c], side ? "B" : "A");
    for (int X = 0; X < 64; X++) if (h[(c * 2 + side) * 64 + X] == 2.f) printf(" %d", X);
    printf("   (result at X = 63: %g)\n", h[(c * 2 + side) * 64 + 63]);
  }
  return 0;
}
