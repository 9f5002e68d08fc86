// GPU box probe: DISCOVER the k pairing and the result map of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3) with one-hot operands: byte pa of
// lane la of A and byte pb of lane lb of B are 1.0, everything else 0; a non-zero result says the two bytes share a k, and where it lands says
// which (row, column) the two lanes are.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// out[(cfg * 32 + pa) * 32 + pb] = (lane << 8 | reg) + 1 of the non-zero result element, 0 if none, -1 if several
__global__ void k_disc(int la, int lb, int cfg, int* out) {
  const int l = threadIdx.x;
  __shared__ int hit[2];
  for (int pa = 0; pa < 32; pa++) for (int pb = 0; pb < 32; pb++) {
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == la) a[pa >> 2] = 0x38 << (8 * (pa & 3));
    if (l == lb) b[pb >> 2] = 0x38 << (8 * (pb & 3));
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    if (l == 0) { hit[0] = 0; hit[1] = 0; }
    __syncthreads();
    for (int r = 0; r < 16; r++) if (c[r] != 0.f) { atomicAdd(&hit[0], 1); hit[1] = ((l << 8) | r) + 1; }
    __syncthreads();
    if (l == 0) out[(cfg * 32 + pa) * 32 + pb] = hit[0] == 0 ? 0 : (hit[0] == 1 ? hit[1] : -1);
    __syncthreads();
  }
}
int main() {
  int* d; hipMalloc(&d, 8 * 1024 * 4);
  const int las[8] = {0, 0, 32, 32, 1, 5, 33, 0}, lbs[8] = {0, 32, 0, 32, 0, 3, 2, 1};
  for (int c = 0; c < 8; c++) hipLaunchKernelGGL(k_disc, dim3(1), dim3(64), 0, 0, las[c], lbs[c], c, d);
  std::vector<int> h(8 * 1024);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int c = 0; c < 8; c++) {
    printf("A lane %2d x B lane %2d:", las[c], lbs[c]);
    int n = 0;
    for (int pa = 0; pa < 32; pa++) for (int pb = 0; pb < 32; pb++) {
      const int v = h[(c * 32 + pa) * 32 + pb];
      if (v != 0) { if (n < 40) printf(" a%d-b%d@L%d.r%d", pa, pb, (v - 1) >> 8, (v - 1) & 255); n++; }
    }
    printf("  (%d pairs)\n", n);
  }
  return 0;
}
