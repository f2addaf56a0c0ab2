// Which element of D = A B does lane l, register r of v_mfma_f64_16x16x4_f64 hold?  (scripts/, experiments only)
//   build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_mfma_f64_layout.hip -o /tmp/f64layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  // A[i][k]: lane (i = lane % 16, k = lane / 16) -> value 100 i + k ;  B[k][j]: lane (j = lane % 16, k = lane / 16) -> (k == 0) ? 1 : 0 ... D[i][j] = A[i][0] = 100 i
  const double a = 100.0 * (lane % 16) + (lane / 16);
  const double b0 = (lane / 16 == 0) ? 1.0 : 0.0;
  d4 c = {0, 0, 0, 0};
  d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, c, 0, 0, 0);
  // second product: B[k][j] = (k == 0) ? j : 0, A[i][0] = 1 -> D[i][j] = j
  const double a1 = (lane / 16 == 0) ? 1.0 : 0.0;
  const double b1 = (lane / 16 == 0) ? (double)(lane % 16) : 0.0;
  d4 e = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { out[(lane * 4 + r) * 2] = d[r]; out[(lane * 4 + r) * 2 + 1] = e[r]; }
}
int main() {
  double* o; hipMalloc(&o, 64 * 4 * 2 * 8);
  k<<<1, 64>>>(o);
  double h[64 * 4 * 2]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; lane += 5)
    for (int r = 0; r < 4; ++r) printf("lane %2d reg %d: row %g col %g\n", lane, r, h[(lane * 4 + r) * 2] / 100.0, h[(lane * 4 + r) * 2 + 1]);
  return 0;
}
