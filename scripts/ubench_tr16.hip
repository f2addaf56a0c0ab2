// Lane map of gfx950's ds_read_b64_tr_b16, measured: every lane supplies its own 8-byte-aligned LDS address; which
// (lane, 16-bit element) does each of a lane's four results come from?   Expected (and relied upon by the weight-gradient
// GEMM of csrc/ndq_mlp.h, Cfg::WG_TR): result j of lane l = element (l & 3) of the 8 bytes addressed by lane
// (l & ~15) + 4 j + ((l & 15) >> 2).      hipcc --offload-arch=gfx950 scripts/ubench_tr16.hip -o /tmp/ubench_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__global__ void k(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)((char*)lds + addr[threadIdx.x]));
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  for (int j = 0; j < 4; ++j) out[4 * threadIdx.x + j] = (unsigned short)(bits >> (16 * j));
}

int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
  int bad = 0;
  for (int variant = 0; variant < 3; ++variant) {
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, i = l & 15;
      if (variant == 0) h_addr[l] = g * 1024 + (i >> 2) * 64 + (i & 3) * 16;          // rows of 64 B, chunks 16 B apart
      else if (variant == 1) h_addr[l] = 8 * ((l * 37 + 11) % 1024);                    // scattered, all distinct
      else h_addr[l] = (g * 4 + (i >> 2)) * 32 + (i & 3) * 8;                           // the guide's dense image
    }
    hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int byte = 2 * h_out[4 * l + j];
        int src = -1, el = -1;
        for (int m = 0; m < 64; ++m) if (byte >= h_addr[m] && byte < h_addr[m] + 8) { src = m; el = (byte - h_addr[m]) / 2; }
        const int es = (l & ~15) + 4 * j + ((l & 15) >> 2), ee = l & 3;
        if (src != es || el != ee) {
          if (bad < 12 * (variant + 1)) printf("  lane %2d result %d: from lane %2d element %d (expected lane %2d element %d)\n", l, j, src, el, es, ee);
          ++bad;
        }
      }
  }
  printf(bad ? "MISMATCH: %d results differ from the expected lane map\n" : "lane map as expected (%d differences)\n", bad);
  return bad != 0;
}
