// Reproducer for wrong results of a packed-fp32 instruction with op_sel on gfx950 (MI355X), seen in compiler-generated
// code (ROCm 7.2, clang 22):
//
//     v_pk_mul_f32  v[60:61], v[202:203], 0 op_sel_hi:[1,0]
//     v_pk_mul_f32  v[2:3], v[10:11], v[58:59] op_sel:[0,1]     ; low half = v10 * v59  (src1's HIGH half)
//     v_mfma_f32_16x16x32_bf16 a[52:55], v[116:119], v[82:85], a[52:55]
//
// In the fused closure kernel of one stream set, lanes 48..63 of the second packed multiply's LOW result intermittently
// came back as v10 * 0 -- the operand of the packed instruction before it -- which surfaced as one non-deterministic
// gradient entry.  Found by assembly-level bisection (one s_nop after that v_pk_mul makes the kernel bit-exact again);
// the file started life as a suspected write-after-read hazard, hence its name, but the result is wrong whether or not
// anything overwrites the sources afterwards.  This program replays the sequence with hard-coded registers and counts
// wrong lanes for several variants (profiles/archive/r01/r01u_pk_mfma_hazard.txt): ~11 % of the executions wrong as found; exact
// with one wait state or any non-MFMA instruction between the packed op and the MFMA, with scalar multiplies instead,
// or with the f32 MFMA as the follower.  Rewriting every packed-fp32 instruction that carries op_sel as two scalar
// instructions (neurodiffeq_amd/_hipcc.py) also cured a second, older corruption that only showed with two waves per
// SIMD.
//   hipcc --offload-arch=gfx950 -O2 neurodiffeq_amd/csrc/canary_pk_war.hip -o /tmp/pk_war && /tmp/pk_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define PRE                                                                                                   \
  "v_mov_b32 v10, %[x0]\n v_mov_b32 v11, %[x1]\n v_mov_b32 v202, %[y0]\n v_mov_b32 v203, %[y1]\n"           \
  "v_accvgpr_write_b32 a16, %[z0]\n v_accvgpr_write_b32 a17, %[z1]\n"                                       \
  "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"                         \
  "v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n"                             \
  "v_accvgpr_write_b32 a52, 0\n v_accvgpr_write_b32 a53, 0\n v_accvgpr_write_b32 a54, 0\n v_accvgpr_write_b32 a55, 0\n" \
  "v_accvgpr_write_b32 a40, 0\n v_accvgpr_write_b32 a41, 0\n v_accvgpr_write_b32 a42, 0\n v_accvgpr_write_b32 a43, 0\n" \
  "s_nop 7\n s_nop 7\n"                                                                                      \
  "v_mfma_f32_16x16x32_bf16 a[40:43], v[116:119], v[82:85], a[40:43]\n"                                     \
  "v_mfma_f32_16x16x32_bf16 a[52:55], v[116:119], v[82:85], a[52:55]\n"                                     \
  "v_mfma_f32_16x16x32_bf16 a[40:43], v[116:119], v[82:85], a[40:43]\n"                                     \
  "v_add_f32_e64 v58, v202, v202\n v_add_f32_e64 v59, v203, v203\n"                                         \
  "v_pk_mul_f32 v[60:61], v[202:203], 0 op_sel_hi:[1,0]\n"
#define POST                                                                                                  \
  "s_nop 7\n s_nop 7\n s_nop 7\n"                                                                            \
  "v_mov_b32 %[o0], v2\n v_mov_b32 %[o1], v3\n v_mov_b32 %[n0], v10\n"
#define MFMA "v_mfma_f32_16x16x32_bf16 a[52:55], v[116:119], v[82:85], a[52:55]\n"
#define PK "v_pk_mul_f32 v[2:3], v[10:11], v[58:59] op_sel:[0,1]\n"
#define RD "v_accvgpr_read_b32 v10, a16\n v_accvgpr_read_b32 v11, a17\n"

#define KERNEL(NAME, BODY)                                                                                    \
  __global__ __launch_bounds__(256) void NAME(const float* in, int* bad, int iters, float* dbg) {                        \
    const int t = blockIdx.x * blockDim.x + threadIdx.x;                                                     \
    int wrong = 0; float prev = 0.f;                                                                            \
    for (int it = 0; it < iters; ++it) {                                                                      \
      const float x0 = in[(t + it) & 1023] + 1.0f, x1 = x0 + 0.25f, y0 = 0.5f + 0.001f * (it & 7), y1 = y0 + 0.125f; \
      const float z0 = -x0 - 3.0f, z1 = -x1 - 5.0f;                                                          \
      float o0, o1, n0;                                                                                       \
      asm volatile(PRE BODY POST                                                                              \
                   : [o0] "=&v"(o0), [o1] "=&v"(o1), [n0] "=&v"(n0)                                           \
                   : [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [z0] "v"(z0), [z1] "v"(z1)       \
                   : "v2", "v3", "v10", "v11", "v58", "v59", "v60", "v61", "v202", "v203", "v82", "v83", "v84", \
                     "v85", "v116", "v117", "v118", "v119", "a16", "a17", "a40", "a41", "a42", "a43", "a52",  \
                     "a53", "a54", "a55");                                                                    \
      const float want0 = x0 * (2.0f * y1), want1 = x1 * (2.0f * y1);                                         \
      if (o0 != want0 || o1 != want1) { if (!wrong && blockIdx.x == 7 && threadIdx.x == 50) { dbg[0] = o0; dbg[1] = want0; dbg[2] = x0 * (2.0f * y0); dbg[3] = prev; dbg[4] = o1; dbg[5] = want1; dbg[6] = (float)it; } ++wrong; } prev = want0;                                                    \
    }                                                                                                         \
    if (wrong) atomicAdd(&bad[threadIdx.x & 63], wrong);                                                     \
  }

KERNEL(k_as_found, PK MFMA RD)                                     // the sequence as the compiler emitted it
KERNEL(k_no_mfma, PK RD)                                           // overwrite directly after the packed multiply
KERNEL(k_two_mfma, PK MFMA MFMA RD)
KERNEL(k_nop0, PK "s_nop 0\n" MFMA RD)                             // one wait state after the packed multiply
KERNEL(k_nop_after_mfma, PK MFMA "s_nop 0\n" RD)
KERNEL(k_scalar_mul, "v_mul_f32 v2, v10, v59\n v_mul_f32 v3, v11, v59\n" MFMA RD)   // unpacked multiplies instead
KERNEL(k_vmov, PK MFMA "v_mov_b32 v10, v202\n v_accvgpr_read_b32 v11, a17\n")       // plain VALU overwrite
KERNEL(k_valu_between, PK MFMA "v_add_f32 v60, v202, v203\n" RD)                   // an unrelated VALU op first
KERNEL(k_waitcnt, PK "s_waitcnt lgkmcnt(0)\n" MFMA RD)                              // a non-VALU instruction after the pk
KERNEL(k_late, PK MFMA "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n" RD)               // 32 wait states before the overwrite
KERNEL(k_hi_first, PK MFMA "v_accvgpr_read_b32 v11, a17\n v_accvgpr_read_b32 v10, a16\n")   // high half overwritten first
KERNEL(k_src1, PK MFMA "v_mov_b32 v59, v202\n")                                    // overwrite the OTHER source
KERNEL(k_f32_mfma, PK "v_mfma_f32_16x16x4_f32 a[52:55], v116, v82, a[52:55]\n" RD) // f32 MFMA instead of bf16
KERNEL(k_pk_fma, "v_pk_fma_f32 v[2:3], v[10:11], v[58:59], 0 op_sel:[0,1,0]\n" MFMA RD)
KERNEL(k_no_overwrite, PK MFMA)                                                   // nothing overwrites the sources
KERNEL(k_indep_mfma, PK "v_mfma_f32_16x16x32_bf16 a[40:43], v[116:119], v[82:85], 0\n" RD)   // MFMA without a srcC dependency
KERNEL(k_very_late, PK MFMA "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n" RD)
KERNEL(k_opsel_hi_only, "v_pk_mul_f32 v[2:3], v[10:11], v[58:59] op_sel_hi:[1,0]\n v_mul_f32 v2, v10, v59\n v_mul_f32 v3, v11, v59\n" "v_pk_mul_f32 v[60:61], v[10:11], v[58:59] op_sel_hi:[1,0]\n" MFMA)


// Cross-wave variant: 512 threads = 8 waves = 2 per SIMD.  Waves 0..3 run ONLY the packed multiplies (with one wait state
// after each -- the intra-wave fix), waves 4..7 ONLY bf16 MFMAs: if the hazard is about what the SIMD issues next, no
// matter from which wave, the packed results still go wrong.
__global__ __launch_bounds__(512) void k_cross_wave(const float* in, int* bad, int iters, float* dbg) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  int wrong = 0;
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
      const float x0 = in[(t + it) & 1023] + 1.0f, x1 = x0 + 0.25f, y0 = 0.5f + 0.001f * (it & 7), y1 = y0 + 0.125f;
      float o0, o1;
      asm volatile("v_mov_b32 v10, %[x0]\n v_mov_b32 v11, %[x1]\n v_mov_b32 v102, %[y0]\n v_mov_b32 v103, %[y1]\n"
                   "v_add_f32_e64 v58, v102, v102\n v_add_f32_e64 v59, v103, v103\n s_nop 1\n"
                   "v_pk_mul_f32 v[60:61], v[102:103], 0 op_sel_hi:[1,0]\n"
                   "v_pk_mul_f32 v[2:3], v[10:11], v[58:59] op_sel:[0,1]\n"
                   "s_nop 0\n"
                   "v_pk_mul_f32 v[60:61], v[102:103], 0 op_sel_hi:[1,0]\n"
                   "v_pk_mul_f32 v[4:5], v[10:11], v[58:59] op_sel:[0,1]\n"
                   "s_nop 7\n"
                   "v_mov_b32 %[o0], v2\n v_mov_b32 %[o1], v4\n"
                   : [o0] "=&v"(o0), [o1] "=&v"(o1)
                   : [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1)
                   : "v2", "v3", "v4", "v5", "v10", "v11", "v58", "v59", "v60", "v61", "v102", "v103");
      const float want = x0 * (2.0f * y1);
      if (o0 != want || o1 != want) ++wrong;
    }
  } else {
    for (int it = 0; it < iters; ++it)
      asm volatile("v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"
                   "v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n s_nop 1\n"
                   "v_mfma_f32_16x16x32_bf16 a[40:43], v[116:119], v[82:85], 0\n"
                   "v_mfma_f32_16x16x32_bf16 a[52:55], v[116:119], v[82:85], 0\n"
                   "v_mfma_f32_16x16x32_bf16 a[40:43], v[116:119], v[82:85], a[40:43]\n"
                   "v_mfma_f32_16x16x32_bf16 a[52:55], v[116:119], v[82:85], a[52:55]\n"
                   ::: "v82", "v83", "v84", "v85", "v116", "v117", "v118", "v119", "a40", "a41", "a42", "a43", "a52",
                       "a53", "a54", "a55");
  }
  if (wrong) atomicAdd(&bad[threadIdx.x & 63], wrong);
}

// library mode (tests/test_gpu_parity.py builds this file through neurodiffeq_amd._hipcc.compile_shared, i.e. WITH the
// assembly fix-up pass, and expects zero): wrong results of the sequence as the compiler emitted it
extern "C" long ndq_pk_war_count(int iters) {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 101) / 101.0f;
  float* in; int* bad; float* dbg;
  if (hipMalloc(&in, 4096) != hipSuccess || hipMalloc(&bad, 256) != hipSuccess || hipMalloc(&dbg, 64) != hipSuccess) return -1;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 256);
  hipLaunchKernelGGL(k_as_found, dim3(1024), dim3(256), 0, 0, in, bad, iters, dbg);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  int hb[64]; hipMemcpy(hb, bad, 256, hipMemcpyDeviceToHost);
  long tot = 0; for (int l = 0; l < 64; ++l) tot += hb[l];
  hipFree(in); hipFree(bad); hipFree(dbg);
  return tot;
}

#ifndef NDQ_PK_WAR_LIB
int main(int argc, char** argv) {
  const bool only_cross = argc > 1 && argv[1][0] == 'c';
  const int blocks = 1024, threads = 256, iters = 2000;
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 101) / 101.0f;
  float* in; int* bad; float* dbg; hipMalloc(&dbg, 64);
  hipMalloc(&in, 4096); hipMalloc(&bad, 256);
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  struct { const char* name; void (*k)(const float*, int*, int, float*); } ks[] = {
      {"as found: pk_mul, mfma, accvgpr_read", k_as_found}, {"no mfma in between", k_no_mfma},
      {"two mfma in between", k_two_mfma}, {"s_nop 0 after pk_mul", k_nop0},
      {"s_nop 0 after mfma", k_nop_after_mfma}, {"v_mul_f32 x2 instead of pk", k_scalar_mul},
      {"v_mov overwrite instead of accvgpr_read", k_vmov},
      {"unrelated VALU between mfma and overwrite", k_valu_between}, {"s_waitcnt between pk and mfma", k_waitcnt},
      {"32 wait states between mfma and overwrite", k_late}, {"high half overwritten first", k_hi_first},
      {"overwrite of src1 (v59) instead", k_src1}, {"f32 MFMA (16x16x4) in between", k_f32_mfma},
      {"v_pk_fma_f32 instead of v_pk_mul_f32", k_pk_fma},
      {"no overwrite at all", k_no_overwrite}, {"independent mfma (srcC = 0)", k_indep_mfma},
      {"128 wait states between mfma and overwrite", k_very_late}};
  for (auto& e : ks) {
    if (only_cross) break;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(bad, 0, 256); hipMemset(dbg, 0, 64);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, in, bad, iters, dbg);
      hipDeviceSynchronize();
      int hb[64]; hipMemcpy(hb, bad, 256, hipMemcpyDeviceToHost);
      long tot = 0, hi = 0; for (int l = 0; l < 64; ++l) { tot += hb[l]; if (l >= 48) hi += hb[l]; }
      printf("%-44s rep %d: wrong results %ld of %ld (lanes 48..63: %ld)\n", e.name, rep, tot,
             (long)blocks * threads * iters, hi);
      if (rep == 0 && tot) { float d[7]; hipMemcpy(d, dbg, 28, hipMemcpyDeviceToHost); printf("      sample (block 7 lane 50, iteration %g): got lo %.9g want %.9g  [x0*2*y0 = %.9g, previous iteration want = %.9g]  got hi %.9g want %.9g\n", d[6], d[0], d[1], d[2], d[3], d[4], d[5]); }
    }
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(bad, 0, 256);
    hipLaunchKernelGGL(k_cross_wave, dim3(blocks), dim3(512), 0, 0, in, bad, iters, dbg);
    hipDeviceSynchronize();
    int hb[64]; hipMemcpy(hb, bad, 256, hipMemcpyDeviceToHost);
    long tot = 0, hi = 0; for (int l = 0; l < 64; ++l) { tot += hb[l]; if (l >= 48) hi += hb[l]; }
    printf("%-44s rep %d: wrong results %ld of %ld (lanes 48..63: %ld)\n", "cross-wave: pk waves + mfma waves, 2 per SIMD", rep, tot,
           (long)blocks * 256 * iters, hi);
  }
  return 0;
}
#endif
