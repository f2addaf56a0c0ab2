// One-shot all-reduce of the small [gradient | loss] message of data-parallel training (4.7 KB at C2 ...
// ~103 KB at C5 -- latency-bound, so the collective's own latency decides the scaling efficiency).  Instead of a
// ring / tree, every rank WRITES its vector straight into every peer's inbox over xGMI (MI355X: 7 direct links per
// GPU, one hop to every peer of the node) and then adds the R vectors it has received itself, in rank order -- every
// rank performs the same additions in the same order, so replicas stay bit-identical.  ONE kernel launch per call:
//   push   peer q's inbox[parity][my rank][:] = my vector   (remote stores), system-scope fence, flag = step
//   wait   until my own flags of this parity show `step` from every rank (system-scope acquire; the spin is bounded
//          only by `spin_limit` iterations of ~1 us -- default 2^28, minutes: a rank that lags behind a checkpoint, a
//          JIT build or a garbage collection is waited for, like RCCL would; a wait that does run out POISONS the
//          result (NaN) and counts in the status word, which the solver checks at every history flush and turns into
//          an error -- nobody trains on a stale inbox)
//   sum    recv[i] = inbox[parity][0][i] + inbox[parity][1][i] + ...   (fixed order)
// Inboxes are fine-grained (uncached) device memory shared through HIP IPC; two parities double-buffer successive calls:
// a rank can only be one call ahead of its slowest peer (it needs that peer's flag of the current call to finish it),
// so the buffer of call k is never overwritten before every rank has summed it.
// The entry point has ncclAllReduce's signature, so ndq_fused_step.allreduce / .comm take it unchanged (include/ndq.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <new>

namespace ndq {

constexpr int kOneshotMaxRanks = 16;
constexpr int kOneshotChunk = 4096;          // floats per workgroup
constexpr unsigned long long kOneshotSpinLimit = 1ull << 28;   // default; NDQ_ONESHOT_SPIN_LIMIT overrides (0: no limit)

struct OneshotDev {                 // what the kernel needs; lives in the host-side context, passed by value
  float* inbox[kOneshotMaxRanks];           // peer q's inbox base (q == rank: my own), layout [2][world][max_len]
  unsigned* flags[kOneshotMaxRanks];        // peer q's flags base, layout [2][world][max_blocks]
  unsigned* status;                         // my status word: number of flag waits that timed out
  int rank, world, max_len, max_blocks;
  unsigned long long spin_limit;            // iterations a flag wait may take (0: unbounded)
};

// wait until *f == step; false if the spin limit ran out first (counted in *status)
__device__ __forceinline__ bool oneshot_wait(const unsigned* f, unsigned step, unsigned long long limit, unsigned* status) {
  unsigned long long spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != step) {
    if (limit != 0 && ++spins > limit) {
      atomicAdd(status, 1u);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

__global__ __launch_bounds__(1024) void oneshot_allreduce_kernel(OneshotDev c, const float* __restrict__ send,
                                                                 float* __restrict__ recv, int len, unsigned step) {
  const int parity = step & 1u, blk = blockIdx.x;
  const int lo = blk * kOneshotChunk;
  const int hi = lo + kOneshotChunk < len ? lo + kOneshotChunk : len;
  const size_t slot = ((size_t)parity * c.world + c.rank) * c.max_len;
  // ---- push my chunk into every inbox (my own included)
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float v = send[i];
    for (int q = 0; q < c.world; ++q) __hip_atomic_store(c.inbox[q] + slot + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < c.world) {
    unsigned* f = c.flags[threadIdx.x] + ((size_t)parity * c.world + c.rank) * c.max_blocks + blk;
    __hip_atomic_store(f, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- wait for every rank's chunk of this call
  __shared__ int timed_out;
  if (threadIdx.x == 0) timed_out = 0;
  __syncthreads();
  if ((int)threadIdx.x < c.world) {
    const unsigned* f = c.flags[c.rank] + ((size_t)parity * c.world + threadIdx.x) * c.max_blocks + blk;
    if (!oneshot_wait(f, step, c.spin_limit, c.status)) timed_out = 1;     // a peer never arrived
  }
  __syncthreads();
  const bool bad = timed_out != 0;
  // ---- fixed-order sum (a chunk whose wait ran out is NaN: whatever consumes it cannot go unnoticed)
  const float* mine = c.inbox[c.rank] + (size_t)parity * c.world * c.max_len;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    float s = __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int q = 1; q < c.world; ++q)
      s += __hip_atomic_load(mine + (size_t)q * c.max_len + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    recv[i] = bad ? __builtin_nanf("") : s;
  }
}

struct Oneshot {
  OneshotDev dev;
  void* base;                      // my allocation: [inbox floats | flags | status]
  void* peer_base[kOneshotMaxRanks];
  size_t inbox_bytes, flag_bytes;
  unsigned step;
};

inline size_t oneshot_layout(int world, int max_len, size_t* inbox_bytes, size_t* flag_bytes, int* max_blocks) {
  *max_blocks = 1024;      // flags per (parity, source rank): the stand-alone call uses ceil(len / 4096) of them, the fused
                           // data-parallel tail kernel (ndq_api.hip: reduce_tail_dp_kernel) one per 64 gradient columns
  if ((max_len + kOneshotChunk - 1) / kOneshotChunk > *max_blocks) *max_blocks = (max_len + kOneshotChunk - 1) / kOneshotChunk;
  *inbox_bytes = ((size_t)2 * world * max_len * sizeof(float) + 255) & ~(size_t)255;
  *flag_bytes = ((size_t)2 * world * *max_blocks * sizeof(unsigned) + 255) & ~(size_t)255;
  return *inbox_bytes + *flag_bytes + 256;
}

}  // namespace ndq
