// Grid-barrier latency on MI355X (gfx950): what a persistent kernel would pay per chip-wide synchronisation instead of a
// kernel launch (BASELINE config 1 - 100k vertices x one signal - is 30 launches of 5.2 us replayed as one hipGraph).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
// Prints microseconds per barrier for 256 / 512 / 1024 workgroups, bare and with the release / acquire fences and a
// cross-workgroup data exchange through global memory that a recurrence step needs (and checks that exchange).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned* flag, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == gridDim.x * gen) {
      __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// two levels: workgroups of one XCD (blockIdx & 7: round-robin dispatch) arrive on their XCD's counter, the last of them on
// the global one; the release goes the same way back (one flag per XCD, each on its own 128-byte line)
__device__ __forceinline__ void grid_barrier2(unsigned* mem, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7u, per = (gridDim.x + 7u - x) / 8u;
    unsigned* cnt = mem + 64 + x * 32;     // per-XCD arrival counter
    unsigned* flg = mem + 64 + 256 + x * 32;  // per-XCD release flag
    const unsigned a = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (a == per * gen) {  // last of this XCD
      const unsigned g = __hip_atomic_fetch_add(mem, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (g == 8u * gen) __hip_atomic_store(mem + 32, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else while (__hip_atomic_load(mem + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(flg, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(flg, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// mode 0: barriers only.  mode 1: every workgroup writes a value per round, fences, barrier, reads another workgroup's
// value of this round (plain stores / loads between agent-scope fences) and counts mismatches.
// mode 2 / 3: the same with the two-level barrier (counter = its 4 KB of state); in mode 3 only the first wave of a
// workgroup executes the fences (the caches they write back / invalidate are shared by the workgroup)
__global__ void k_loop(unsigned* counter, unsigned* flag, int rounds, int mode, unsigned* box, unsigned* errors) {
  unsigned bad = 0;
  const bool exch = mode == 1 || mode == 3;
  for (int r = 0; r < rounds; ++r) {
    if (exch) {
      if (threadIdx.x == 0) box[(r & 1) * gridDim.x + blockIdx.x] = (unsigned)r * 4096u + blockIdx.x;
      if (mode == 1 || threadIdx.x < 64) __atomic_thread_fence(__ATOMIC_RELEASE);  // (agent scope on one device)
    }
    if (mode >= 2) grid_barrier2(counter, (unsigned)r + 1u);
    else grid_barrier(counter, flag, (unsigned)r + 1u);
    if (exch) {
      if (mode == 1 || threadIdx.x < 64) __atomic_thread_fence(__ATOMIC_ACQUIRE);
      __syncthreads();
      const unsigned peer = (blockIdx.x + 37u + threadIdx.x) % gridDim.x;
      const unsigned v = box[(r & 1) * gridDim.x + peer];
      if (v != (unsigned)r * 4096u + peer) ++bad;
    }
  }
  if (bad) atomicAdd(errors, bad);
}

int main() {
  unsigned *counter, *flag, *box, *errors;
  CK(hipMalloc(&counter, 4096)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&box, 2 * 4096 * 4)); CK(hipMalloc(&errors, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  for (int mode = 0; mode < 4; ++mode)
    for (int threads : {64, 256, 512})
      for (int grid : {256, 512, 1024}) {
        int fit = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_loop, threads, 0));
        if (fit * 256 < grid) continue;  // every workgroup must be resident
        float best = 1e30f;
        unsigned err = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(counter, 0, 4096)); CK(hipMemset(flag, 0, 4)); CK(hipMemset(errors, 0, 4));
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(k_loop, dim3(grid), dim3(threads), 0, 0, counter, flag, rounds, mode, box, errors);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        }
        printf("{\"mode\": \"%s\", \"workgroups\": %d, \"threads\": %d, \"us_per_barrier\": %.3f, \"exchange_errors\": %u}\n",
               mode == 0 ? "one counter, barrier only" : mode == 1 ? "one counter, fences + exchange" : mode == 2 ? "two levels, barrier only" : "two levels, fences by one wave + exchange", grid, threads, best * 1e3f / rounds, err);
      }
  return 0;
}
