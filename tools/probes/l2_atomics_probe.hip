// Are WORKGROUP-scope (no sc1) global float atomics performed in the XCD's own L2, coherently for every CU of that XCD, and how fast
// are they compared with agent-scope atomics and with plain slab stores? (Idea under test: split-K weight-gradient tiles accumulated
// with XCD-local atomics instead of fp32 slabs + a second-stage reduction.)
//   every workgroup reads HW_REG_XCC_ID (x), counts itself on cnt[x] (agent scope), then adds its 128 x 256 fp32 "tile" (values 1.0)
//   mode 0: workgroup-scope atomics into region[x]          -> expected region[x][i] == cnt[x] iff the XCD's L2 serialises them
//   mode 1: agent-scope atomics into ONE shared region      -> expected == number of workgroups
//   mode 2: plain 16-byte stores into slab[blockIdx]        (today's first stage; the second stage is not included)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int TILE = 128 * 256;
__global__ __launch_bounds__(256) void probe(float* region, float* slabs, unsigned* cnt, int mode, int reps) {
  const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;            // HW_REG_XCC_ID, bits [3:0]
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&cnt[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0 && x != blockIdx.x % 8) __hip_atomic_fetch_add(&cnt[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int r = 0; r < reps; ++r) {
    if (mode == 0) {
      float* g = region + (size_t)x * TILE;
      for (int i = threadIdx.x; i < TILE; i += 256) (void)__hip_atomic_fetch_add(g + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (mode == 1) {
      for (int i = threadIdx.x; i < TILE; i += 256) (void)__hip_atomic_fetch_add(region + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      float4* s = reinterpret_cast<float4*>(slabs + (size_t)blockIdx.x * TILE);
      for (int i = threadIdx.x; i < TILE / 4; i += 256) s[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
  }
}
int main() {
  float *region, *slabs; unsigned* cnt;
  const int maxwg = 2048;
  CK(hipMalloc(&region, (size_t)8 * TILE * 4)); CK(hipMalloc(&slabs, (size_t)maxwg * TILE * 4)); CK(hipMalloc(&cnt, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> h((size_t)8 * TILE);
  for (int nwg : {64, 256, 1024}) for (int mode : {0, 1, 2}) {
    const int reps = 1;
    float best = 1e9f; int bad = 0; unsigned c[16];
    for (int it = 0; it < 5; ++it) {
      CK(hipMemset(region, 0, (size_t)8 * TILE * 4)); CK(hipMemset(cnt, 0, 64));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, region, slabs, cnt, mode, reps);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      CK(hipMemcpy(h.data(), region, (size_t)8 * TILE * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, cnt, 64, hipMemcpyDeviceToHost));
      if (mode == 0) for (int x = 0; x < 8; ++x) for (int i = 0; i < TILE; ++i) bad += h[(size_t)x * TILE + i] != (float)(c[x] * reps);
      if (mode == 1) for (int i = 0; i < TILE; ++i) bad += h[i] != (float)(nwg * reps);
    }
    printf("nwg %4d mode %d (%s): %8.1f us per launch, %6.1f GB/s of tile bytes | wrong elements %d | per-XCD workgroups %u %u %u %u %u %u %u %u, off the b%%8 rule: %u\n",
           nwg, mode, mode == 0 ? "XCD-local workgroup-scope atomics" : mode == 1 ? "agent-scope atomics, one region" : "plain slab stores",
           best * 1e3, (double)nwg * TILE * 4 / (best * 1e-3) / 1e9, bad, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]);
  }
  return 0;
}
