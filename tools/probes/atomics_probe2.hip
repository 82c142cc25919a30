// Phase stamps of the GRN exchange (see atomics_probe.hip): per round, workgroup 0 records shader-cycle stamps after
// (a) its atomics are acknowledged, (b) the arrival counter reached the target, (c) the read-back. Variants: NG result vectors
// (workgroup w adds into vector w % NG), rotated start column, returning / non-returning atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ float ald(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__global__ __launch_bounds__(512) void probe(float* G, unsigned* sync, int H, int NG, int rot, int ret, int rounds, float* sink, long long* stamps) {
  __shared__ float part[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < H; i += 512) part[i] = 1.0f + (float)(i & 7);
  __syncthreads();
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    float* g = G + (size_t)r * H * NG;
    const long long t0 = clock64();
    float s = 0.f;
    const int start = rot ? (blockIdx.x * 80) % H : 0;
    for (int j = tid; j < H; j += 512) {
      const int jj = (j + start) % H;
      if (ret) s += unsafeAtomicAdd(g + (size_t)(blockIdx.x % NG) * H + jj, part[jj]);
      else (void)unsafeAtomicAdd(g + (size_t)(blockIdx.x % NG) * H + jj, part[jj]);
    }
    if (ret) asm volatile("" ::"v"(s)); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) {
      __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * gridDim.x;
      while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    const long long t2 = clock64();
    for (int j = tid; j < H; j += 512) {
      float t = 0.f;
      for (int q = 0; q < NG; ++q) t += ald(g + (size_t)q * H + j);
      acc += t;
    }
    asm volatile("" ::"v"(acc));
    __syncthreads();
    const long long t3 = clock64();
    if (blockIdx.x == 0 && tid == 0) { stamps[r * 3 + 0] = t1 - t0; stamps[r * 3 + 1] = t2 - t1; stamps[r * 3 + 2] = t3 - t2; }
  }
  if (acc == 12345.f) sink[0] = acc;
}
int main() {
  const int H = 640, rounds = 40;
  float *G, *sink; unsigned* sync; long long* st;
  const size_t gbytes = (size_t)rounds * H * 64 * 4;
  CK(hipMalloc(&G, gbytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&sync, 64)); CK(hipMalloc(&st, rounds * 3 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nwg : {256}) for (int ret : {0, 1}) for (int rot : {0, 1}) for (int NG : {1, 4, 16}) {
    CK(hipMemset(G, 0, gbytes)); CK(hipMemset(sync, 0, 64));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 0, 0, G, sync, H, NG, rot, ret, rounds, sink, st);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(rounds * 3);
    CK(hipMemcpy(h.data(), st, rounds * 24, hipMemcpyDeviceToHost));
    double a = 0, b = 0, c = 0;
    for (int r = 5; r < rounds; ++r) { a += h[r * 3]; b += h[r * 3 + 1]; c += h[r * 3 + 2]; }
    const int n = rounds - 5;
    printf("nwg %d ret %d rot %d NG %2d: %6.2f us/round | wg0 cycles: atomics %6.0f  barrier %6.0f  readback %6.0f\n", nwg, ret, rot, NG,
           ms * 1e3 / rounds, a / n, b / n, c / n);
  }
  return 0;
}
