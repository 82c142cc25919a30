// Micro-benchmark behind csrc/ps.cuh's statistics exchange: NWG workgroups each add a 640-float partial vector into a global vector
// with device-scope float atomics (address stride `stride` floats), arrive on a counter, poll it, then read the vector back with sc1
// loads - the GRN exchange of one block. Reports us per round for several strides / variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float ald(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// no-return atomics + s_waitcnt vmcnt(0) before the arrival (variant A)
__global__ __launch_bounds__(512) void probe_noret(float* G, unsigned* sync, int H, int rounds, float* sink) {
  __shared__ float part[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < H; i += 512) part[i] = 1.0f + (float)(i & 7);
  __syncthreads();
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    float* g = G + (size_t)r * H;
    for (int j = tid; j < H; j += 512) (void)unsafeAtomicAdd(g + j, part[j]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * gridDim.x;
      while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    for (int j = tid; j < H; j += 512) acc += ald(g + j);
  }
  if (acc == 12345.f) sink[0] = acc;
}

// variant D: write-through slabs, arrival counter 1, NRED reducer workgroups sum column chunks over all slabs and publish, counter 2
__global__ __launch_bounds__(512) void probe_slab(float* slab, float* res, unsigned* sync, int H, int rounds, float* sink) {
  __shared__ float part[2048];
  __shared__ float red[512];
  const int tid = threadIdx.x, nwg = gridDim.x;
  for (int i = tid; i < H; i += 512) part[i] = 1.0f + (float)(i & 7);
  __syncthreads();
  float acc = 0.f;
  const int NRED = H / 16;                      // reducer r owns columns [16 r, 16 r + 16)
  for (int r = 0; r < rounds; ++r) {
    float* sl = slab + ((size_t)(r & 1) * nwg + blockIdx.x) * H;
    for (int j = tid; j < H; j += 512) __hip_atomic_store(reinterpret_cast<unsigned*>(sl + j), __float_as_uint(part[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)blockIdx.x < NRED) {
        const unsigned target = (unsigned)(r + 1) * nwg;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if ((int)blockIdx.x < NRED) {               // 512 threads: column c = tid & 15, slab group tid >> 4 (32 groups)
      const int c = tid & 15, gq = tid >> 4;
      float t = 0.f;
      for (int w = gq; w < nwg; w += 32) t += ald(slab + ((size_t)(r & 1) * nwg + w) * H + blockIdx.x * 16 + c);
      red[tid] = t;
      __syncthreads();
      if (tid < 16) {
        float u = 0.f;
        for (int q = 0; q < 32; ++q) u += red[q * 16 + tid];
        __hip_atomic_store(reinterpret_cast<unsigned*>(res + (size_t)r * H + blockIdx.x * 16 + tid), __float_as_uint(u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
      const unsigned target = (unsigned)(r + 1) * NRED;
      while (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    for (int j = tid; j < H; j += 512) acc += ald(res + (size_t)r * H + j);
  }
  if (acc == 12345.f) sink[0] = acc;
  if (blockIdx.x == 0 && tid == 0) sink[1] = ald(res + (size_t)(rounds - 1) * H);
}

// grouped variant: workgroup w adds into vector (w % NG) of NG vectors, after the barrier every workgroup sums the NG vectors
__global__ __launch_bounds__(512) void probe_grouped(float* G, unsigned* sync, int H, int NG, int rounds, float* sink) {
  __shared__ float part[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < H; i += 512) part[i] = 1.0f + (float)(i & 7);
  __syncthreads();
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    float* g = G + (size_t)r * H * NG;
    float s = 0.f;
    for (int j = tid; j < H; j += 512) s += unsafeAtomicAdd(g + (size_t)(blockIdx.x % NG) * H + j, part[j]);
    asm volatile("" ::"v"(s));
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * gridDim.x;
      while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    for (int j = tid; j < H; j += 512) {
      float t = 0.f;
      for (int q = 0; q < NG; ++q) t += ald(g + (size_t)q * H + j);
      acc += t;
    }
  }
  if (acc == 12345.f) sink[0] = acc;
}

// mode 0: atomics + barrier + readback; 1: barrier only; 2: atomics only (no barrier wait, no readback)
__global__ __launch_bounds__(512) void probe(float* G, unsigned* sync, int H, int stride, int rounds, int mode, float* sink) {
  __shared__ float part[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < H; i += 512) part[i] = 1.0f + (float)(i & 7);
  __syncthreads();
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    float* g = G + (size_t)r * H * stride;
    if (mode != 1) {
      float s = 0.f;
      for (int j = tid; j < H; j += 512) s += unsafeAtomicAdd(g + (size_t)j * stride, part[j]);
      asm volatile("" ::"v"(s));
    }
    __syncthreads();
    if (mode != 2) {
      if (tid == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(r + 1) * gridDim.x;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    }
    if (mode == 0) for (int j = tid; j < H; j += 512) acc += ald(g + (size_t)j * stride);
  }
  if (acc == 12345.f) sink[0] = acc;
}

int main() {
  const int H = 640, rounds = 50, NWG = 256;
  float *G, *sink; unsigned* sync;
  const size_t gbytes = (size_t)rounds * H * 1024 * 4;
  CK(hipMalloc(&G, gbytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&sync, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nwg : {NWG, 64}) for (int mode : {0, 1, 2}) for (int stride : {1, 16, 64, 1024}) {
    if (mode == 1 && stride != 1) continue;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(G, 0, gbytes)); CK(hipMemset(sync, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 0, 0, G, sync, H, stride, rounds, mode, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    std::vector<float> h(4);
    CK(hipMemcpy(h.data(), G, 16, hipMemcpyDeviceToHost));
    printf("nwg %3d mode %d (%s) stride %4d floats: %7.2f us per round   (G[0] = %.0f)\n", nwg, mode,
           mode == 0 ? "atomics+barrier+readback" : mode == 1 ? "barrier only" : "atomics only", stride, best * 1e3f / rounds, h[0]);
  }
  for (int nwg : {256, 64}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(G, 0, gbytes)); CK(hipMemset(sync, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe_noret, dim3(nwg), dim3(512), 0, 0, G, sync, H, rounds, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    std::vector<float> h(4);
    CK(hipMemcpy(h.data(), G, 16, hipMemcpyDeviceToHost));
    printf("nwg %3d no-return atomics + vmcnt(0) + barrier + readback: %7.2f us per round (G[0] = %.0f)\n", nwg, best * 1e3f / rounds, h[0]);
    best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(G, 0, gbytes)); CK(hipMemset(sync, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe_slab, dim3(nwg), dim3(512), 0, 0, G, G + (size_t)2 * 256 * H, sync, H, rounds, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CK(hipMemcpy(h.data(), sink, 16, hipMemcpyDeviceToHost));
    printf("nwg %3d write-through slabs + 40 reducers + two counters:     %7.2f us per round (last res[0] = %.0f)\n", nwg, best * 1e3f / rounds, h[1]);
  }
  for (int nwg : {256}) for (int NG : {1, 4}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(G, 0, gbytes)); CK(hipMemset(sync, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe_grouped, dim3(nwg), dim3(512), 0, 0, G, sync, H, NG, rounds, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("nwg %3d grouped atomics, %2d vectors + barrier + readback of all vectors: %7.2f us per round\n", nwg, NG, best * 1e3f / rounds);
  }
  return 0;
}
