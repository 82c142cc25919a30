// What does a cross-lane signal cost the lane that raises it?  (a) plain dependent launches, (b) hipEventRecord behind every launch,
// (c) the launch's own completion as the event (hipExtLaunchKernelGGL stopEvent), (b2)/(c2) the same with a second stream waiting for every
// event and running a kernel of its own.  hipcc --offload-arch=gfx950 -O2 tools/probes/ext_event_probe.hip -o /tmp/ext_event_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void bump(float* x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] += 1.f; }
__global__ void spin(float* x, long long ticks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} if (threadIdx.x == 0) x[0] += 1.f; }
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
  const int N = 400, n = 1 << 16;
  float *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, n * 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(N); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s1));
      for (int i = 0; i < N; ++i) {
        if (mode == 2 || mode == 4) hipExtLaunchKernelGGL(bump, dim3(n / 256), dim3(256), 0, s1, nullptr, ev[i], 0, x, n);
        else hipLaunchKernelGGL(bump, dim3(n / 256), dim3(256), 0, s1, x, n);
        if (mode == 1 || mode == 3) CK(hipEventRecord(ev[i], s1));
        if (mode >= 3) { CK(hipStreamWaitEvent(s2, ev[i], 0)); hipLaunchKernelGGL(bump, dim3(n / 256), dim3(256), 0, s2, y, n); }
      }
      CK(hipEventRecord(t1, s1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      const char* nm[5] = {"plain", "hipEventRecord", "ext stopEvent", "hipEventRecord + wait on lane 2", "ext stopEvent + wait on lane 2"};
      if (rep) printf("%-34s %6.2f us per kernel on the signalling lane\n", nm[mode], ms * 1e3 / N);
    }
  // correctness of a wait on a stop event: lane 2 must see the spin kernel's write
  CK(hipMemset(x, 0, 4)); CK(hipMemset(y, 0, 4)); CK(hipDeviceSynchronize());
  int bad = 0;
  for (int i = 0; i < 50; ++i) {
    hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, nullptr, ev[i], 0, x, 20000LL);        // 200 us at 100 MHz
    CK(hipStreamWaitEvent(s2, ev[i], 0));
    CK(hipMemcpyAsync(y + 1 + i, x, 4, hipMemcpyDeviceToDevice, s2));
  }
  CK(hipDeviceSynchronize());
  std::vector<float> h(64); CK(hipMemcpy(h.data(), y, 64 * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < 50; ++i) if (h[1 + i] != (float)(i + 1)) ++bad;
  printf("wait on a stop event: %d of 50 hand-offs wrong\n", bad);
  return 0;
}
