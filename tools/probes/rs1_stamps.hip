// Stand-alone timing + phase stamps of the one-shot wide fused pointwise kernel (csrc/rsc1.cuh), random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRSC1_STAMPS tools/probes/rs1_stamps.hip -o /tmp/rs1_stamps && /tmp/rs1_stamps [C] [M] [mode] [cps] [rt]
#include "../../mmearth-train_amd/csrc/rsc1.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KC, int MODE, int RT, int CPS> static int run(int M, int wgs) {
  const int cps = CPS;
  const int HN = 4 * KC;
  std::vector<uint16_t> hA((size_t)M * KC), hW((size_t)HN * KC), hR((size_t)M * HN);
  srand(1);
  for (auto& v : hA) v = (uint16_t)(0x3f00 + (rand() & 0xff));
  for (auto& v : hW) v = (uint16_t)(0x3c00 + (rand() & 0xff));
  for (auto& v : hR) v = (uint16_t)(0x3f00 + (rand() & 0xff));
  std::vector<float> hv(HN, 0.5f);
  uint16_t *dA, *dW, *dR, *dout, *dxh, *dxn; float *dv0, *dv1, *dbias, *drstd, *dws; uint8_t* dact;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dR, hR.size() * 2)); CK(hipMalloc(&dout, hR.size() * 2));
  CK(hipMalloc(&dxh, hA.size() * 2)); CK(hipMalloc(&dxn, hA.size() * 2));
  CK(hipMalloc(&dv0, HN * 4)); CK(hipMalloc(&dv1, HN * 4)); CK(hipMalloc(&dbias, HN * 4)); CK(hipMalloc(&drstd, M * 4)); CK(hipMalloc(&dws, (size_t)64 << 20));
  CK(hipMalloc(&dact, M)); CK(hipMemset(dact, 1, M));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dR, hR.data(), hR.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dv0, hv.data(), HN * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv1, hv.data(), HN * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, hv.data(), HN * 4, hipMemcpyHostToDevice));
  RsP p{};
  p.A = dA; p.W = dW; p.ldw = KC; p.bias = dbias; p.v0 = dv0; p.v1 = dv1; p.out = dout; p.xhat = dxh; p.xn = dxn; p.rstd = drstd; p.R = dR; p.ws = dws; p.act = dact; p.M = M;
  p.perwave = 1;
  const size_t lds = (size_t)cps * KC * 2 + (size_t)4 * 2 * cps * 4 + (size_t)2 * KC * 4;
  CK(hipFuncSetAttribute((const void*)rsc_wide1_kernel<KC, MODE, RT, CPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int ntiles = (M + 64 * RT - 1) / (64 * RT), ny = HN / cps, gxmax = std::max(1, wgs / ny), tpw = (ntiles + gxmax - 1) / gxmax;
  dim3 g((ntiles + tpw - 1) / tpw, ny);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rsc_wide1_kernel<KC, MODE, RT, CPS>), g, dim3(256), lds, 0, p, ntiles);
  CK(hipDeviceSynchronize());
  const int reps = 30;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rsc_wide1_kernel<KC, MODE, RT, CPS>), g, dim3(256), lds, 0, p, ntiles);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nwg = g.x * g.y;
  printf("C=%d M=%d mode=%d cps=%d RT=%d grid=(%d,%d)=%d lds=%zu: %.1f us per launch\n", KC, M, MODE, cps, RT, g.x, g.y, nwg, lds, ms / reps * 1e3);
#ifdef RSC1_STAMPS
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((rsc_wide1_kernel<KC, MODE, RT, CPS>), g, dim3(256), lds, 0, p, ntiles);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st((size_t)4096 * 4 * 8);
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(rsc1_stamp_buf), st.size() * 8));
  const int n = std::min(nwg, 4096);
  double avg[6] = {0};
  unsigned long long w0 = ~0ull, w1 = 0;
  for (int b = 0; b < n; ++b) for (int w = 0; w < 4; ++w) { w0 = std::min(w0, st[((size_t)b * 4 + w) * 8 + 6]); w1 = std::max(w1, st[((size_t)b * 4 + w) * 8 + 7]); }
  for (int b = 0; b < n; ++b) for (int w = 0; w < 4; ++w) for (int k = 0; k < 6; ++k) avg[k] += (double)(st[((size_t)b * 4 + w) * 8 + k] - st[((size_t)b * 4 + w) * 8]);
  const char* names[6] = {"start", "every operand landed", "LN done (last tile)", "products done (last tile)", "barrier 2", "slab written"};
  double prev = 0;
  for (int k = 0; k < 6; ++k) { const double v = avg[k] / (n * 4.0); printf("  %-24s %9.0f ticks (+%7.0f)\n", names[k], v, v - prev); prev = v; }
  printf("  wall clock (100 MHz): first wave start -> last wave end %.2f us\n", (double)(w1 - w0) / 100.0);
  // dispatch profile: start / end of the workgroups in dispatch order (wave 0), relative to the first start
  std::vector<std::pair<double, double>> se;
  for (int b = 0; b < n; ++b) se.push_back({(double)(st[(size_t)b * 4 * 8 + 6] - w0) / 100.0, (double)(st[(size_t)b * 4 * 8 + 7] - w0) / 100.0});
  std::sort(se.begin(), se.end());
  for (int q = 0; q <= 10; ++q) { const int i = std::min(n - 1, q * n / 10); printf("  workgroup at %3d%% of start order: start %6.2f us, end %6.2f us (lives %5.2f)\n", q * 10, se[i].first, se[i].second, se[i].second - se[i].first); }
#endif
  return 0;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 320, M = argc > 2 ? atoi(argv[2]) : 4864, mode = argc > 3 ? atoi(argv[3]) : 0;
  const int cps = argc > 4 ? atoi(argv[4]) : (C == 320 ? 64 : 128), rt = argc > 5 ? atoi(argv[5]) : 1, wgs = argc > 6 ? atoi(argv[6]) : 768;
#define CASE(C_, MODE_, RT_, CPS_) if (C == C_ && mode == MODE_ && rt == RT_ && cps == CPS_) return run<C_, MODE_, RT_, CPS_>(M, wgs)
  CASE(320, 0, 1, 64); CASE(320, 0, 2, 64); CASE(320, 1, 1, 64); CASE(320, 1, 2, 64); CASE(320, 2, 1, 64); CASE(320, 2, 2, 64);
  CASE(160, 0, 1, 128); CASE(160, 0, 2, 128); CASE(160, 1, 1, 128); CASE(160, 1, 2, 128); CASE(160, 1, 1, 64); CASE(160, 0, 2, 64); CASE(160, 1, 2, 64); CASE(160, 2, 2, 64); CASE(160, 2, 1, 128);
  printf("unsupported\n");
  return 1;
}
