// Stand-alone timing + phase stamps of the matrix-core depthwise WEIGHT-GRADIENT kernel (csrc/dwmfma_wg.cuh); random masks (19 of 49).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDWM_STAMPS tools/probes/dwwg_stamps.hip -o tools/probes/dwwg_stamps && tools/probes/dwwg_stamps [C] [N]
#include "../../mmearth-train_amd/csrc/dwmfma_wg.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CCH> static int run(int C, int N) {
  const int G = 7, L = 49, keep = 19;
  const size_t M = (size_t)N * keep * 64;
  std::vector<int> vis(N * keep), inv(N * L, -1);
  srand(1);
  for (int n = 0; n < N; ++n) {
    std::vector<int> perm(L);
    for (int i = 0; i < L; ++i) perm[i] = i;
    for (int i = L - 1; i > 0; --i) std::swap(perm[i], perm[rand() % (i + 1)]);
    std::sort(perm.begin(), perm.begin() + keep);
    for (int k = 0; k < keep; ++k) { vis[n * keep + k] = perm[k]; inv[n * L + perm[k]] = k; }
  }
  std::vector<uint16_t> hx(M * C);
  for (auto& v : hx) v = (uint16_t)(0x3f00 + (rand() & 0xff));
  int *dvis, *dinv; uint16_t *dx, *ddd; float* ws;
  CK(hipMalloc(&dvis, vis.size() * 4)); CK(hipMalloc(&dinv, inv.size() * 4));
  CK(hipMalloc(&dx, M * C * 2)); CK(hipMalloc(&ddd, M * C * 2)); CK(hipMalloc(&ws, (size_t)N * 50 * C * 4));
  CK(hipMemcpy(dvis, vis.data(), vis.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, hx.data(), M * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ddd, hx.data(), M * C * 2, hipMemcpyHostToDevice));
  DwWgP a{};
  a.x = dx; a.dd = ddd; a.s_kh = 7 * C; a.s_kw = C; a.s_c = 1;
  a.g.vis = dvis; a.g.inv = dinv; a.g.N = N; a.g.keep = keep; a.g.grid = G; a.g.S = 8; a.C = C; a.CC = 8; a.TP = 1; a.tiles_side = 1; a.ws = ws;
  DwWgGroupP gr; gr.count = 0;
  using D = DwMfmaWg<CCH>;
  const size_t lds = D::lds(keep);
  CK(hipFuncSetAttribute((const void*)dwconv7_wgrad_mfma_kernel<CCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((dwconv7_wgrad_mfma_kernel<CCH>), dim3(N, C / CCH, 1), dim3(D::NT), lds, 0, a, gr);
  CK(hipDeviceSynchronize());
  const int reps = 30;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dwconv7_wgrad_mfma_kernel<CCH>), dim3(N, C / CCH, 1), dim3(D::NT), lds, 0, a, gr);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("wgrad S=8 C=%d CCH=%d N=%d lds=%zu: %.1f us per launch (without the fold)\n", C, CCH, N, lds, ms / reps * 1e3);
#ifdef DWM_STAMPS
  const int nwg = N * (C / CCH), NW = D::NT / 64, K = 16;
  std::vector<unsigned long long> st((size_t)nwg * 16 * 48);
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(dwm_stamp_buf), st.size() * 8));
  const char* names[16] = {"start", "loads issued", "geometry + X transposes", "D part 0 staged", "barrier", "MFMAs part 0", "D part 1 staged (load exposed)", "barrier",
                           "MFMAs part 1", "D part 2 staged (load exposed)", "barrier", "MFMAs part 2", "barrier", "diagonals (LDS atomics)", "barrier", "slab written"};
  double prev = 0;
  for (int k = 0; k < K; ++k) {
    double v = 0;
    for (int b = 0; b < nwg; ++b)
      for (int w = 0; w < NW; ++w) v += (double)(st[((size_t)b * 16 + w) * 48 + k] - st[((size_t)b * 16 + w) * 48]);
    v /= (double)nwg * NW;
    printf("  %-32s %9.0f cycles (+%7.0f)\n", names[k], v, v - prev);
    prev = v;
  }
#endif
  return 0;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 40, N = argc > 2 ? atoi(argv[2]) : 256;
  if (C % 40 == 0) return run<40>(C, N);
  if (C % 32 == 0) return run<32>(C, N);
  return 1;
}
