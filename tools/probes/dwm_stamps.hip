// Stand-alone timing + phase stamps of the matrix-core depthwise kernel (csrc/dwmfma.cuh), random masks (19 of 49 patches), random rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDWM_STAMPS tools/probes/dwm_stamps.hip -o tools/probes/dwm_stamps && tools/probes/dwm_stamps [S] [C] [N] [add]
#include "../../mmearth-train_amd/csrc/dwmfma.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int S, int CCH> static int run(int C, int N, int with_add) {
  const int G = 7, L = 49, keep = 19;
  const size_t M = (size_t)N * keep * S * S;
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
  std::vector<float> hw(49 * C);
  for (auto& v : hw) v = (float)(rand() % 200 - 100) / 300.f;
  int *dvis, *dinv; uint16_t *dx, *dout, *dadd; float* dw; uint8_t* dact;
  CK(hipMalloc(&dvis, vis.size() * 4)); CK(hipMalloc(&dinv, inv.size() * 4));
  CK(hipMalloc(&dx, M * C * 2)); CK(hipMalloc(&dout, M * C * 2)); CK(hipMalloc(&dadd, M * C * 2)); CK(hipMalloc(&dw, hw.size() * 4));
  CK(hipMalloc(&dact, M));
  CK(hipMemcpy(dvis, vis.data(), vis.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, hx.data(), M * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dadd, hx.data(), M * C * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dact, 1, M));
  DwP a{};
  a.x = dx; a.out = dout; a.add = with_add ? dadd : nullptr; a.w = dw; a.bias = nullptr; a.s_kh = 7 * C; a.s_kw = C; a.s_c = 1; a.flip = with_add;
  a.g.vis = dvis; a.g.inv = dinv; a.g.N = N; a.g.keep = keep; a.g.grid = G; a.g.S = S; a.C = C; a.CC = 8; a.TP = 1; a.tiles_side = 1; a.act = dact;
  using D = DwMfma<S, CCH>;
  const size_t lds = D::lds(keep);
  CK(hipFuncSetAttribute((const void*)dwconv7_mfma_kernel<S, CCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((dwconv7_mfma_kernel<S, CCH>), dim3(N, C / CCH), dim3(D::NT), lds, 0, a);
  CK(hipDeviceSynchronize());
  const int reps = 30;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dwconv7_mfma_kernel<S, CCH>), dim3(N, C / CCH), dim3(D::NT), lds, 0, a);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("S=%d C=%d CCH=%d N=%d add=%d lds=%zu: %.1f us per launch (%.2f TB/s of 1r + 1w)\n", S, C, CCH, N, with_add, lds, ms / reps * 1e3,
         2.0 * M * C * 2 / (ms / reps * 1e-3) / 1e12);
#ifdef DWM_STAMPS
  const int nwg = N * (C / CCH), NW = D::NT / 64;
  std::vector<unsigned long long> st((size_t)nwg * 16 * 48);
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(dwm_stamp_buf), st.size() * 8));
  const int npass = S == 8 ? 3 : 2;
  const int K = 6 + 6 * npass;
  std::vector<double> avg(K, 0.0);
  for (int b = 0; b < nwg; ++b)
    for (int w = 0; w < NW; ++w)
      for (int k = 0; k < K; ++k) avg[k] += (double)(st[((size_t)b * 16 + w) * 48 + k] - st[((size_t)b * 16 + w) * 48]);
  const char* names[6] = {"mfma done", "barrier 1", "add staged", "epilogue", "barrier 3", "copy-out issued"};
  const char* pro[6] = {"start", "loads issued", "small operands + barrier", "toeplitz build", "transposes", "barrier"};
  double prev = 0;
  for (int k = 0; k < K; ++k) {
    const double v = avg[k] / ((double)nwg * NW);
    printf("  %-28s %9.0f ticks (+%7.0f)\n", k < 6 ? pro[k] : names[(k - 6) % 6], v, v - prev);
    prev = v;
  }
  printf("  (s_memtime ticks = shader cycles, ~2.1 GHz; averages over every wave of every workgroup)\n");
#endif
  return 0;
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 8, C = argc > 2 ? atoi(argv[2]) : 40, N = argc > 3 ? atoi(argv[3]) : 256, add = argc > 4 ? atoi(argv[4]) : 0;
  if (S == 8 && C % 40 == 0) return run<8, 40>(C, N, add);
  if (S == 8 && C % 32 == 0) return run<8, 32>(C, N, add);
  if (S == 4 && C % 40 == 0) return run<4, 40>(C, N, add);
  if (S == 4 && C % 32 == 0) return run<4, 32>(C, N, add);
  printf("unsupported\n");
  return 1;
}
