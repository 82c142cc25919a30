// stand-alone A/B of gemm_nt4_kernel<256> with plain vs sc1 output stores (compile twice: -DNT4_SC1_STORES)
#include "/root/repo/mmearth-train_amd/csrc/gemm.cuh"
#include "/root/repo/mmearth-train_amd/csrc/gemm_nt4.cuh"
#include <cstdio>
#include <vector>
int main() {
  const int M = 12544, N = 2048, K = 512;
  bf16_t *A, *B, *C; float* bias;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
  hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(B, 0x3c, (size_t)N * K * 2); hipMemset(bias, 0, N * 4);
  GemmP p{}; p.A = A; p.B = B; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
  using Cf = Nt4Cfg<256>;
  hipFuncSetAttribute((const void*)gemm_nt4_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS);
  const int tm = (M + 255) / 256, tn = N / 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(gemm_nt4_kernel<256>, dim3((tm + 7) / 8 * 8 * tn), dim3(512), Cf::LDS, 0, p, tm, tn);
  hipEventRecord(e0);
  for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(gemm_nt4_kernel<256>, dim3((tm + 7) / 8 * 8 * tn), dim3(512), Cf::LDS, 0, p, tm, tn);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.1f us per GEMM (%d x %d x %d) err %s\n",
#ifdef NT4_SC1_STORES
         "sc1 stores  ",
#else
         "plain stores",
#endif
         ms / 50 * 1e3, M, N, K, hipGetErrorString(hipGetLastError()));
  return 0;
}
