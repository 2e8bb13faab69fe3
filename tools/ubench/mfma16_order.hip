// Does v_mfma_f32_16x16x4_f32 accumulate its four k products in ascending k with one rounding each (= an fmaf chain)?
// Compares, bit for bit, C = A(16x64) . B(64x16) accumulated by 16 MFMAs against fmaf chains on the host.  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* C, int K) {
  const int l = threadIdx.x, row = l & 15, kq = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[row * K + k0 + kq], B[(k0 + kq) * 16 + row], acc, 0, 0, 0);
  // C layout: col = l & 15, rows 4 * (l >> 4) + r
  for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
int main() {
  const int K = 64;
  std::vector<float> A(16 * K), B(K * 16), C(256), R(256), R2(256);
  std::mt19937 g(1); std::normal_distribution<float> d(0.f, 1.f);
  for (auto& v : A) v = d(g);
  for (auto& v : B) v = d(g);
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  int same_chain = 0, same_pair = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float acc = 0.f;
    for (int kk = 0; kk < K; ++kk) acc = fmaf(A[i * K + kk], B[kk * 16 + j], acc);
    R[i * 16 + j] = acc;
    same_chain += memcmp(&acc, &C[i * 16 + j], 4) == 0;
  }
  printf("16x16x4 f32: %d of 256 results bit-identical to the ascending-k fmaf chain (max |diff| %g)\n", same_chain,
         [&] { float m = 0; for (int i = 0; i < 256; ++i) m = fmaxf(m, fabsf(C[i] - R[i])); return m; }());
  return 0;
}
