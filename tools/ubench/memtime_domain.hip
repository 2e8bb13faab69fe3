// Which hardware unit owns the s_memtime counter?  512 one-wave workgroups record (HW_ID, XCC_ID, s_memtime, s_memrealtime); the host prints,
// per XCC / SE / CU, the spread of (memtime - k * realtime) to see which grouping shares a counter.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
__global__ void k(unsigned long long* out) {
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    const unsigned xc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
    unsigned long long* o = out + (size_t)blockIdx.x * 4;
    o[0] = hw; o[1] = xc; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
  }
}
int main() {
  const int N = 512;
  unsigned long long* d; hipMalloc(&d, N * 32);
  std::vector<unsigned long long> h(N * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(N), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, N * 32, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 3; ++mode) {
      std::map<unsigned, std::vector<long long>> by;
      for (int i = 0; i < N; ++i) {
        const unsigned hw = (unsigned)h[i * 4], xcc = h[i * 4 + 1] & 0xf, se = (hw >> 13) & 7, cu = ((hw >> 8) & 0xf) | ((hw >> 12) & 1) << 4;
        const unsigned key = mode == 0 ? xcc : mode == 1 ? xcc * 8 + se : (xcc * 8 + se) * 32 + cu;
        by[key].push_back((long long)h[i * 4 + 2] - (long long)h[i * 4 + 3] * 21);
      }
      long long worst = 0; size_t groups = 0, multi = 0;
      for (auto& kv : by) {
        auto& v = kv.second; std::sort(v.begin(), v.end());
        ++groups; if (v.size() > 1) ++multi;
        worst = std::max(worst, v.back() - v.front());
      }
      printf("rep %d grouping %s: %zu groups (%zu with > 1 sample), worst spread inside a group %lld ticks\n", rep, mode == 0 ? "xcc" : mode == 1 ? "xcc,se" : "xcc,se,sh,cu", groups, multi, worst);
    }
    printf("sample rows: "); for (int i = 0; i < 4; ++i) printf("[hw %llx xcc %llx mt %llu rt %llu] ", h[i*4], h[i*4+1], h[i*4+2], h[i*4+3]); printf("\n");
  }
  return 0;
}
