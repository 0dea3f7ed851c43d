// hwid.hip - where do the waves of a workgroup land?  Prints (workgroup, wave) -> XCC / SE / CU / SIMD / wave slot
// from HW_REG_HW_ID, for 4-wave and 16-wave workgroups.  Build: hipcc --offload-arch=gfx950 -O2 -o hwid_ubench hwid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID, bits 3:0
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;           // keep the wave resident for a while
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[2 * w] = hw;
    out[2 * w + 1] = xcc | (x == 12345.f ? 1u << 31 : 0u);
  }
}

int main() {
  for (int waves : {4, 7, 16}) {
    const int nwg = 2048;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 2 * nwg * waves);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(64 * waves), 0, 0, d, 20000);
    std::vector<unsigned> h(2 * nwg * waves);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("== %d waves per workgroup: first 6 workgroups (wave: simd/slot)\n", waves);
    for (int g = 0; g < 6; ++g) {
      printf("wg %d xcc %u se %u cu %u :", g, h[2 * g * waves + 1] & 15, (h[2 * g * waves] >> 13) & 7, (h[2 * g * waves] >> 8) & 15);
      for (int w = 0; w < waves; ++w) printf(" %u/%u", (h[2 * (g * waves + w)] >> 4) & 3, h[2 * (g * waves + w)] & 15);
      printf("\n");
    }
    // histogram: how often is wave w of a workgroup on SIMD s
    int hist[16][4] = {};
    int same_slot = 0;
    for (int g = 0; g < nwg; ++g) {
      bool s = true;
      for (int w = 0; w < waves; ++w) {
        hist[w][(h[2 * (g * waves + w)] >> 4) & 3]++;
        s = s && ((h[2 * (g * waves + w)] & 15) == (h[2 * g * waves] & 15));
      }
      same_slot += s;
    }
    for (int w = 0; w < waves; ++w) printf("wave %2d on simd0..3: %5d %5d %5d %5d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("workgroups whose waves all share one slot id: %d of %d\n", same_slot, nwg);
    hipFree(d);
  }
  return 0;
}
