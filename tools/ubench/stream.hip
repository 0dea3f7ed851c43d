// micro-benchmark: the HBM traffic pattern of the Griffin-Lim iteration kernel without any arithmetic
// (per slot: 4 B + 8 B read, 8 B written), frame-sequential per workgroup, to find the memory ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using v4f = float __attribute__((ext_vector_type(4)));
constexpr int QP = 448, STRIDE = 9408;
template <int NT>
__global__ void __launch_bounds__(512) k(const float* __restrict__ S, float* __restrict__ tp, int frames_per_wg, int nthreads_active) {
  const unsigned q = threadIdx.x;
  if ((int)q >= nthreads_active) return;
  for (int f = 0; f < frames_per_wg; ++f) {
    const size_t fr = (size_t)blockIdx.x * frames_per_wg + f;
    const v4f* s4 = reinterpret_cast<const v4f*>(S + fr * STRIDE);
    v4f* t4 = reinterpret_cast<v4f*>(tp + fr * STRIDE * 2);
    v4f a[5], b[10];
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = NT ? __builtin_nontemporal_load(s4 + i * QP + q) : s4[i * QP + q];
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = NT ? __builtin_nontemporal_load(t4 + i * QP + q) : t4[i * QP + q];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      v4f r = b[i] * a[i / 2].x + a[i / 2];
      if (NT) __builtin_nontemporal_store(r, t4 + i * QP + q); else t4[i * QP + q] = r;
    }
  }
}
int main(int argc, char** argv) {
  const int B = 64, T = 512;
  const size_t nfr = (size_t)B * T;
  float *S, *tp;
  hipMalloc(&S, nfr * STRIDE * 4); hipMalloc(&tp, nfr * STRIDE * 8);
  hipMemset(S, 0, nfr * STRIDE * 4); hipMemset(tp, 0, nfr * STRIDE * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nt = 0; nt < 2; ++nt)
    for (int nwg : {256, 512, 1024, 2048, 4096})
      for (int thr : {448, 512}) {
        const int fpw = (int)(nfr / nwg);
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(e0);
          if (nt) k<1><<<nwg, thr>>>(S, tp, fpw, 448); else k<0><<<nwg, thr>>>(S, tp, fpw, 448);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double bytes = (double)nfr * QP * (5 * 16 + 10 * 16 + 10 * 16);
        printf("nt=%d wgs=%4d threads=%d: %.3f ms  %.0f GB/s actual (r 240 + w 160 B per lane-frame)\n", nt, nwg, thr, best, bytes / best / 1e6);
      }
  return 0;
}
