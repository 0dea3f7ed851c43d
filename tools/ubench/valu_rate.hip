// What one VALU instruction costs a SIMD on gfx950, by kind: every thread runs 16 independent chains of the same instruction
// (so nothing waits for a result), `waves` waves per SIMD, one workgroup per CU; cycles from s_memtime around the loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
using c2 = float __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(1024) rate_kernel(float* out, unsigned long long* cyc, int iters, unsigned long long sconst) {
  const float t = (float)threadIdx.x * 1e-3f;
  float x[16];
  c2 p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = t + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = c2{t + i, t - i};
  const float b = 0.999f, c = 1e-3f;
  const c2 b2 = c2{0.999f, 0.998f}, cc2 = c2{1e-3f, 2e-3f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(cc2));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "s"(sconst), "v"(cc2));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc2));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 4) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 5) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(b2), "v"(cc2));
      REP8(X) REP8(X)
#undef X
    } else if (KIND == 6) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(x[(i + 8) & 15]));
      REP16(X)
#undef X
    } else if (KIND == 7) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 8) {  // the mix of the wave kernel: one packed, one plain, alternating
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5" : "+v"(p[i]), "+v"(x[i]) : "v"(b2), "v"(cc2), "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if (KIND == 9) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(p[i]) : "v"(b2), "v"(cc2));
      REP8(X) REP8(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 4096, grid = 256;
  for (int waves_per_simd : {1, 2, 4}) {
    const int block = 256 * waves_per_simd;
    const float sc = 0.999f;
    unsigned sb;
    memcpy(&sb, &sc, 4);
    const unsigned long long sconst = ((unsigned long long)sb << 32) | sb;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, out, cyc, 64, sconst);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, out, cyc, iters, sconst);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= grid;
    const double n_inst = (double)iters * 16 * waves_per_simd;  // VALU instructions per SIMD
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on this family: report both the tick count and the event time
    printf("%-34s %d waves/SIMD: %8.3f ms  %6.2f ns per instruction per SIMD  (%.0f counter ticks)\n", name, waves_per_simd, ms, ms * 1e6 / n_inst, mean);
  }
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&cyc, 256 * 8);
  run<0>("v_fma_f32", out, cyc);
  run<7>("v_add_f32", out, cyc);
  run<1>("v_pk_fma_f32", out, cyc);
  run<5>("v_pk_fma_f32 op_sel_hi:[1,0,1]", out, cyc);
  run<2>("v_pk_fma_f32 sgpr pair", out, cyc);
  run<9>("v_pk_fma_f32 clamp", out, cyc);
  run<3>("v_pk_add_f32", out, cyc);
  run<4>("v_pk_mul_f32", out, cyc);
  run<6>("v_mov_b32_dpp wave_shr:1", out, cyc);
  run<8>("pk_fma + fma alternating", out, cyc);
  return 0;
}
