// micro-benchmark: issue rate of plain vs packed fp32 VALU on gfx950, by waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
using v2f = float __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  v2f av = {a, a}, bv = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(av), "v"(bv));
      }
    } else if (MODE == 2) {  // add
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    } else if (MODE == 3) {  // pk_add
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                     "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(av), "v"(bv));
      }
    } else if (MODE == 4) {  // dependent chain of fma
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
    } else if (MODE == 5) {  // v_sqrt
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                     "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
void run(const char* name, int waves_per_simd, float* out) {
  int iters = 4000;
  int block = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
  int blocks_per_cu = (64 * 4 * waves_per_simd) / block;
  int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, block>>>(out, 10, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<grid, block>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts_per_wave = (double)iters * 64;
  double waves_per_simd_total = waves_per_simd;  // per SIMD
  // cycles per instruction per SIMD assuming 2.4 GHz nominal (report also time)
  double ns_per_inst_simd = ms * 1e6 / (insts_per_wave * waves_per_simd_total);
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.3f ns per wave-instruction per SIMD (= %.2f clk @2.4GHz, %.2f clk @2.0GHz)\n", name, waves_per_simd, ms, ns_per_inst_simd, ns_per_inst_simd * 2.4, ns_per_inst_simd * 2.0);
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 8 * sizeof(float));
  for (int w : {1, 2, 4, 8}) {
    run<0>("fma", w, out); run<1>("pk_fma", w, out); run<2>("add", w, out); run<3>("pk_add", w, out); run<5>("sqrt", w, out);
  }
  run<4>("fma_dep", 1, out); run<4>("fma_dep", 2, out); run<4>("fma_dep", 4, out);
  return 0;
}
