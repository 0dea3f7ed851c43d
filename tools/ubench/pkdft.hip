// micro-benchmark + semantic check: radix-21 row passes of the frame engine written with packed fp32 (v_pk_*_f32 on
// (re, im) register pairs) against the plain fp32 form of rfx_core.h, at the Griffin-Lim kernel's occupancy (7-wave
// workgroups, 77.6 KB of LDS: two per CU).   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize pkdft.hip -o pkdft_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../riffusion-hobby_amd/csrc/rfx_core.h"
using namespace rfx;
using c2 = float __attribute__((ext_vector_type(2)));

__device__ __forceinline__ c2 bc(float c) { return c2{c, c}; }
__device__ __forceinline__ c2 pfma(c2 a, c2 b, c2 c) { return __builtin_elementwise_fma(a, b, c); }
// a + i b, a - i b
__device__ __forceinline__ c2 add_i(c2 a, c2 b) { c2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ c2 sub_i(c2 a, c2 b) { c2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ c2 pcmul(c2 a, c2 w) {
  c2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
__device__ __forceinline__ c2 pcmulc(c2 a, c2 w) {
  c2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
template <bool INV>
__device__ __forceinline__ void pdft3(c2& x0, c2& x1, c2& x2) {
  const float q = INV ? -0.86602540378443864676f : 0.86602540378443864676f;
  c2 s = x1 + x2, d = x1 - x2;
  c2 a = pfma(bc(-0.5f), s, x0);
  x0 = x0 + s;
  c2 qd = bc(q) * d;
  x1 = sub_i(a, qd);
  x2 = add_i(a, qd);
}
template <bool INV>
__device__ __forceinline__ void pdft7(c2& x0, c2& x1, c2& x2, c2& x3, c2& x4, c2& x5, c2& x6) {
  constexpr float c1 = 0.62348980185873353053f, c2_ = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
  constexpr float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
  c2 p1 = x1 + x6, m1 = x1 - x6, p2 = x2 + x5, m2 = x2 - x5, p3 = x3 + x4, m3 = x3 - x4;
  c2 a1 = pfma(bc(c3), p3, pfma(bc(c2_), p2, pfma(bc(c1), p1, x0)));
  c2 a2 = pfma(bc(c1), p3, pfma(bc(c3), p2, pfma(bc(c2_), p1, x0)));
  c2 a3 = pfma(bc(c2_), p3, pfma(bc(c1), p2, pfma(bc(c3), p1, x0)));
  c2 b1 = pfma(bc(s3), m3, pfma(bc(s2), m2, bc(s1) * m1));
  c2 b2 = pfma(bc(-s1), m3, pfma(bc(-s3), m2, bc(s2) * m1));
  c2 b3 = pfma(bc(s2), m3, pfma(bc(-s1), m2, bc(s3) * m1));
  x0 = x0 + p1 + p2 + p3;
  if (INV) {
    x1 = add_i(a1, b1); x6 = sub_i(a1, b1); x2 = add_i(a2, b2); x5 = sub_i(a2, b2); x3 = add_i(a3, b3); x4 = sub_i(a3, b3);
  } else {
    x1 = sub_i(a1, b1); x6 = add_i(a1, b1); x2 = sub_i(a2, b2); x5 = add_i(a2, b2); x3 = sub_i(a3, b3); x4 = add_i(a3, b3);
  }
}
template <bool INV>
__device__ __forceinline__ void pdft21(c2 (&x)[21]) {
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) pdft3<INV>(x[(3 * n2) % 21], x[(7 + 3 * n2) % 21], x[(14 + 3 * n2) % 21]);
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
    pdft7<INV>(x[(7 * k1) % 21], x[(7 * k1 + 3) % 21], x[(7 * k1 + 6) % 21], x[(7 * k1 + 9) % 21],
               x[(7 * k1 + 12) % 21], x[(7 * k1 + 15) % 21], x[(7 * k1 + 18) % 21]);
  c2 y[21];
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) y[(7 * k1 + 15 * k2) % 21] = x[(7 * k1 + 3 * k2) % 21];
#pragma unroll
  for (int i = 0; i < 21; ++i) x[i] = y[i];
}

// ---- semantic check: one thread per sample: dft21 fwd/inv, cmul, cmulc against the plain forms
__global__ void check_kernel(const cf* in, const cf* w, cf* out_pk, cf* out_plain) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  cf x[21]; c2 p[21];
  for (int k = 0; k < 21; ++k) { x[k] = in[i * 21 + k]; p[k] = c2{x[k].re, x[k].im}; }
  dft21<false>(x); pdft21<false>(p);
  for (int k = 0; k < 21; ++k) { x[k] = cmul(x[k], w[k]); p[k] = pcmul(p[k], c2{w[k].re, w[k].im}); }
  dft21<true>(x); pdft21<true>(p);
  for (int k = 0; k < 21; ++k) { x[k] = cmulc(x[k], w[k]); p[k] = pcmulc(p[k], c2{w[k].re, w[k].im}); }
  for (int k = 0; k < 21; ++k) { out_plain[i * 21 + k] = x[k]; out_pk[i * 21 + k] = cf{p[k].x, p[k].y}; }
}

// ---- throughput: P2-like (column access, forward + twiddle) then P3'-like (row access, inverse + conj twiddle) passes
constexpr int kLds = kCubeElems * 8;
template <int PK>
__global__ void __launch_bounds__(448, 4) pass_kernel(const cf* tw, cf* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool active = lane < 63;
  const int l = active ? lane : 62, row3 = l / 21, idx = l - 21 * row3, k1 = wave * 3 + row3;
  if (PK) {
    c2* cube = reinterpret_cast<c2*>(smem);
    for (int i = threadIdx.x; i < kCubeElems; i += 448) cube[i] = c2{(float)(i % 13) - 6.f, (float)(i % 7) - 3.f};
    c2 w[21];
    for (int k = 0; k < 21; ++k) w[k] = c2{tw[idx * 21 + k].re, tw[idx * 21 + k].im};
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      c2 x[21];
      if (active) {
#pragma unroll
        for (int a = 0; a < 21; ++a) x[a] = cube[cube_at(k1, a, idx)];
        pdft21<false>(x);
        cube[cube_at(k1, 0, idx)] = x[0];
#pragma unroll
        for (int ka = 1; ka < 21; ++ka) cube[cube_at(k1, ka, idx)] = pcmul(x[ka], w[ka]);
      }
      __syncthreads();
      if (active) {
#pragma unroll
        for (int b = 0; b < 21; ++b) x[b] = cube[cube_at(k1, idx, b)];
        pdft21<true>(x);
        cube[cube_at(k1, idx, 0)] = x[0] * bc(1.f / 441.f);
#pragma unroll
        for (int b = 1; b < 21; ++b) cube[cube_at(k1, idx, b)] = pcmulc(x[b], w[b]) * bc(1.f / 441.f);
      }
      __syncthreads();
    }
    if (active) { c2 v = cube[cube_at(k1, idx, 3)]; out[blockIdx.x * 448 + threadIdx.x] = cf{v.x, v.y}; }
  } else {
    cf* cube = reinterpret_cast<cf*>(smem);
    for (int i = threadIdx.x; i < kCubeElems; i += 448) cube[i] = cf{(float)(i % 13) - 6.f, (float)(i % 7) - 3.f};
    cf w[21];
    for (int k = 0; k < 21; ++k) w[k] = tw[idx * 21 + k];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      cf x[21];
      if (active) {
#pragma unroll
        for (int a = 0; a < 21; ++a) x[a] = cube[cube_at(k1, a, idx)];
        dft21<false>(x);
        cube[cube_at(k1, 0, idx)] = x[0];
#pragma unroll
        for (int ka = 1; ka < 21; ++ka) cube[cube_at(k1, ka, idx)] = cmul(x[ka], w[ka]);
      }
      __syncthreads();
      if (active) {
#pragma unroll
        for (int b = 0; b < 21; ++b) x[b] = cube[cube_at(k1, idx, b)];
        dft21<true>(x);
        const float s = 1.f / 441.f;
        cube[cube_at(k1, idx, 0)] = cf{x[0].re * s, x[0].im * s};
#pragma unroll
        for (int b = 1; b < 21; ++b) { cf v = cmulc(x[b], w[b]); cube[cube_at(k1, idx, b)] = cf{v.re * s, v.im * s}; }
      }
      __syncthreads();
    }
    if (active) out[blockIdx.x * 448 + threadIdx.x] = cube[cube_at(k1, idx, 3)];
  }
}

template <int PK>
float run(const cf* tw, cf* out, int iters, int grid) {
  hipFuncSetAttribute((const void*)pass_kernel<PK>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  pass_kernel<PK><<<grid, 448, kLds>>>(tw, out, 4);
  hipEventRecord(e0);
  pass_kernel<PK><<<grid, 448, kLds>>>(tw, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const int n = 4096;
  std::vector<cf> in(n * 21), w(21), a(n * 21), b(n * 21);
  srand(1);
  for (auto& v : in) v = cf{(float)rand() / RAND_MAX - 0.5f, (float)rand() / RAND_MAX - 0.5f};
  for (int k = 0; k < 21; ++k) w[k] = cf{(float)cos(0.3 * k + 0.1), (float)-sin(0.3 * k + 0.1)};
  cf *din, *dw, *da, *db;
  hipMalloc(&din, n * 21 * 8); hipMalloc(&dw, 21 * 8); hipMalloc(&da, n * 21 * 8); hipMalloc(&db, n * 21 * 8);
  hipMemcpy(din, in.data(), n * 21 * 8, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), 21 * 8, hipMemcpyHostToDevice);
  check_kernel<<<n / 64, 64>>>(din, dw, da, db);
  hipMemcpy(a.data(), da, n * 21 * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 21 * 8, hipMemcpyDeviceToHost);
  double maxd = 0, maxv = 0;
  for (int i = 0; i < n * 21; ++i) {
    maxd = fmax(maxd, fmax(fabs(a[i].re - b[i].re), fabs(a[i].im - b[i].im)));
    maxv = fmax(maxv, fabs(b[i].re));
  }
  printf("semantic check packed vs plain: max |diff| %.3g of max %.3g -> %s\n", maxd, maxv, maxd < 1e-4 * maxv ? "OK" : "FAIL");
  std::vector<cf> tw(441);
  for (int i = 0; i < 21; ++i) for (int j = 0; j < 21; ++j) tw[i * 21 + j] = cf{(float)cos(6.283185307179586 * (i * j % 441) / 441), (float)-sin(6.283185307179586 * (i * j % 441) / 441)};
  cf *dtw, *dout; hipMalloc(&dtw, 441 * 8); hipMalloc(&dout, 2048 * 448 * 8);
  hipMemcpy(dtw, tw.data(), 441 * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep)
    for (int grid : {256, 512}) {
      const int iters = 2000;
      float p = run<0>(dtw, dout, iters, grid), q = run<1>(dtw, dout, iters, grid);
      printf("grid %4d (%d WG/CU) x %d double passes: plain %.3f ms (%.2f us per double pass), packed %.3f ms (%.2f us)  ratio %.3f\n", grid, grid / 256, iters, p,
             p * 1e3 / iters, q, q * 1e3 / iters, p / q);
    }
  return 0;
}
