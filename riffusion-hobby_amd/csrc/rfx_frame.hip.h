// rfx_frame.hip.h - device-side frame engine shared by the Griffin-Lim and the forward STFT kernels.
//
// One workgroup = 7 waves (448 threads, lane 63 of each wave idle) owns one frame at a time and
// keeps the 21 x 441 complex slot matrix ("cube", 74 088 B) in LDS.  Thread roles by pass:
//   P1 / P1' : n' = wave*63 + lane             (a = n'/21 = wave*3 + lane/21, b = lane%21)
//   P2 / P2' : (k1, b)  with k1 = wave*3 + lane/21, b  = lane%21
//   P3 / P3' : (k1, ka) with k1 = wave*3 + lane/21, ka = lane%21
// so a thread keeps the same (row-triple, idx) identity throughout; the P2<->P3 exchange stays
// inside a wave's three rows and only the P1<->P2 exchange crosses waves (one barrier each way).
#pragma once
#include <hip/hip_runtime.h>
#include "rfx_core.h"

namespace rfx {

using v4f = float __attribute__((ext_vector_type(4)));
using v2f = float __attribute__((ext_vector_type(2)));

struct ThreadId {
  int wave, lane, row3, idx;  // row3 = lane/21 (0..2), idx = lane%21
  int npr;                    // P1 index n' (== P3 index q = k1*21+ka)
  int k1;                     // row owned in P2/P3
  bool active;
};

__device__ __forceinline__ ThreadId thread_id() {
  ThreadId t;
  t.wave = threadIdx.x >> 6;
  t.lane = threadIdx.x & 63;
  t.active = t.lane < 63;
  const int l = t.active ? t.lane : 62;  // idle lane shadows lane 62 (loads only, never stores)
  t.row3 = l / 21;
  t.idx = l - 21 * t.row3;
  t.k1 = t.wave * 3 + t.row3;
  t.npr = t.wave * 63 + l;
  return t;
}

// per-thread constants that live in registers for the whole run of frames
struct ThreadConst {
  cf tw1[21];     // g(n')^k1
  cf tw2[21];     // w441^(idx*i)
  float win[10];  // hann[441*j + n']
};

__device__ __forceinline__ void load_thread_const(ThreadConst& c, const ThreadId& t, const cf* __restrict__ tw1,
                                                  const cf* __restrict__ tw2, const float* __restrict__ win) {
#pragma unroll
  for (int k = 0; k < 21; ++k) c.tw1[k] = tw1[k * kHop + t.npr];
#pragma unroll
  for (int k = 0; k < 21; ++k) c.tw2[k] = tw2[k * 21 + t.idx];
#pragma unroll
  for (int j = 0; j < 10; ++j) c.win[j] = win[j * kHop + t.npr];
}

// wave-private LDS hand-off (P2 <-> P3 touch only the wave's own three rows).  DS operations of one
// wave execute in issue order, so only the compiler has to be kept from reordering them.
#ifndef RFX_WAVE_SYNC_IS_BARRIER
#define RFX_WAVE_SYNC_IS_BARRIER 0
#endif
__device__ __forceinline__ void wave_sync() {
#if RFX_WAVE_SYNC_IS_BARRIER
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// forward transform of one frame: u[10] (windowed samples of thread n') -> R[21] (slots of thread q)
__device__ __forceinline__ void frame_forward(const float (&u)[10], cf (&R)[21], cf* cube, const ThreadId& t,
                                              const ThreadConst& c) {
  cf v[21];
  p1_forward(u, v);
  if (t.active) p1_store(v, c.tw1, cube, t.npr);
  __syncthreads();
  if (t.active) p2_forward(cube, c.tw2, t.k1, t.idx);
  wave_sync();
  p3_forward(cube, R, t.k1, t.idx);
}

// inverse transform of one frame: Z[21] (slots of thread q) -> y[10] (un-normalised hops of thread n')
__device__ __forceinline__ void frame_inverse(cf (&Z)[21], float (&y)[10], cf* cube, const ThreadId& t,
                                              const ThreadConst& c) {
  if (t.active) p3_inverse(cube, Z, c.tw2, t.k1, t.idx);
  wave_sync();
  if (t.active) p2_inverse(cube, t.k1, t.idx);
  __syncthreads();
  cf V[21];
  p1_load(cube, c.tw1, V, t.npr);
  p1_inverse(V, y);
}

}  // namespace rfx
