// rfx_frame.hip.h - device-side frame engine shared by the Griffin-Lim and the forward STFT kernels.
//
// One workgroup = kWaves waves (rfx_core.h; 7 x 63 or 5 x 63 + 3 x 42 active lanes) owns one frame at a time and
// keeps the 21 x 441 complex slot matrix ("cube", 74 088 B) in LDS, followed by the 21 x 21 table
// of w441 twiddles (3 528 B): 77 616 B per workgroup, so two workgroups share a CU's 160 KiB and
// one's LDS-exchange / HBM phases overlap the other's butterflies.  Thread roles by pass:
//   P1 / P1' : n' = wave*63 + lane             (a = n'/21 = wave*3 + lane/21, b = lane%21)
//   P2 / P2' : (k1, b)  with k1 = wave*3 + lane/21, b  = lane%21
//   P3 / P3' : (k1, ka) with k1 = wave*3 + lane/21, ka = lane%21
// so a thread keeps the same (row-triple, idx) identity throughout; the P2<->P3 exchange stays
// inside a wave's three rows and only the P1<->P2 exchange crosses waves (one barrier each way).
// Register budget is 128 VGPRs (4 waves per SIMD): the g(n')^k1 twiddles stream from their
// L2-resident table at the point of use, the w441 twiddles from LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "rfx_core.h"

namespace rfx {

using v4f = float __attribute__((ext_vector_type(4)));
using v2f = float __attribute__((ext_vector_type(2)));
using v4u = unsigned __attribute__((ext_vector_type(4)));
using v2u = unsigned __attribute__((ext_vector_type(2)));

// ---- buffer-descriptor addressing: every global access is  SRD (SGPRs, wave-uniform)  +  one 32-bit
// per-lane byte offset  +  a scalar byte offset, so no 64-bit per-lane address ever occupies VGPRs.
// Descriptors are built only from kernel arguments and blockIdx, which keeps them provably uniform.
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int kAuxNT = 2;  // gfx950 cache-policy bit "nt": streamed once, do not keep in L2
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes), 0x00020000);
}
template <int AUX = 0>
__device__ __forceinline__ float ld1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ v2f ld2(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
}
// integer tables.  NOT `bit_cast<unsigned>(ld2(...).y)`: hipcc (ROCm 7.2) narrows a 64-bit buffer load whose float halves are both
// bit-cast back to integers to a 32-bit load and hands out the low half twice (seen in the IR and the ISA, round 5).
template <int AUX = 0>
__device__ __forceinline__ unsigned ld1u(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ v2u ld2u(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ v4f ld4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void st1(float v, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void st2(v2f v, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void st4(v4f v, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, voff, soff, AUX);
}

constexpr int kCubeBytes = kCubeElems * (int)sizeof(cf);    // 74 088 at the dense row stride
// w441 twiddles in LDS, one row per idx = lane % 21: row[j] = w441^(idx * (j + 1)), j = 0..19, padded to 22
// entries = 176 B so that a thread fetches its 20 twiddles with ten 16-byte reads and the 16 lanes of a
// ds_read_b128 group (row stride 44 dwords) spread over all 64 banks.  A separate static array, not part of
// the dynamic cube allocation: the compiler then knows that cube stores never alias twiddle reads (with one
// shared base pointer it serialised every twiddle read behind the preceding cube store: 40 LDS round trips
// per frame on the critical path).
constexpr int kTw2Row = 22;
constexpr int kTw2Bytes = 21 * kTw2Row * (int)sizeof(cf);   // 3 696 (static)
constexpr int kFrameDynLdsBytes = kCubeBytes;               // dynamic part: the cube
constexpr int kFrameLdsBytes = kCubeBytes + kTw2Bytes;      // 77 784 per workgroup in total

struct ThreadId {
  int idx;   // lane % 21
  int npr;   // P1 index n' (== P3 index q = k1*21+ka)
  int k1;    // row owned in P2/P3
  int pad;   // 0..6: this (idle) lane zeroes padding position 64*pad+63 of the HBM slot groups; -1 otherwise
  bool active;
};

__device__ __forceinline__ ThreadId thread_id() {
  ThreadId t;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  static_assert(kWaves == 7 || kWaves == 8, "7 waves x 3 rows, or 5 x 3 + 3 x 2 rows");
  // kWaves == 8: waves 0..4 own rows 3w..3w+2, waves 5..7 rows 15+2(w-5) and the next one
  const int rows = (kWaves == 7 || wave < 5) ? 3 : 2;
  const int row0 = (kWaves == 7 || wave < 5) ? 3 * wave : 15 + 2 * (wave - 5);
  t.active = lane < 21 * rows;
  const int l = t.active ? lane : 21 * rows - 1;  // idle lanes shadow the wave's last active lane (loads only, never stores)
  const int row3 = l / 21;
  t.idx = l - 21 * row3;
  t.k1 = row0 + row3;
  t.npr = t.k1 * 21 + t.idx;
  if (kWaves == 7) t.pad = lane == 63 ? wave : -1;
  else t.pad = (wave == 7 && lane >= 42 && lane < 49) ? lane - 42 : -1;
  return t;
}

struct FrameCtx {
  cf* cube;          // LDS
  const v4f* tw2row; // LDS row of this thread's w441^(idx*k), k = 1..20, as ten 16-byte pairs
  rsrc_t tw1;        // global tw1[21][441]
  unsigned npr8;     // n' * sizeof(cf)
};

// copies the w441 table into LDS; the caller must barrier before the first transform
__device__ __forceinline__ FrameCtx frame_ctx(char* smem, const ThreadId& t, const cf* __restrict__ tw1,
                                              const cf* __restrict__ tw2) {
  __shared__ __attribute__((aligned(16))) cf tw2_lds[21 * kTw2Row];
  FrameCtx f;
  f.cube = reinterpret_cast<cf*>(smem);
  if (threadIdx.x < 420) {  // tw2 is symmetric: tw2[k][idx] == tw2[idx][k]
    const int idx = threadIdx.x / 20, j = threadIdx.x - idx * 20;
    tw2_lds[idx * kTw2Row + j] = tw2[idx * 21 + j + 1];
  }
  f.tw2row = reinterpret_cast<const v4f*>(tw2_lds + t.idx * kTw2Row);
  f.tw1 = make_rsrc(tw1, 21 * kHop * sizeof(cf));
  f.npr8 = (unsigned)t.npr * 8u;
  return f;
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

// the thread's 20 w441 twiddles (k = 1..20) come in as ten ds_read_b128, in two batches: k = 1..10 before
// the butterflies (in flight underneath them), k = 11..20 once the first outputs have left their registers
struct Tw2 {
  cf w[21];
};
template <int HALF>
__device__ __forceinline__ void load_tw2_half(Tw2& tw, const FrameCtx& f) {
#pragma unroll
  for (int i = HALF * 5; i < HALF * 5 + 5; ++i) {
    const v4f p = f.tw2row[i];
    tw.w[2 * i + 1] = cf{p.x, p.y};
    tw.w[2 * i + 2] = cf{p.z, p.w};
  }
}
struct Tw2Stage {
  Tw2& tw;
  const FrameCtx& f;
  __device__ __forceinline__ void operator()(int s) const {
    if (s == 0) load_tw2_half<0>(tw, f); else load_tw2_half<1>(tw, f);
    RFX_SCHED_FENCE();
  }
};

// the 20 non-trivial g(n')^k1 twiddles of this thread, fetched from the L2-resident table in one
// burst so that their latency overlaps the butterflies / the barrier that precede their use
struct Tw1 {
  cf w[21];
};
// SHORT (round 5, forward kernel): only g^1 .. g^10 and g^20 are fetched (g^20 parked in w[11]); tw1_get derives the rows
// 11..19 at the point of use as g^(20-k) = g^20 conj(g^k), like rfx_fam_core.h::fam_g_pow: 88 instead of 160 bytes per thread
// and frame from L2 and 18 registers less in flight, for nine more packed complex products in P1
template <bool SHORT = false>
__device__ __forceinline__ void load_tw1(Tw1& tw, const FrameCtx& f) {
#pragma unroll
  for (int k = 1; k < (SHORT ? 12 : 21); ++k) {
#if defined(RFX_FWD_ABL) && RFX_FWD_ABL == 8  // timing ablation of the forward kernel (wrong results): no twiddle fetches
    tw.w[k] = cf{1.f, (float)k};
    continue;
#endif
    v2f w = ld2(f.tw1, f.npr8, (unsigned)(SHORT && k == 11 ? 20 : k) * (kHop * 8u));
    tw.w[k] = cf{w.x, w.y};
  }
}
template <bool SHORT>
__device__ __forceinline__ cf tw1_get(const Tw1& tw, int k) {
  if (!SHORT || k <= 10) return tw.w[k];
  return k == 20 ? tw.w[11] : cmulc(tw.w[11], tw.w[20 - k]);
}

// forward transform of one frame: u[10] (windowed samples of thread n') -> R[21] (slots of thread q).
// `after_barrier()` runs right after the cross-wave barrier: the caller uses it to put HBM loads in
// flight underneath P2/P3.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
  __device__ __forceinline__ void operator()(int) const {}
};
template <bool TWSHORT = false, class Hook, class Pre = NoHook, class Mid = NoHook>
__device__ __forceinline__ void frame_forward_tw(const float (&u)[10], cf (&R)[21], const FrameCtx& f, const ThreadId& t,
                                                 const Tw1& tw, Hook after_barrier, Pre before_barrier = Pre(),
                                                 Mid before_p3 = Mid()) {
  if (t.active) p1_forward_store(u, [&tw](int k) { return tw1_get<TWSHORT>(tw, k); }, f.cube, t.npr);
  before_barrier();
  __syncthreads();
#ifndef RFX_NO_PRIO
  // the long barrier-free stretch (P2 .. P2', ~1450 instructions per wave) runs at raised priority, the short one
  // between the synthesis barrier and the next analysis barrier at normal priority: measured 1-2 % (26.9 -> 26.2-26.5 ms
  // per 64 tiles x 32 iterations); the opposite assignment and __launch_bounds__ min-waves 3 were slower
  __builtin_amdgcn_s_setprio(2);
#endif
  after_barrier();
  if (t.active) {
    Tw2 w2;
    p2_forward(f.cube, [&w2](int k) { return w2.w[k]; }, t.k1, t.idx, Tw2Stage{w2, f});
  }
  before_p3();
  wave_sync();
  p3_forward(f.cube, R, t.k1, t.idx);
}
template <class Hook, class Pre = NoHook>
__device__ __forceinline__ void frame_forward(const float (&u)[10], cf (&R)[21], const FrameCtx& f, const ThreadId& t,
                                              Hook after_barrier, Pre before_barrier = Pre()) {
  Tw1 tw;
  load_tw1(tw, f);
  frame_forward_tw(u, R, f, t, tw, after_barrier, before_barrier);
}

// inverse transform of one frame: Z[21] (slots of thread q) -> y[10] (un-normalised hops of thread n')
// `tw` is (re)loaded here, in flight across the barrier, and handed back to the caller: P1' and the next
// frame's P1 use the same g(n')^k1 values (conjugated), so a Griffin-Lim iteration fetches them once per frame.
template <class Pre = NoHook, class Post = NoHook, class Probe = NoHook>
__device__ __forceinline__ void frame_inverse_tw(cf (&Z)[21], float (&y)[10], const FrameCtx& f, const ThreadId& t, Tw1& tw,
                                                 Pre before_barrier = Pre(), Post after_barrier = Post(), Probe probe = Probe()) {
  if (t.active) {
    Tw2 w2;
    p3_inverse(f.cube, Z, [&w2](int k) { return w2.w[k]; }, t.k1, t.idx, Tw2Stage{w2, f});
  }
  probe(0);  // timing builds only
  wave_sync();
  if (t.active) p2_inverse(f.cube, t.k1, t.idx);
  probe(1);
  load_tw1(tw, f);
  before_barrier();
  __syncthreads();
#ifndef RFX_NO_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  after_barrier();
  p1_load_inverse(f.cube, [&tw](int k) { return tw.w[k]; }, y, t.npr);
}
template <class Pre = NoHook, class Post = NoHook>
__device__ __forceinline__ void frame_inverse(cf (&Z)[21], float (&y)[10], const FrameCtx& f, const ThreadId& t,
                                              Pre before_barrier = Pre(), Post after_barrier = Post()) {
  Tw1 tw;
  frame_inverse_tw(Z, y, f, t, tw, before_barrier, after_barrier);
}

}  // namespace rfx
