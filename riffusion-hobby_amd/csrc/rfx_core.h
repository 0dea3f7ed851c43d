// rfx_core.h - per-thread arithmetic of the 17640-point framed transform, written once for both the
// gfx950 kernels (hipcc) and the host-side phase emulator used by the CPU tests (g++).
//
// Geometry (riffusion/spectrogram_params.py:24-27,62-81 at 44.1 kHz): n_fft N = 17640 = 40*441,
// win_length = 4410 = 10*441 centred in the frame (zero padding 6615 = 15*441 on both sides),
// hop = 441 = 21*21.  A frame transform (what torch.stft / torch.istft do per frame inside
// torchaudio's Spectrogram / GriffinLim, spectrogram_converter.py:47-73) is decomposed as
//
//   sample index  n = 441*J + n'      J = 15..24 (only the 10 windowed hops are non-zero), n' = 0..440
//   bin index     k = k1 + 40*k'      k1 = 0..39, k' = 0..440
//   X[k1+40k'] = sum_{n'} w441^{n'k'} * g(n')^{k1} * sum_{j=0..9} u[441j+n'] * w40^{j*k1}
//                with g(n') = exp(-2*pi*i*(n'+6615)/17640)
//
// i.e. P1: a 40-point DFT of 10 real inputs per n' (only k1 = 0..20 kept: rows 21..39 are complex
// conjugates because the input is real), a twiddle, then 21 complex 441-point FFTs (rows), each done
// as 21x21 (P2 over a, twiddle, P3 over b; n' = 21a+b, k' = ka+21kb) with Good-Thomas 3x7 radix-21
// butterflies.  The 21*441 = 9261 row outputs ("slots") cover the 8821 one-sided bins: slot k<=8820
// holds X[k]; slot k>8820 holds conj(X[17640-k]) (440 bins with k mod 40 in {0,20} appear twice).
// Everything between the forward and the inverse transform in Griffin-Lim is per-bin and commutes
// with conjugation, so slots are treated as independent bins and no mirror exchange is ever needed.
//
// The inverse retraces the same three passes backwards (DIF forward / DIT inverse: no reordering
// between them), ending in a pruned 40-point inverse that yields the 10 windowed hops.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RFX_HD __host__ __device__ __forceinline__
#else
#define RFX_HD inline __attribute__((always_inline))
#endif

namespace rfx {

struct cf {
  float re, im;
};

constexpr int kHop = 441;        // hop_length; also the row length of the 21 x 441 slot matrix
constexpr int kRows = 21;        // kept k1 residues 0..20
constexpr int kSlots = 9261;     // 21 * 441
constexpr int kNfft = 17640;
constexpr int kWin = 4410;
constexpr int kBins = 8821;      // n_fft/2 + 1
constexpr int kQPad = 448;          // 441 owner threads padded to 7 waves x 64 lanes in HBM
constexpr int kFrameStride = 9408;  // 21 * 448 positions per frame in HBM: every wave-load is whole 128-B lines
constexpr int kWinHops = 10;     // win_length / hop
constexpr int kHalfHops = 5;     // frame t is centred on sample 441*t: it spans hop blocks t-5 .. t+4
#ifndef RFX_ROW_STRIDE
#define RFX_ROW_STRIDE 441
#endif
constexpr int kRowStride = RFX_ROW_STRIDE;          // LDS elements between cube rows (>= 441)
constexpr int kCubeElems = 20 * kRowStride + kHop;
// Workgroup shape of the frame engine.  The 441 thread roles (P1: n'; P2/P3: (row k1, idx)) are dealt to waves by whole
// rows, so that the P2 <-> P3 exchange stays inside a wave: 7 waves x 3 rows fill 63 of 64 lanes.  A CU holds two
// workgroups (LDS) = 14 waves on four SIMDs = 4/4/3/3.  RFX_WAVES=8 (five waves with 3 rows, three with 2) puts 4 waves
// on EVERY SIMD at the same per-wave instruction stream; measured SLOWER (27.4 vs 26.2 ms per 64 tiles x 32 iterations):
// the kernel is bound by aggregate instruction issue, not by the imbalance.
#ifndef RFX_WAVES
#define RFX_WAVES 7
#endif
constexpr int kWaves = RFX_WAVES;
constexpr int kThreads = 64 * kWaves;

// Packed fp32 butterflies on the device (round 4): a translation unit opts in with `#define RFX_PK 1` ahead of this header
// (rfx_gl.hip, rfx_stft.hip; -DRFX_NO_PK builds the plain forms for A/B runs).  The generic engine stays plain: the packed
// complex product's aligned output pair costs it registers it does not have (0 -> 180..800 B of scratch, measured in the ISA).
#if defined(RFX_NO_PK)
#undef RFX_PK
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(RFX_PK)
// ---- packed fp32: a complex value is an aligned VGPR pair (re, im) and a butterfly step is one v_pk_*_f32 - half the
// issue slots of the plain form for the same ALU cycles, so a wave that is alone in being ready on its SIMD issues twice
// the work per slot (tools/ubench/valu.hip: one wave per SIMD 2.59 ns per v_pk_fma against 2.40 ns per v_fma).  The
// compiler matches broadcasts and whole-vector negation from vector code but not the VOP3P source selects that swap or
// negate halves (it builds the swapped pair with v_mov): those forms are inline asm.  Semantics checked against the plain
// forms on the device by tools/ubench/pkdft.hip.
using c2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c2 pk_of(cf a) { return c2{a.re, a.im}; }
__device__ __forceinline__ cf pk_to(c2 a) { return cf{a.x, a.y}; }
__device__ __forceinline__ c2 pk_bc(float c) { return c2{c, c}; }
__device__ __forceinline__ c2 pk_fma(c2 a, c2 b, c2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ c2 pk_add_i(c2 a, c2 b) { c2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }  // a + i b
__device__ __forceinline__ c2 pk_sub_i(c2 a, c2 b) { c2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }  // a - i b
#define RFX_PK_ADD(name, mods) \
  __device__ __forceinline__ c2 name(c2 a, c2 b) { c2 r; asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b)); return r; }
RFX_PK_ADD(pk_add_cj, "neg_hi:[0,1]")                                                  // a + conj(b)
RFX_PK_ADD(pk_sub_cj, "neg_lo:[0,1]")                                                  // a - conj(b)
RFX_PK_ADD(pk_cj_sub, "neg_lo:[0,1] neg_hi:[1,0]")                                     // conj(a - b)
RFX_PK_ADD(pk_cj_add_i, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]")      // conj(a + i b)
RFX_PK_ADD(pk_add_sw, "op_sel:[0,1] op_sel_hi:[1,0]")                                  // (a.re + b.im, a.im + b.re)
RFX_PK_ADD(pk_sub_sw, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")        // (a.re - b.im, a.im - b.re)
// complex products as ONE two-instruction statement: hipcc pads an asm statement whose input the preceding VALU has just
// written with an s_nop (it cannot see that no hazard exists), so the dependent pair must not be two statements
__device__ __forceinline__ c2 pk_cmul(c2 a, c2 w) {  // a * w = a.re * (w.re, w.im) + a.im * (-w.im, w.re)
  c2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=&v"(r) : "v"(a), "v"(w));
  return r;
}
__device__ __forceinline__ c2 pk_cmulc(c2 a, c2 w) {  // a * conj(w) = a.re * (w.re, -w.im) + a.im * (w.im, w.re)
  c2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
      : "=&v"(r) : "v"(a), "v"(w));
  return r;
}
// a -+ i q d with the real constant q in the low half of an SGPR pair (both result halves read it)
__device__ __forceinline__ c2 pk_fma_sub_i(c2 a, float q, c2 d) {  // a - i q d = (a.re + q d.im, a.im - q d.re)
  c2 r;
  const unsigned long long qq = __builtin_bit_cast(unsigned, q);
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(d), "s"(qq), "v"(a));
  return r;
}
__device__ __forceinline__ c2 pk_fma_add_i(c2 a, float q, c2 d) {  // a + i q d = (a.re - q d.im, a.im + q d.re)
  c2 r;
  const unsigned long long qq = __builtin_bit_cast(unsigned, q);
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(d), "s"(qq), "v"(a));
  return r;
}
__device__ __forceinline__ cf cmul(cf a, cf b) { return pk_to(pk_cmul(pk_of(a), pk_of(b))); }
__device__ __forceinline__ cf cmulc(cf a, cf b) { return pk_to(pk_cmulc(pk_of(a), pk_of(b))); }
#else
RFX_HD cf cmul(cf a, cf b) { return cf{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// a * conj(b)
RFX_HD cf cmulc(cf a, cf b) { return cf{a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }
#endif

// ------------------------------------------------------------------------------------------------
// Slot <-> HBM position.  A P3 thread q = k1*21 + ka (0..440) owns kb = 0..20; in memory it sits
// at qp = (q/63)*64 + q%63 = its threadIdx (7 waves x 63 active lanes, lane 63 is padding), so that
// one wave-wide 16-B-per-lane access covers exactly eight whole 128-B lines.  Complex arrays
// (tprev, injected angles) hold two consecutive kb per 16 bytes, float arrays (|S|) four.
// ------------------------------------------------------------------------------------------------
RFX_HD int slot_qp(int q) { return q + q / 63; }
RFX_HD int slot_pos_c(int q, int kb) {
  return kb < 20 ? ((kb >> 1) * kQPad + slot_qp(q)) * 2 + (kb & 1) : 20 * kQPad + slot_qp(q);
}
RFX_HD int slot_pos_f(int q, int kb) {
  return kb < 20 ? ((kb >> 2) * kQPad + slot_qp(q)) * 4 + (kb & 3) : 20 * kQPad + slot_qp(q);
}
// inverse maps: position -> (q, kb); returns false for padding positions
RFX_HD bool pos_c_to_slot(int p, int& q, int& kb) {
  int qp;
  if (p < 20 * kQPad) {
    const int g = p / (2 * kQPad), rem = p - g * 2 * kQPad;
    qp = rem >> 1;
    kb = 2 * g + (rem & 1);
  } else {
    qp = p - 20 * kQPad;
    kb = 20;
  }
  q = (qp >> 6) * 63 + (qp & 63);
  return (qp & 63) != 63;
}
RFX_HD bool pos_f_to_slot(int p, int& q, int& kb) {
  int qp;
  if (p < 20 * kQPad) {
    const int g = p / (4 * kQPad), rem = p - g * 4 * kQPad;
    qp = rem >> 2;
    kb = 4 * g + (rem & 3);
  } else {
    qp = p - 20 * kQPad;
    kb = 20;
  }
  q = (qp >> 6) * 63 + (qp & 63);
  return (qp & 63) != 63;
}
// bin held by slot (k1, ka, kb); *conj_out set when the slot holds the conjugate of that bin
RFX_HD int slot_bin(int k1, int ka, int kb, bool* conj_out) {
  int k = k1 + 40 * (ka + 21 * kb);
  bool c = k > kNfft / 2;
  if (conj_out) *conj_out = c;
  return c ? kNfft - k : k;
}

// ------------------------------------------------------------------------------------------------
// Radix-3 / radix-7 / radix-21 butterflies.  INV selects exp(+i...) kernels.
// ------------------------------------------------------------------------------------------------
template <bool INV>
RFX_HD void dft3(cf& x0, cf& x1, cf& x2) {
  const float q = INV ? -0.86602540378443864676f : 0.86602540378443864676f;
  cf s{x1.re + x2.re, x1.im + x2.im};
  cf d{x1.re - x2.re, x1.im - x2.im};
  float ar = fmaf(-0.5f, s.re, x0.re), ai = fmaf(-0.5f, s.im, x0.im);
  x0 = cf{x0.re + s.re, x0.im + s.im};
  x1 = cf{fmaf(q, d.im, ar), fmaf(-q, d.re, ai)};
  x2 = cf{fmaf(-q, d.im, ar), fmaf(q, d.re, ai)};
}

template <bool INV>
RFX_HD void dft7(cf& x0, cf& x1, cf& x2, cf& x3, cf& x4, cf& x5, cf& x6) {
  constexpr float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
  constexpr float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
  cf p1{x1.re + x6.re, x1.im + x6.im}, m1{x1.re - x6.re, x1.im - x6.im};
  cf p2{x2.re + x5.re, x2.im + x5.im}, m2{x2.re - x5.re, x2.im - x5.im};
  cf p3{x3.re + x4.re, x3.im + x4.im}, m3{x3.re - x4.re, x3.im - x4.im};
  // a_k = x0 + sum_j cos(2 pi j k / 7) p_j ;  b_k = sum_j sin(2 pi j k / 7) m_j
  cf a1{fmaf(c3, p3.re, fmaf(c2, p2.re, fmaf(c1, p1.re, x0.re))), fmaf(c3, p3.im, fmaf(c2, p2.im, fmaf(c1, p1.im, x0.im)))};
  cf a2{fmaf(c1, p3.re, fmaf(c3, p2.re, fmaf(c2, p1.re, x0.re))), fmaf(c1, p3.im, fmaf(c3, p2.im, fmaf(c2, p1.im, x0.im)))};
  cf a3{fmaf(c2, p3.re, fmaf(c1, p2.re, fmaf(c3, p1.re, x0.re))), fmaf(c2, p3.im, fmaf(c1, p2.im, fmaf(c3, p1.im, x0.im)))};
  cf b1{fmaf(s3, m3.re, fmaf(s2, m2.re, s1 * m1.re)), fmaf(s3, m3.im, fmaf(s2, m2.im, s1 * m1.im))};
  cf b2{fmaf(-s1, m3.re, fmaf(-s3, m2.re, s2 * m1.re)), fmaf(-s1, m3.im, fmaf(-s3, m2.im, s2 * m1.im))};
  cf b3{fmaf(s2, m3.re, fmaf(-s1, m2.re, s3 * m1.re)), fmaf(s2, m3.im, fmaf(-s1, m2.im, s3 * m1.im))};
  x0 = cf{x0.re + p1.re + p2.re + p3.re, x0.im + p1.im + p2.im + p3.im};
  // forward: X_k = a_k - i b_k, X_{7-k} = a_k + i b_k ; inverse swaps them
  cf lo1{a1.re + b1.im, a1.im - b1.re}, hi1{a1.re - b1.im, a1.im + b1.re};
  cf lo2{a2.re + b2.im, a2.im - b2.re}, hi2{a2.re - b2.im, a2.im + b2.re};
  cf lo3{a3.re + b3.im, a3.im - b3.re}, hi3{a3.re - b3.im, a3.im + b3.re};
  if (INV) {
    x1 = hi1; x6 = lo1; x2 = hi2; x5 = lo2; x3 = hi3; x4 = lo3;
  } else {
    x1 = lo1; x6 = hi1; x2 = lo2; x5 = hi2; x3 = lo3; x4 = hi3;
  }
}

// 21-point DFT, natural order in and out, prime-factor (Good-Thomas) 3 x 7: no internal twiddles.
//   input  n = (7*n1 + 3*n2) mod 21,   output k = (7*k1 + 15*k2) mod 21
#if defined(__HIP_DEVICE_COMPILE__) && defined(RFX_PK)
template <bool INV>
__device__ __forceinline__ void pk_dft3(c2& x0, c2& x1, c2& x2) {
  const float q = INV ? -0.86602540378443864676f : 0.86602540378443864676f;
  const c2 s = x1 + x2, d = x1 - x2;
  const c2 a = pk_fma(pk_bc(-0.5f), s, x0);
  x0 = x0 + s;
  x1 = pk_fma_sub_i(a, q, d);
  x2 = pk_fma_add_i(a, q, d);
}
template <bool INV>
__device__ __forceinline__ void pk_dft7(c2& x0, c2& x1, c2& x2, c2& x3, c2& x4, c2& x5, c2& x6) {
  constexpr float c1 = 0.62348980185873353053f, c2_ = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
  constexpr float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
  const c2 p1 = x1 + x6, m1 = x1 - x6, p2 = x2 + x5, m2 = x2 - x5, p3 = x3 + x4, m3 = x3 - x4;
  const c2 a1 = pk_fma(pk_bc(c3), p3, pk_fma(pk_bc(c2_), p2, pk_fma(pk_bc(c1), p1, x0)));
  const c2 a2 = pk_fma(pk_bc(c1), p3, pk_fma(pk_bc(c3), p2, pk_fma(pk_bc(c2_), p1, x0)));
  const c2 a3 = pk_fma(pk_bc(c2_), p3, pk_fma(pk_bc(c1), p2, pk_fma(pk_bc(c3), p1, x0)));
  const c2 b1 = pk_fma(pk_bc(s3), m3, pk_fma(pk_bc(s2), m2, pk_bc(s1) * m1));
  const c2 b2 = pk_fma(pk_bc(-s1), m3, pk_fma(pk_bc(-s3), m2, pk_bc(s2) * m1));
  const c2 b3 = pk_fma(pk_bc(s2), m3, pk_fma(pk_bc(-s1), m2, pk_bc(s3) * m1));
  x0 = x0 + p1 + p2 + p3;
  if (INV) {
    x1 = pk_add_i(a1, b1); x6 = pk_sub_i(a1, b1); x2 = pk_add_i(a2, b2); x5 = pk_sub_i(a2, b2); x3 = pk_add_i(a3, b3); x4 = pk_sub_i(a3, b3);
  } else {
    x1 = pk_sub_i(a1, b1); x6 = pk_add_i(a1, b1); x2 = pk_sub_i(a2, b2); x5 = pk_add_i(a2, b2); x3 = pk_sub_i(a3, b3); x4 = pk_add_i(a3, b3);
  }
}
#endif

template <bool INV>
RFX_HD void dft21(cf (&x)[21]) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(RFX_PK)
  c2 p[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) p[i] = c2{x[i].re, x[i].im};
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) pk_dft3<INV>(p[(3 * n2) % 21], p[(7 + 3 * n2) % 21], p[(14 + 3 * n2) % 21]);
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
    pk_dft7<INV>(p[(7 * k1) % 21], p[(7 * k1 + 3) % 21], p[(7 * k1 + 6) % 21], p[(7 * k1 + 9) % 21],
                 p[(7 * k1 + 12) % 21], p[(7 * k1 + 15) % 21], p[(7 * k1 + 18) % 21]);
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) x[(7 * k1 + 15 * k2) % 21] = cf{p[(7 * k1 + 3 * k2) % 21].x, p[(7 * k1 + 3 * k2) % 21].y};
  return;
#endif
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) dft3<INV>(x[(3 * n2) % 21], x[(7 + 3 * n2) % 21], x[(14 + 3 * n2) % 21]);
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
    dft7<INV>(x[(7 * k1) % 21], x[(7 * k1 + 3) % 21], x[(7 * k1 + 6) % 21], x[(7 * k1 + 9) % 21],
              x[(7 * k1 + 12) % 21], x[(7 * k1 + 15) % 21], x[(7 * k1 + 18) % 21]);
  cf y[21];
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) y[(7 * k1 + 15 * k2) % 21] = x[(7 * k1 + 3 * k2) % 21];
#pragma unroll
  for (int i = 0; i < 21; ++i) x[i] = y[i];
}

// ------------------------------------------------------------------------------------------------
// 20th / 40th roots of unity as compile-time tables (cos, sin of 2*pi*i/20 and 2*pi*i/40)
// ------------------------------------------------------------------------------------------------
#define RFX_C20_TABLE                                                                                       \
  {1.0f, 0.95105651629515357212f, 0.80901699437494742410f, 0.58778525229247312917f, 0.30901699437494742410f, \
   0.0f, -0.30901699437494742410f, -0.58778525229247312917f, -0.80901699437494742410f,                      \
   -0.95105651629515357212f, -1.0f, -0.95105651629515357212f, -0.80901699437494742410f,                     \
   -0.58778525229247312917f, -0.30901699437494742410f, 0.0f, 0.30901699437494742410f,                       \
   0.58778525229247312917f, 0.80901699437494742410f, 0.95105651629515357212f}
#define RFX_S20_TABLE                                                                                       \
  {0.0f, 0.30901699437494742410f, 0.58778525229247312917f, 0.80901699437494742410f, 0.95105651629515357212f, \
   1.0f, 0.95105651629515357212f, 0.80901699437494742410f, 0.58778525229247312917f, 0.30901699437494742410f, \
   0.0f, -0.30901699437494742410f, -0.58778525229247312917f, -0.80901699437494742410f,                      \
   -0.95105651629515357212f, -1.0f, -0.95105651629515357212f, -0.80901699437494742410f,                     \
   -0.58778525229247312917f, -0.30901699437494742410f}
// cos / sin of 2*pi*k/40 for k = 0..10
#define RFX_C40_TABLE                                                                                       \
  {1.0f, 0.98768834059513772619f, 0.95105651629515357212f, 0.89100652418836786236f, 0.80901699437494742410f, \
   0.70710678118654752440f, 0.58778525229247312917f, 0.45399049973954679156f, 0.30901699437494742410f,       \
   0.15643446504023086901f, 0.0f}
#define RFX_S40_TABLE                                                                                       \
  {0.0f, 0.15643446504023086901f, 0.30901699437494742410f, 0.45399049973954679156f, 0.58778525229247312917f, \
   0.70710678118654752440f, 0.80901699437494742410f, 0.89100652418836786236f, 0.95105651629515357212f,       \
   0.98768834059513772619f, 1.0f}

// full 40th-root tables: cos / sin of 2*pi*i/40, i = 0..39
#define RFX_C40F_TABLE {1.00000000000000000000f, 0.98768834059513777035f, 0.95105651629515353118f, 0.89100652418836789881f, 0.80901699437494745126f, 0.70710678118654757274f, 0.58778525229247313710f, 0.45399049973954680448f, 0.30901699437494745126f, 0.15643446504023092447f, 0.00000000000000006123f, -0.15643446504023059140f, -0.30901699437494734024f, -0.45399049973954669346f, -0.58778525229247302608f, -0.70710678118654746172f, -0.80901699437494734024f, -0.89100652418836778779f, -0.95105651629515353118f, -0.98768834059513765933f, -1.00000000000000000000f, -0.98768834059513777035f, -0.95105651629515375323f, -0.89100652418836789881f, -0.80901699437494756229f, -0.70710678118654768376f, -0.58778525229247324813f, -0.45399049973954691550f, -0.30901699437494756229f, -0.15643446504023103549f, -0.00000000000000018370f, 0.15643446504023067467f, 0.30901699437494722922f, 0.45399049973954663795f, 0.58778525229247291506f, 0.70710678118654735069f, 0.80901699437494734024f, 0.89100652418836778779f, 0.95105651629515353118f, 0.98768834059513765933f}
#define RFX_S40F_TABLE {0.00000000000000000000f, 0.15643446504023086896f, 0.30901699437494739575f, 0.45399049973954674897f, 0.58778525229247313710f, 0.70710678118654746172f, 0.80901699437494745126f, 0.89100652418836778779f, 0.95105651629515353118f, 0.98768834059513777035f, 1.00000000000000000000f, 0.98768834059513777035f, 0.95105651629515364220f, 0.89100652418836789881f, 0.80901699437494745126f, 0.70710678118654757274f, 0.58778525229247324813f, 0.45399049973954685999f, 0.30901699437494750677f, 0.15643446504023097998f, 0.00000000000000012246f, -0.15643446504023073018f, -0.30901699437494689615f, -0.45399049973954669346f, -0.58778525229247302608f, -0.70710678118654746172f, -0.80901699437494734024f, -0.89100652418836778779f, -0.95105651629515353118f, -0.98768834059513765933f, -1.00000000000000000000f, -0.98768834059513777035f, -0.95105651629515364220f, -0.89100652418836800983f, -0.80901699437494756229f, -0.70710678118654768376f, -0.58778525229247335915f, -0.45399049973954697101f, -0.30901699437494761780f, -0.15643446504023111876f}

// P1 forward: v[k1] = sum_{j=0..9} u[j] * w40^{j*k1}, k1 = 0..20, w40 = exp(-2*pi*i/40), u real.
//   j = 2p+s:   E[k] = sum_p u[2p] w20^{pk},   O[k] = sum_p u[2p+1] w40^{(2p+1)k}
//   v[k] = E[k] + O[k],   v[20-k] = conj(E[k] - O[k])                                  (u real)
// and, splitting p into even / odd (Ee, Eo, Oe, Oo), the mirror k -> 10-k comes for free:
//   E[10-k] = conj(Ee - Eo),   O[10-k] = -i * conj(Oe - Oo)
// so only k = 1..4 need the trigonometric sums (k = 0, 5, 10 are sign patterns / eighth roots).
// The rows come out in groups {0, 20, 10}, {5, 15}, {k, 20-k, 10-k, 10+k} (k = 1..4); each group is handed
// to `put(row, value)` as soon as it exists, so that a caller which twiddles and stores it right away
// never holds more than one group (8 registers instead of 42).  RFX_SCHED_FENCE keeps the compiler's
// scheduler from re-clustering the groups (it would otherwise compute all 21 rows first and spill).
#if defined(__HIP_DEVICE_COMPILE__)
#define RFX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define RFX_SCHED_FENCE() ((void)0)
#endif

template <class PUT>
RFX_HD void p1_forward_rows(const float (&u)[10], PUT put) {
  constexpr float C40[40] = RFX_C40F_TABLE;
  constexpr float S40[40] = RFX_S40F_TABLE;
  constexpr float R = 0.70710678118654752440f;
  const float e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], e4 = u[8];
  const float o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7], o4 = u[9];
  {  // k = 0 and k = 10
    const float ee = e0 + e2 + e4, eo = e1 + e3, oe = o0 + o2 + o4, oo = o1 + o3;
    const float E0 = ee + eo, O0 = oe + oo;
    put(0, cf{E0 + O0, 0.f});
    put(20, cf{E0 - O0, 0.f});
    put(10, cf{ee - eo, -(oe - oo)});  // E[10] = ee - eo (real), O[10] = -i (oe - oo)
  }
  {  // k = 5: w20^{5p} = (-i)^p, w40^{5(2p+1)} = w8^{2p+1}
    const float Er = e0 - e2 + e4, Ei = e3 - e1;
    const float Or = R * ((o0 + o3 + o4) - (o1 + o2)), Oi = R * ((o2 + o3) - (o0 + o1 + o4));
    put(5, cf{Er + Or, Ei + Oi});
    put(15, cf{Er - Or, -(Ei - Oi)});
  }
  RFX_SCHED_FENCE();
#pragma unroll
  for (int k = 1; k <= 4; ++k) {
    // w20^{pk} = w40^{2pk}; forward kernel exp(-i theta) = cos - i sin
    const float eer = fmaf(C40[(8 * k) % 40], e4, fmaf(C40[(4 * k) % 40], e2, e0));
    const float eei = -fmaf(S40[(8 * k) % 40], e4, S40[(4 * k) % 40] * e2);
    const float eor = fmaf(C40[(6 * k) % 40], e3, C40[(2 * k) % 40] * e1);
    const float eoi = -fmaf(S40[(6 * k) % 40], e3, S40[(2 * k) % 40] * e1);
    const float oer = fmaf(C40[(9 * k) % 40], o4, fmaf(C40[(5 * k) % 40], o2, C40[k % 40] * o0));
    const float oei = -fmaf(S40[(9 * k) % 40], o4, fmaf(S40[(5 * k) % 40], o2, S40[k % 40] * o0));
    const float oor = fmaf(C40[(7 * k) % 40], o3, C40[(3 * k) % 40] * o1);
    const float ooi = -fmaf(S40[(7 * k) % 40], o3, S40[(3 * k) % 40] * o1);
#if defined(__HIP_DEVICE_COMPILE__) && defined(RFX_PK)
    {  // the same four outputs on (re, im) pairs: E = Ee + Eo, O = Oe + Oo, F' = Ee - Eo, D = Oe - Oo
      const c2 ee{eer, eei}, eo{eor, eoi}, oe{oer, oei}, oo{oor, ooi};
      const c2 E = ee + eo, O = oe + oo, Fp = ee - eo, D = oe - oo;
      put(k, pk_to(E + O));
      put(20 - k, pk_to(pk_cj_sub(E, O)));
      put(10 - k, pk_to(pk_cj_add_i(Fp, D)));  // conj(F') + (-i) conj(D) = conj(F' + i D)
      put(10 + k, pk_to(pk_sub_i(Fp, D)));     // conj(conj(F') - (-i) conj(D)) = F' - i D
    }
#else
    const float Er = eer + eor, Ei = eei + eoi, Or = oer + oor, Oi = oei + ooi;  // E[k], O[k]
    put(k, cf{Er + Or, Ei + Oi});
    put(20 - k, cf{Er - Or, -(Ei - Oi)});
    // mirror 10-k:  E[10-k] = conj(Ee - Eo);  O[10-k] = -i conj(D), D = Oe - Oo:  -i (dr - i di) = -di - i dr
    const float Fr = eer - eor, Fi = -(eei - eoi);
    const float dr = oer - oor, di = oei - ooi;
    const float Gr = -di, Gi = -dr;
    put(10 - k, cf{Fr + Gr, Fi + Gi});
    put(10 + k, cf{Fr - Gr, -(Fi - Gi)});
#endif
    RFX_SCHED_FENCE();
  }
}
RFX_HD void p1_forward(const float (&u)[10], cf (&v)[21]) {
  p1_forward_rows(u, [&v](int k, cf x) { v[k] = x; });
}

// P1 inverse: y[j] = 0.5*(V0.re + (-1)^j V20.re) + sum_{k=1..19} Re(V[k] w40^{-k j}),  j = 0..9
// (the caller folds the factor 2/N and the synthesis window into one multiplier).
// Rows k and 20-k pair up (w40^{-(20-k) j} = (-1)^j conj(w40^{-k j})):  B = V[k] + conj(V[20-k]) feeds the
// even outputs, D = V[k] - conj(V[20-k]) the odd ones; then k and 10-k pair up once more:
//   even j = 2p :  w40^{-(10-k) 2p} = (-1)^p conj(w40^{-2kp})        ->  B[k] +- conj(B[10-k])
//   odd  j = 2p+1: w40^{-(10-k)(2p+1)} = i (-1)^p conj(w40^{-k(2p+1)}) ->  D[k] +- (-i) conj(D[10-k])
// leaving four complex-times-constant real parts per output plus the k = 5 and k = 10 sign patterns.
// The rows are pulled through `get(row)` group by group ({0, 20, 10}, {5, 15}, then {k, 20-k, 10-k, 10+k})
// and accumulated into y at once: a caller that reads them from LDS never holds more than one group.
// `raw(row)` fetches a row, `fix(row, value)` finishes it (removes the twiddle): the fetches of group k+1 are
// issued before the arithmetic of group k, so an LDS round trip hides behind ~50 instructions.
template <class RAW, class FIX>
RFX_HD void p1_inverse_rows(RAW raw, FIX fix, float (&y)[10]) {
  constexpr float C40[40] = RFX_C40F_TABLE;
  constexpr float S40[40] = RFX_S40F_TABLE;
  constexpr float R = 0.70710678118654752440f;
  cf n0 = raw(1), n1 = raw(19), n2 = raw(9), n3 = raw(11);
  {
    const cf V0 = fix(0, raw(0)), V20 = fix(20, raw(20)), V10 = fix(10, raw(10)), V5 = fix(5, raw(5)), V15 = fix(15, raw(15));
    const float he = 0.5f * (V0.re + V20.re), ho = 0.5f * (V0.re - V20.re);
    const cf B5{V5.re + V15.re, V5.im - V15.im}, D5{V5.re - V15.re, V5.im + V15.im};
    // k = 5, odd outputs: Re(D5 w8^{-(2p+1)}),  w8^{-1} = R(1+i), w8^{-3} = R(-1+i), w8^{-5} = R(-1-i), w8^{-7} = R(1-i)
    const float s5 = R * (D5.re - D5.im), t5 = R * (D5.re + D5.im);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      // even j = 2p: k = 10: (-1)^p V10.re ;  k = 5: Re(B5 i^p)
      float a = (p & 1) ? he - V10.re : he + V10.re;
      a += (p % 4 == 0) ? B5.re : (p % 4 == 1) ? -B5.im : (p % 4 == 2) ? -B5.re : B5.im;
      y[2 * p] = a;
      // odd j = 2p+1: k = 10: (-1)^{p+1} V10.im ;  k = 5: p=0: s5, p=1: -t5, p=2: -s5, p=3: t5, p=4: s5
      float c = (p & 1) ? ho + V10.im : ho - V10.im;
      c += (p % 4 == 0) ? s5 : (p % 4 == 1) ? -t5 : (p % 4 == 2) ? -s5 : t5;
      y[2 * p + 1] = c;
    }
  }
  RFX_SCHED_FENCE();
#pragma unroll
  for (int k = 1; k <= 4; ++k) {
    const cf r0 = n0, r1 = n1, r2 = n2, r3 = n3;
    if (k < 4) {
      n0 = raw(k + 1);
      n1 = raw(19 - k);
      n2 = raw(9 - k);
      n3 = raw(11 + k);
    }
    RFX_SCHED_FENCE();
    const cf Vk = fix(k, r0), Vm = fix(20 - k, r1), Wk = fix(10 - k, r2), Wm = fix(10 + k, r3);
#if defined(__HIP_DEVICE_COMPILE__) && defined(RFX_PK)
    const c2 pBk = pk_add_cj(pk_of(Vk), pk_of(Vm)), pDk = pk_sub_cj(pk_of(Vk), pk_of(Vm));
    const c2 pBn = pk_add_cj(pk_of(Wk), pk_of(Wm)), pDn = pk_sub_cj(pk_of(Wk), pk_of(Wm));
    const cf Ge = pk_to(pk_add_cj(pBk, pBn)), He = pk_to(pk_sub_cj(pBk, pBn));
    const cf Go = pk_to(pk_sub_sw(pDk, pDn)), Ho = pk_to(pk_add_sw(pDk, pDn));
#else
    const cf Bk{Vk.re + Vm.re, Vk.im - Vm.im}, Dk{Vk.re - Vm.re, Vk.im + Vm.im};      // B[k], D[k]
    const cf Bn{Wk.re + Wm.re, Wk.im - Wm.im}, Dn{Wk.re - Wm.re, Wk.im + Wm.im};      // B[10-k], D[10-k]
    const cf Ge{Bk.re + Bn.re, Bk.im - Bn.im}, He{Bk.re - Bn.re, Bk.im + Bn.im};      // B[k] +- conj(B[10-k])
    // (-i) * conj(x + i y) = -y - i x
    const float qr = -Dn.im, qi = -Dn.re;
    const cf Go{Dk.re + qr, Dk.im + qi}, Ho{Dk.re - qr, Dk.im - qi};                  // D[k] +- (-i) conj(D[10-k])
#endif
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const cf ze = (p & 1) ? He : Ge;
      y[2 * p] = fmaf(-S40[(2 * k * p) % 40], ze.im, fmaf(C40[(2 * k * p) % 40], ze.re, y[2 * p]));
      const cf zo = (p & 1) ? Ho : Go;
      y[2 * p + 1] = fmaf(-S40[(k * (2 * p + 1)) % 40], zo.im, fmaf(C40[(k * (2 * p + 1)) % 40], zo.re, y[2 * p + 1]));
    }
    RFX_SCHED_FENCE();
  }
}
RFX_HD void p1_inverse(const cf (&V)[21], float (&y)[10]) {
  p1_inverse_rows([&V](int k) { return V[k]; }, [](int, cf c) { return c; }, y);
}

// ------------------------------------------------------------------------------------------------
// LDS "cube" addressing: element (k1, a|ka, b) of the 21 x 21 x 21 work array
// ------------------------------------------------------------------------------------------------
RFX_HD int cube_at(int k1, int a, int b) { return k1 * kRowStride + a * 21 + b; }

// Twiddles are passed as accessors `tw(i) -> cf` so that the kernels can stream them from a table
// (L2 / LDS) exactly where they are consumed instead of pinning 42 registers per table, while the
// host emulator indexes plain arrays.

// P1 + store: thread n' = 21a+b computes its 21 rows group by group, twiddles each by g(n')^k1 and
// scatters it to row k1 of the cube
template <class TW>
RFX_HD void p1_forward_store(const float (&u)[10], TW tw1, cf* cube, int npr) {
  p1_forward_rows(u, [&](int k1, cf v) { cube[k1 * kRowStride + npr] = k1 == 0 ? v : cmul(v, tw1(k1)); });
}
// load + P1': thread n' gathers rows k1, removes the twiddle and folds them into its 10 output hops
template <class TW>
RFX_HD void p1_load_inverse(const cf* cube, TW tw1, float (&y)[10], int npr) {
  p1_inverse_rows([&](int k1) { return cube[k1 * kRowStride + npr]; },
                  [&](int k1, cf c) { return k1 == 0 ? c : cmulc(c, tw1(k1)); }, y);
}
// P2 (forward, in place): thread (k1, b): DFT over a, then twiddle w441^{b*ka}.
// `stage(0)` runs before the butterflies and `stage(1)` once outputs 0..kTwSplit have been stored: the
// kernels fetch the twiddles 1..kTwSplit / kTwSplit+1..20 there (two batches of registers instead of 40).
constexpr int kTwSplit = 10;
struct NoStage {
  RFX_HD void operator()(int) const {}
};
template <class TW, class STAGE = NoStage>
RFX_HD void p2_forward(cf* cube, TW tw2, int k1, int b, STAGE stage = STAGE()) {
  cf x[21];
#pragma unroll
  for (int a = 0; a < 21; ++a) x[a] = cube[cube_at(k1, a, b)];
  stage(0);
  dft21<false>(x);
  cube[cube_at(k1, 0, b)] = x[0];
#pragma unroll
  for (int ka = 1; ka <= kTwSplit / 2; ++ka) cube[cube_at(k1, ka, b)] = cmul(x[ka], tw2(ka));
  stage(1);
#pragma unroll
  for (int ka = kTwSplit / 2 + 1; ka < 21; ++ka) cube[cube_at(k1, ka, b)] = cmul(x[ka], tw2(ka));
}
// P2' (inverse, in place): thread (k1, b): inverse DFT over ka
RFX_HD void p2_inverse(cf* cube, int k1, int b) {
  cf x[21];
#pragma unroll
  for (int ka = 0; ka < 21; ++ka) x[ka] = cube[cube_at(k1, ka, b)];
  dft21<true>(x);
#pragma unroll
  for (int a = 0; a < 21; ++a) cube[cube_at(k1, a, b)] = x[a];
}
// P3 (forward): thread (k1, ka): DFT over b -> R[kb] in registers
RFX_HD void p3_forward(const cf* cube, cf (&R)[21], int k1, int ka) {
#pragma unroll
  for (int b = 0; b < 21; ++b) R[b] = cube[cube_at(k1, ka, b)];
  dft21<false>(R);
}
// P3' (inverse): thread (k1, ka): inverse DFT over kb, conj twiddle, store
template <class TW, class STAGE = NoStage>
RFX_HD void p3_inverse(cf* cube, cf (&Z)[21], TW tw2, int k1, int ka, STAGE stage = STAGE()) {
  stage(0);
  dft21<true>(Z);
  cube[cube_at(k1, ka, 0)] = Z[0];
#pragma unroll
  for (int b = 1; b <= kTwSplit / 2; ++b) cube[cube_at(k1, ka, b)] = cmulc(Z[b], tw2(b));
  stage(1);
#pragma unroll
  for (int b = kTwSplit / 2 + 1; b < 21; ++b) cube[cube_at(k1, ka, b)] = cmulc(Z[b], tw2(b));
}

// ------------------------------------------------------------------------------------------------
// Griffin-Lim per-bin update (torchaudio functional.griffinlim loop body):
//   angles = rebuilt - m*tprev ; angles /= (|angles| + 1e-16) ; next = S*angles
// `a` below is already rebuilt - m*tprev: the STFT is linear, so the kernels analyse the signal
// x_k - m*x_{k-1} instead of subtracting two spectra (see rfx_gl.hip).
// ------------------------------------------------------------------------------------------------
// Round 6: `eps2` is the square of that 1e-16 in the units `a` arrives in.  The kernels analyse the signal times a power of two
// per row (GlRowScale: the row's magnitudes brought to the 2^25 the path was tuned and tested at), so that re^2 + im^2 can
// neither overflow (|a| > 1.8e19 squared to inf, rsq gave 0: silence, at max_value 1e20) nor vanish; the factor S / (|a| + eps)
// is the unscaled one up to that power of two, which the product with the scaled `a` cancels: the result is S * angles as before.
RFX_HD cf gl_project(cf a, float S, float eps2 = 1e-32f) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef RFX_PROJECT_SQRT_RCP
  // One quarter-rate instruction per bin: S / (|a| + 1e-16) = S * rsq(|a|^2 + 1e-32) up to fp32 rounding
  // whenever |a| > 1e-8 (the 1e-16 is then below half an ulp of |a|) and exactly 0 for a == 0, as in the
  // reference.  Only for 0 < |a| < 1e-8 - thirteen orders of magnitude below the spectra this path sees -
  // does the factor differ (it stays bounded by 1e16 either way); tests/test_gpu_stft_gl.py covers the
  // zero and tiny-magnitude cases.  v_rsq_f32 is 1 ulp, like the v_sqrt_f32 / v_rcp_f32 pair it replaces.
  const float sc = S * __builtin_amdgcn_rsqf(fmaf(a.re, a.re, fmaf(a.im, a.im, eps2)));
#else
  // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sqrt / divide expansions
  const float mag = __builtin_amdgcn_sqrtf(fmaf(a.re, a.re, a.im * a.im));
  const float sc = S * __builtin_amdgcn_rcpf(mag + 1e-16f);
#endif
#else
  const float mag = sqrtf(fmaf(a.re, a.re, a.im * a.im));
  const float sc = S / (mag + 1e-16f);
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(RFX_PK)
  return pk_to(pk_of(a) * pk_bc(sc));
#else
  return cf{a.re * sc, a.im * sc};
#endif
}

// Counter-based uniform [0, 1) values for the random starts (Griffin-Lim's rand_init phases, the SGD's initial spectrogram):
// stand-ins for torch.rand, whose Philox stream cannot be reproduced bit for bit by construction (parity tests inject the
// initial values).  One 32-bit mix per value on a per-frame key.  (Rounds 1-3 ran a 64-bit splitmix per value - three 64-bit
// multiplications, ~65 issue slots: the initial ISTFT launch cost as much as a full Griffin-Lim iteration because of it.)
RFX_HD unsigned mix32(unsigned x) {
  x ^= x >> 16;
  x *= 0x21f0aaadu;
  x ^= x >> 15;
  x *= 0x735a2d97u;
  x ^= x >> 15;
  return x;
}
// key of frame `frame` (the global frame index b * T + t) under `seed`
RFX_HD unsigned rand_frame_key(unsigned long long seed, unsigned long long frame) {
  return mix32((unsigned)frame ^ mix32((unsigned)(frame >> 32) ^ (unsigned)(seed >> 32)) ^ mix32((unsigned)seed ^ 0x632BE59Bu));
}
RFX_HD float rand_unit(unsigned key, int f) { return (float)(mix32(((unsigned)f * 0x9E3779B9u) ^ key) >> 8) * (1.0f / 16777216.0f); }
// (real, imaginary) of bin `bin`: the second word is drawn from the first with one more multiplication
RFX_HD cf rand_unit_pair(unsigned key, int bin) {
  const unsigned a = mix32(((unsigned)bin * 0x9E3779B9u) ^ key);
  unsigned b = (a ^ 0x85EBCA6Bu) * 0xC2B2AE35u;
  b ^= b >> 15;
  return cf{(float)(a >> 8) * (1.0f / 16777216.0f), (float)(b >> 8) * (1.0f / 16777216.0f)};
}

// reflect-padded sample index of torch.stft(center=True, pad_mode="reflect"): position p of the
// un-padded signal of length L
RFX_HD int reflect_index(int p, int L) {
  if (p < 0) p = -p;
  if (p >= L) p = 2 * (L - 1) - p;
  return p;
}

}  // namespace rfx
