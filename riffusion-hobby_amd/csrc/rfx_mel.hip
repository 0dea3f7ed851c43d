// rfx_mel.hip - mel projection on the fp32 matrix cores (replaces torchaudio.transforms.MelScale,
// riffusion/spectrogram_converter.py:76-84, called at :185:  mel = (|X|^T @ fb)^T ).
//
//   out[b][m][t] = sum_p  fbs[p][m] * mag[b*T + t][p]        p = slot position, 0 .. kFrameStride-1
//
// `mag` is the slot-major magnitude stream written by the STFT kernel; `fbs` is the filterbank with
// its rows permuted to slot order (duplicate / padding positions are zero rows), so the product
// equals the reference's GEMM over the 8821 bins up to summation order.  K blocks whose filterbank
// rows are all zero (bins above f_max) are skipped through a block list built at plan creation.
//
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD): the A operand is the filterbank
// (rows = mel), the B operand the magnitudes (columns = frames), so each accumulator register holds
// one mel row across 32 consecutive frames and the (B, M, T) store is 128 contiguous bytes per
// register.  Workgroup = 4 waves = 128 mel x 128 frames, each wave 64 x 64 (2 x 2 MFMA tiles).
//
// Pipeline per K block (32 slot positions): the global loads of block i+1 are issued right after
// block i has been written to LDS and fly during its 64 MFMAs; inside the block the LDS fragments of
// step kp+1 are fetched before the MFMAs of step kp.  All global accesses are buffer-descriptor
// addressed (rows beyond N read as zero through the descriptor's range check), the filterbank is
// padded to a multiple of 128 columns at plan creation, so the loop has no bounds tests and the
// kernel fits 3 workgroups per CU (accumulators in AGPRs).
#include <hip/hip_runtime.h>

#include "rfx_core.h"
#include "rfx_frame.hip.h"
#include "rfx_kernels.h"

namespace rfx {

using f32x16 = float __attribute__((ext_vector_type(16)));

constexpr int kMelBM = 128;  // mel rows per workgroup
constexpr int kMelBN = 128;  // frames per workgroup
constexpr int kMelBK = 32;   // slot positions per step
constexpr int kMagPitch = kMelBK + 1;

__global__ void __launch_bounds__(256) mel_gemm_kernel(MelArgs a) {
  __shared__ float Fs[kMelBK][kMelBM];        // filterbank tile  [k][m]
  __shared__ float Ms[kMelBN][kMagPitch];     // magnitude tile   [n][k], odd pitch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * kMelBM, n0 = blockIdx.y * kMelBN;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;  // wave's 64 x 64 corner inside the tile
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging assignments: filterbank tile 32 x 128 floats = 1024 float4 -> 4 per thread (rows tid/32 + 8*i)
  //                      magnitude tile 128 x 32 floats = 1024 float4 -> 4 per thread (rows tid/8 + 32*i)
  const int fk = tid >> 5, fm4 = (tid & 31) * 4;
  const int gn = tid >> 3, gk4 = (tid & 7) * 4;
  const int rows = min(kMelBN, a.N - n0);  // frames of this tile that exist
  const rsrc_t rF = make_rsrc(a.fbs + m0, ((size_t)kFrameStride * a.Mp - m0) * sizeof(float));
  const rsrc_t rG = make_rsrc(a.mag + (size_t)n0 * kFrameStride, (size_t)rows * kFrameStride * sizeof(float));
  const unsigned offF = (unsigned)((fk * a.Mp + fm4) * sizeof(float));
  const unsigned offG = (unsigned)((gn * kFrameStride + gk4) * sizeof(float));
  const unsigned stepF = (unsigned)(8 * a.Mp * sizeof(float)), stepG = (unsigned)(32 * kFrameStride * sizeof(float));

  v4f fv[4], gv[4];
  auto fetch = [&](int bi) {
    const int k0 = a.kblocks[bi] * kMelBK;
    const unsigned sF = (unsigned)(k0 * a.Mp * sizeof(float)), sG = (unsigned)(k0 * sizeof(float));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fv[i] = ld4(rF, offF + i * stepF, sF);
      gv[i] = ld4(rG, offG + i * stepG, sG);  // rows >= N are outside the descriptor: zero
    }
  };
  fetch(0);

  for (int bi = 0; bi < a.n_kblocks; ++bi) {
    __syncthreads();  // previous block's fragment reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<v4f*>(&Fs[fk + 8 * i][fm4]) = fv[i];
      float* d = &Ms[gn + 32 * i][gk4];
      d[0] = gv[i].x; d[1] = gv[i].y; d[2] = gv[i].z; d[3] = gv[i].w;
    }
    __syncthreads();
    if (bi + 1 < a.n_kblocks) fetch(bi + 1);  // in flight during the 64 MFMAs below

    float fa[2][2], fb[2][2];
    auto frag = [&](int kp, int s) {
      const int k = 2 * kp + lk;
      fa[s][0] = Fs[k][wm + li];
      fa[s][1] = Fs[k][wm + 32 + li];
      fb[s][0] = Ms[wn + li][k];
      fb[s][1] = Ms[wn + 32 + li][k];
    };
    frag(0, 0);
#pragma unroll
    for (int kp = 0; kp < kMelBK / 2; ++kp) {
      const int s = kp & 1;
      if (kp + 1 < kMelBK / 2) frag(kp + 1, s ^ 1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][1], acc[1][1], 0, 0, 0);
    }
  }

  // C/D layout of 32x32 MFMA: col = lane & 31 (frame), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (mel).
  // Stores go through one descriptor based at the tile's first clip: per-lane offset = (clip, 4*lk, t),
  // scalar offset = the wave-uniform part of the mel row; lanes past N get an out-of-range offset (dropped).
  const int b0 = n0 / a.T;
  const size_t clip_floats = (size_t)a.M * a.T;
  const rsrc_t rO = make_rsrc(a.out + b0 * clip_floats, ((size_t)a.N * a.M - b0 * clip_floats) * sizeof(float));
  const bool full_m = m0 + kMelBM <= a.M;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn + 32 * j + li;
    const int b = n / a.T, t = n - b * a.T;
    const unsigned voff = n < a.N ? (unsigned)((((b - b0) * a.M + 4 * lk) * a.T + t) * sizeof(float)) : 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2);  // wave-uniform part of the row
        if (full_m || mu + 4 * lk < a.M) st1(acc[i][j][r], rO, voff, (unsigned)(mu * a.T * sizeof(float)));
      }
  }
}

hipError_t launch_mel_gemm(const MelArgs& a, hipStream_t stream) {
  dim3 grid((a.M + kMelBM - 1) / kMelBM, (a.N + kMelBN - 1) / kMelBN);
  hipLaunchKernelGGL(mel_gemm_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace rfx
