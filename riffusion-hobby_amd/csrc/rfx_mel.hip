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
#include <hip/hip_runtime.h>

#include "rfx_core.h"
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

  for (int bi = 0; bi < a.n_kblocks; ++bi) {
    const int k0 = a.kblocks[bi] * kMelBK;
    float4 fv[4], gv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + fk + 8 * i, m = m0 + fm4;
      if (m + 3 < a.M && (a.M & 3) == 0) {
        fv[i] = *reinterpret_cast<const float4*>(a.fbs + (size_t)k * a.M + m);
      } else {
        float t4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < 4; ++c)
          if (m + c < a.M) t4[c] = a.fbs[(size_t)k * a.M + m + c];
        fv[i] = float4{t4[0], t4[1], t4[2], t4[3]};
      }
      const int n = n0 + gn + 32 * i;
      gv[i] = (n < a.N) ? *reinterpret_cast<const float4*>(a.mag + (size_t)n * kFrameStride + k0 + gk4)
                        : float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // previous step's reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&Fs[fk + 8 * i][fm4]) = fv[i];
      float* d = &Ms[gn + 32 * i][gk4];
      d[0] = gv[i].x; d[1] = gv[i].y; d[2] = gv[i].z; d[3] = gv[i].w;
    }
    __syncthreads();
#pragma unroll
    for (int kp = 0; kp < kMelBK / 2; ++kp) {
      const int k = 2 * kp + lk;
      const float a0 = Fs[k][wm + li], a1 = Fs[k][wm + 32 + li];
      const float b0 = Ms[wn + li][k], b1 = Ms[wn + 32 + li][k];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // C/D layout of 32x32 MFMA: col = lane & 31 (frame), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (mel)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn + 32 * j + li;
    if (n >= a.N) continue;
    const int b = n / a.T, t = n - b * a.T;
    float* dst = a.out + (size_t)b * a.M * a.T + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < a.M) dst[(size_t)m * a.T] = acc[i][j][r];
      }
  }
}

hipError_t launch_mel_gemm(const MelArgs& a, hipStream_t stream) {
  dim3 grid((a.M + kMelBM - 1) / kMelBM, (a.N + kMelBN - 1) / kMelBN);
  hipLaunchKernelGGL(mel_gemm_kernel, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace rfx
