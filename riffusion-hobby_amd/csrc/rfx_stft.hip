// rfx_stft.hip - forward framed transform (replaces torchaudio.transforms.Spectrogram(power=None) +
// torch.abs, riffusion/spectrogram_converter.py:47-59, :179-182) and the layout converters between
// the reference's (B, n_stft, T) tensors and the slot-major frames the gfx950 kernels stream.
#include "rfx_frame.hip.h"
#include "rfx_kernels.h"

namespace rfx {

// one thread per slot position, 16 consecutive frames: reads 64 contiguous bytes of a bin row and
// writes position-contiguous (coalesced) floats into 16 frames
template <bool COMPLEX>
__global__ void __launch_bounds__(256) pack_kernel(const void* __restrict__ src_, void* __restrict__ dst_, int T) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= kFrameStride) return;
  const int tchunk = blockIdx.y * 16;
  const int clip = blockIdx.z;
  int q, kb;
  const bool real_slot = COMPLEX ? pos_c_to_slot(p, q, kb) : pos_f_to_slot(p, q, kb);
  if (!real_slot) {  // padding lane: keep it zero
    const int ntp = min(16, T - tchunk);
    for (int i = 0; i < ntp; ++i) {
      const size_t o = ((size_t)clip * T + tchunk + i) * kFrameStride + p;
      if (COMPLEX) reinterpret_cast<cf*>(dst_)[o] = cf{0.f, 0.f}; else reinterpret_cast<float*>(dst_)[o] = 0.f;
    }
    return;
  }
  bool cj;
  const int bin = slot_bin(q / 21, q % 21, kb, &cj);
  const int nt = min(16, T - tchunk);
  if (COMPLEX) {
    const cf* src = reinterpret_cast<const cf*>(src_) + ((size_t)clip * kBins + bin) * T + tchunk;
    cf* dst = reinterpret_cast<cf*>(dst_) + ((size_t)clip * T + tchunk) * kFrameStride + p;
    for (int i = 0; i < nt; ++i) {
      cf v = src[i];
      if (cj) v.im = -v.im;
      dst[(size_t)i * kFrameStride] = v;
    }
  } else {
    const float* src = reinterpret_cast<const float*>(src_) + ((size_t)clip * kBins + bin) * T + tchunk;
    float* dst = reinterpret_cast<float*>(dst_) + ((size_t)clip * T + tchunk) * kFrameStride + p;
    for (int i = 0; i < nt; ++i) dst[(size_t)i * kFrameStride] = src[i];
  }
}

// slots -> (B, n_stft, T) complex, reading each bin from its primary slot (test / debugging path)
__global__ void __launch_bounds__(256) unpack_complex_kernel(const cf* __restrict__ slots, cf* __restrict__ out, int T) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= kBins) return;
  const int tchunk = blockIdx.y * 16;
  const int clip = blockIdx.z;
  int k = bin;
  bool cj = false;
  if (k % 40 > 20) { k = kNfft - k; cj = true; }
  const int k1 = k % 40, kp = k / 40;
  const int ka = kp % 21, kb = kp / 21;
  const int p = slot_pos_c(k1 * 21 + ka, kb);
  const int nt = min(16, T - tchunk);
  for (int i = 0; i < nt; ++i) {
    cf v = slots[((size_t)clip * T + tchunk + i) * kFrameStride + p];
    if (cj) v.im = -v.im;
    out[((size_t)clip * kBins + bin) * T + tchunk + i] = v;
  }
}

// float slots -> (B, n_stft, T), reading each bin from its primary slot
__global__ void __launch_bounds__(256) unpack_mag_kernel(const float* __restrict__ slots, float* __restrict__ out, int T) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= kBins) return;
  const int tchunk = blockIdx.y * 16;
  const int clip = blockIdx.z;
  int k = bin;
  if (k % 40 > 20) k = kNfft - k;
  const int k1 = k % 40, kp = k / 40;
  const int p = slot_pos_f(k1 * 21 + kp % 21, kp / 21);
  const int nt = min(16, T - tchunk);
  for (int i = 0; i < nt; ++i)
    out[((size_t)clip * kBins + bin) * T + tchunk + i] = slots[((size_t)clip * T + tchunk + i) * kFrameStride + p];
}
hipError_t launch_unpack_mag(const float* slots, float* out_bft, int B, int T, hipStream_t stream) {
  dim3 grid((kBins + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(unpack_mag_kernel, grid, dim3(256), 0, stream, slots, out_bft, T);
  return hipGetLastError();
}

hipError_t launch_pack_mag(const float* lin_bft, float* S_slots, int B, int T, hipStream_t stream) {
  dim3 grid((kFrameStride + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(pack_kernel<false>, grid, dim3(256), 0, stream, (const void*)lin_bft, (void*)S_slots, T);
  return hipGetLastError();
}
hipError_t launch_pack_angles(const cf* ang_bft, cf* slots, int B, int T, hipStream_t stream) {
  dim3 grid((kFrameStride + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(pack_kernel<true>, grid, dim3(256), 0, stream, (const void*)ang_bft, (void*)slots, T);
  return hipGetLastError();
}
hipError_t launch_unpack_complex(const cf* slots, cf* out_bft, int B, int T, hipStream_t stream) {
  dim3 grid((kBins + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(unpack_complex_kernel, grid, dim3(256), 0, stream, slots, out_bft, T);
  return hipGetLastError();
}

// ---- forward STFT: frame t of clip b is centred on sample 441*t of the reflect-padded waveform
__global__ void __launch_bounds__(kThreads) stft_kernel(StftArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, a.tw1, a.tw2);
  const float* __restrict__ winp = a.win + t.npr;
  __syncthreads();

  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  const int clip = blockIdx.x / chunks;
  const int f0 = (blockIdx.x - clip * chunks) * a.frames_per_block;
  const int f1 = min(a.T, f0 + a.frames_per_block);
  const float* __restrict__ x = a.wave + (size_t)clip * a.Lw;

  for (int fr = f0; fr < f1; ++fr) {
    float u[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int p = reflect_index((fr + j - kHalfHops) * kHop + t.npr, a.Lw);
      u[j] = x[p] * winp[j * kHop];
    }
    cf R[21];
    frame_forward(u, R, f, t, [] {});
    const size_t fbase = ((size_t)clip * a.T + fr) * kFrameStride;
    // padded owner index of this thread's slots; seven idle lanes store zeros to the padding positions 63, 127, ...
    // (keeps the mel GEMM free of garbage)
    const bool store = t.active || t.pad >= 0;
    const int q = t.active ? slot_qp(t.npr) : 64 * t.pad + 63;
    if (a.mag && store) {
      float m[21];
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) m[kb] = t.active ? sqrtf(fmaf(R[kb].re, R[kb].re, R[kb].im * R[kb].im)) : 0.f;
      float4* d4 = reinterpret_cast<float4*>(a.mag + fbase);
#pragma unroll
      for (int i = 0; i < 5; ++i) d4[i * kQPad + q] = float4{m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]};
      a.mag[fbase + 20 * kQPad + q] = m[20];
    }
    if (a.spec && store) {
      if (!t.active) {
#pragma unroll
        for (int kb = 0; kb < 21; ++kb) R[kb] = cf{0.f, 0.f};
      }
      float4* d4 = reinterpret_cast<float4*>(a.spec + fbase);
#pragma unroll
      for (int i = 0; i < 10; ++i)
        d4[i * kQPad + q] = float4{R[2 * i].re, R[2 * i].im, R[2 * i + 1].re, R[2 * i + 1].im};
      a.spec[fbase + 20 * kQPad + q] = R[20];
    }
    __syncthreads();  // the next frame's P1 overwrites rows other waves may still be reading in P3
  }
}

// ---- fused forward path (replaces Spectrogram(power=None) -> abs -> MelScale, spectrogram_converter.py:165-185):
// the frame engine as above; the magnitudes of the frame are then parked in LDS (each thread's 21 in the cube
// elements it has just consumed) and every thread forms the mel amplitudes of one or two filters as the banded dot product
//     mel[m] = sum_{i < band_len[m]} fb[band_lo[m] + i][m] * |X[band_lo[m] + i]|
// i.e. the reference's `|X|^T @ fb` with the structural zeros of the triangular filterbank left out (7 976 of its
// 4.5 M products for the default bank), summed in increasing bin order.  The 1.2 GB magnitude stream and the dense
// GEMM of the unfused path disappear; only (B, M, T) floats are written.
__global__ void __launch_bounds__(kThreads, 4) stft_mel_kernel(StftMelArgs a) {  // 128 VGPRs: two workgroups per CU (it took 171 = one per CU without the bound)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, a.tw1, a.tw2);
  const float* __restrict__ winp = a.win + t.npr;
  float* magb = reinterpret_cast<float*>(smem);  // float view of the cube: slot (k1, ka, kb) -> magb[2 * cube_at(k1, ka, 0) + kb]
  __syncthreads();

  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  const int clip = blockIdx.x / chunks;
  const int f0 = (blockIdx.x - clip * chunks) * a.frames_per_block;
  const int f1 = min(a.T, f0 + a.frames_per_block);
  const float* __restrict__ x = a.wave + (size_t)clip * a.Lw;

  // the (up to two) filters of this thread: m0 = threadIdx, m1 = threadIdx + kThreads; waves walk bands of similar length
  const int m0 = threadIdx.x, m1 = threadIdx.x + kThreads;
  const bool has0 = m0 < a.M, has1 = m1 < a.M;
  const int n0 = has0 ? a.band_len[m0] : 0;
  const int n1 = has1 ? a.band_len[m1] : 0;
  float w10[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) w10[j] = winp[j * kHop];

  for (int fr = f0; fr < f1; ++fr) {
    float u[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int p = reflect_index((fr + j - kHalfHops) * kHop + t.npr, a.Lw);
      u[j] = x[p] * w10[j];
    }
    cf R[21];
    frame_forward(u, R, f, t, [] {});
    // P3 read exactly the 21 cube elements (k1, ka, 0..20) that only this thread touches: their memory takes the
    // thread's 21 magnitudes right away (no barrier, no bank-conflicted scatter); `band_addr` tells the mel threads where
    // the bins of their filter ended up
    if (t.active) {
      float* own = magb + 2 * cube_at(t.k1, t.idx, 0);
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) own[kb] = sqrtf(fmaf(R[kb].re, R[kb].re, R[kb].im * R[kb].im));
    }
    __syncthreads();
    {
      // eight weights / addresses in flight per step (tables zero-padded to a multiple of eight rows: the tail multiplies
      // a finite magnitude by zero; padding filters have length 0); the next step's are requested before the current eight
      // are summed; bins are summed in increasing order
      // frame-major scratch (512 contiguous floats per frame: whole-line stores); a tiled transpose brings it into the
      // reference's (B, M, T) layout afterwards - 4-byte stores T floats apart cost 10x the bytes in HBM writes
      float* __restrict__ row = a.mel_tm + ((size_t)clip * a.T + fr) * a.Mpad;
      // one code path for the (up to) two filters of a thread, run one after the other: its register arrays exist once
#pragma unroll 1
      for (int which = 0; which < 2; ++which) {
        const int m = which ? m1 : m0;
        if (m >= a.Mpad) break;  // wave-uniform: only the first wave(s) carry a second filter
        const int n = which ? n1 : n0;
        const float* __restrict__ wt = a.band_wt + m;
        const int* __restrict__ ad = a.band_addr + m;
        float s = 0.f;
        float wn[8];
        int an[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          wn[j] = wt[(size_t)j * a.Mpad];
          an[j] = ad[(size_t)j * a.Mpad];
        }
#pragma unroll 1
        for (int i = 0; i < n; i += 8) {
          float w[8], v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            w[j] = wn[j];
            v[j] = magb[an[j]];
          }
          if (i + 8 < n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              wn[j] = wt[(size_t)(i + 8 + j) * a.Mpad];
              an[j] = ad[(size_t)(i + 8 + j) * a.Mpad];
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) s = fmaf(w[j], v[j], s);
        }
        row[m] = m < a.M ? s : 0.f;
      }
    }
    __syncthreads();  // the next frame's P1 overwrites the magnitudes
  }
}

// (B, T, Mpad) frame-major mel amplitudes -> the reference's (B, M, T): 64 x 64 tiles through LDS, both sides coalesced
__global__ void __launch_bounds__(256) mel_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int M, int Mpad) {
  __shared__ float tile[64][65];
  const int t0 = blockIdx.x * 64, m0 = blockIdx.y * 64, b = blockIdx.z;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  for (int r = ly; r < 64; r += 4) {
    const int tt = t0 + r;
    tile[r][lx] = tt < T ? in[((size_t)b * T + tt) * Mpad + m0 + lx] : 0.f;  // m0 + lx < Mpad always (Mpad % 64 == 0)
  }
  __syncthreads();
  for (int r = ly; r < 64; r += 4) {
    const int m = m0 + r, tt = t0 + lx;
    if (m < M && tt < T) out[((size_t)b * M + m) * T + tt] = tile[lx][r];
  }
}

hipError_t launch_mel_transpose(const float* mel_tm, float* mel, int B, int T, int M, int Mpad, hipStream_t stream) {
  hipLaunchKernelGGL(mel_transpose_kernel, dim3((T + 63) / 64, Mpad / 64, B), dim3(256), 0, stream, mel_tm, mel, T, M, Mpad);
  return hipGetLastError();
}

hipError_t launch_stft_mel(const StftMelArgs& a, hipStream_t stream) {
  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  hipLaunchKernelGGL(stft_mel_kernel, dim3(a.B * chunks), dim3(kThreads), kFrameDynLdsBytes, stream, a);
  const hipError_t e = hipGetLastError();
  return e != hipSuccess ? e : launch_mel_transpose(a.mel_tm, a.mel, a.B, a.T, a.M, a.Mpad, stream);
}

hipError_t prepare_frame_kernels() {
  hipError_t e = hipFuncSetAttribute((const void*)stft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)stft_mel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes);
  return e != hipSuccess ? e : prepare_gl_kernels();
}

hipError_t launch_stft(const StftArgs& a, hipStream_t stream) {
  const size_t lds = kFrameDynLdsBytes;
  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  hipLaunchKernelGGL(stft_kernel, dim3(a.B * chunks), dim3(kThreads), lds, stream, a);
  return hipGetLastError();
}

}  // namespace rfx
