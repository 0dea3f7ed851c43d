// rfx_kernels.h - argument blocks and host launchers of the gfx950 kernels (internal to librfx.so)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "rfx_core.h"

namespace rfx {

struct GlArgs {
  const float* S;        // [B*T][kFrameStride] magnitudes, slot_pos_f order
  cf* tprev;             // [B*T][kFrameStride] previous rebuilt spectrum, slot_pos_c order
  const cf* angles0;     // optional injected initial angles, slot_pos_c order (MODE 0)
  const float* audio_in[2];   // parity partial sums of the previous iteration, [B][Lpad]
  float* audio_out[2];
  const float* out_scale;     // [L]  (2/N) / window-envelope  (torch.istft's division by sum w^2)
  const cf* tw1;              // [21][441]
  const cf* tw2;              // [21][21]
  const float* win;           // [4410]
  int B, T, L, Lpad, nruns;
  float mom;                  // momentum / (1 + momentum)
  unsigned long long seed;
  unsigned long long* timing;  // optional [nblocks][8] phase timers (RFX_TIMING builds only)
};

hipError_t launch_gl_iter(int mode, const GlArgs& g, int nblocks, hipStream_t stream);
int gl_blocks_per_cu();  // resident Griffin-Lim workgroups per CU (occupancy query)
hipError_t launch_gl_combine(const float* a0, const float* a1, float* out, int B, int L, int Lpad, hipStream_t stream);

// layout conversion between the reference's (B, n_stft, T) tensors and slot-major frames
hipError_t launch_pack_mag(const float* lin_bft, float* S_slots, int B, int T, hipStream_t stream);
hipError_t launch_pack_angles(const cf* ang_bft, cf* slots, int B, int T, hipStream_t stream);
hipError_t launch_unpack_complex(const cf* slots, cf* out_bft, int B, int T, hipStream_t stream);

// forward STFT magnitude of arbitrary-length waveforms: wave [B][Lw] -> mag slots [B*T][kFrameStride]
struct StftArgs {
  const float* wave;   // [B][Lw]
  float* mag;          // [B*T][kFrameStride] |X| in slot_pos_f order (nullable)
  cf* spec;            // [B*T][kFrameStride] X in slot_pos_c order (nullable)
  const cf* tw1;
  const cf* tw2;
  const float* win;
  int B, T, Lw, frames_per_block;
};
hipError_t launch_stft(const StftArgs& a, hipStream_t stream);

}  // namespace rfx
