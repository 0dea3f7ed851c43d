"""
MI355X-native drop-in for the spectrogram <-> audio path of riffusion-hobby.

Module names mirror the reference package (`riffusion.spectrogram_params`,
`riffusion.spectrogram_converter`, `riffusion.spectrogram_image_converter`, `riffusion.util.*`) so
that callers written against the reference import this implementation unchanged once
`riffusion-hobby_amd/` precedes the reference on `sys.path`.  All arithmetic runs in hand-written
HIP kernels for gfx950 reached through the C ABI of `librfx.so` (include/rfx.h); there is no CPU
fallback.
"""
__version__ = "0.1.0"
