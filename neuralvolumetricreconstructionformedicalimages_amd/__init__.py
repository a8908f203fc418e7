"""MI355X-native Neural Attenuation Field hot path (hash-grid encoder, sigma-MLP, ray march, Adam) behind the
reference project's Python surface.  Compute lives in lib/libnaf_hip.so (hand-written HIP for gfx950, C ABI in
include/naf_hip.h); PyTorch-ROCm supplies device memory, streams and torch.distributed only."""

__version__ = "0.1.0"
