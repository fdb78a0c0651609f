"""neural_graph_mapping_amd: MI355X-native (gfx950) render/train hot path of Neural Graph Mapping.

Only the per-field NeRF path is here (ray sampler, encodings, per-field MLP fwd/bwd, compositor,
losses, sparse Adam); the SLAM / mapping loop of the reference stays in the caller.  All compute
goes through hand-written HIP kernels behind the C ABI in include/ngm_hip.h.
"""
__all__ = ["models", "renderer", "ops", "distributed", "build"]
