"""repo_amd -- MI355X-native implementation of RePo's world-model + imagination update.

Host side mirrors the reference's Python surface (algorithms/repo: Dreamer, RePo;
models; common.buffers / common.utils); all arithmetic runs in hand-written HIP kernels
(repo_amd/csrc) behind the C ABI declared in include/repo_hip.h.
"""
__version__ = "0.1.0"
