"""dynhor_amd -- MI355X-native (gfx950) NeuS reconstruction hot path for EAST-J/Dynhor.

Host-side mirror of the upstream-NeuS Python surface over the C ABI in include/dynhor_hip.h.
The HIP extension is mandatory: there is no CPU or eager-PyTorch fallback in this package.
"""
__version__ = "0.1.0"
