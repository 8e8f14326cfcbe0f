"""mapf_gpt_amd -- MI355X-native (gfx950) implementation of MAPF-GPT's per-step hot path.

Importing the package does not need a GPU; anything that computes does (mapf_gpt_amd._lib fails
loudly when the HIP library is missing -- there is no CPU fallback on the product path).
"""
__version__ = "0.1.0"
