"""lens_amd: MI355X-native implementation of the LENS SNN inference-and-matching hot path.

Layout: csrc/ (HIP kernels + C ABI, built to liblens_amd.so), _lib (ctypes binding),
engine (operator layer), run_model / blitnet / dataset / metrics / loggers / checker (host-side
mirror of the reference's Python surface for this path), synthetic (benchmark inputs),
dist (place-database sharding over RCCL).
"""
__version__ = "0.1.0"
