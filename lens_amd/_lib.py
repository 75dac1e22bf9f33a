"""ctypes binding of liblens_amd.so (the C ABI declared in include/lens_amd.h).

There is no CPU fallback: if the shared library is missing or a kernel is launched without a
GPU the call raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
# LENS_AMD_SO selects another in-tree build of the same sources (kernel experiments); the default is the product.
SO_PATH = os.environ.get("LENS_AMD_SO") or os.path.join(_PKG, "liblens_amd.so")
SRC = os.path.join(_PKG, "csrc", "lens_kernels.hip")
INCLUDES = [os.path.join(_PKG, "csrc", n)                            # textually included by SRC
            for n in ("feature_loop_n2f4.inc", "fire_loop.inc", "carry_l2_loop.inc", "output_loop_p1.inc", "output_loop_p2.inc", "output_loop_p3.inc",
                      "output_loop_p4.inc")]
HEADER = os.path.join(ROOT, "include", "lens_amd.h")

# -ffp-contract=off: the only fused operations are the explicit fmaf calls (numerics contract).
# -fno-slp-vectorize / -packed-fp32-ops: no packed f32 VALU (v_pk_add/mul/fma_f32) anywhere in the device code:
#   they have no rate advantage on gfx950, cannot take the |x| source modifier the event loop relies on, and
#   the target feature is switched off outright so that no optimisation pass can form them (the host pass of
#   hipcc does not know the feature and says so in a harmless warning).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
               "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
               "-fPIC", "-shared", "-Wall"]

_c_p = ctypes.c_void_p
_i = ctypes.c_int

# name -> (restype, argtypes); must list every symbol include/lens_amd.h declares
SIGNATURES = {
    "lens_amd_abi_version": (_i, []),
    "lens_amd_strerror": (ctypes.c_char_p, [_i]),
    "lens_accumulate_events": (_i, [_c_p, _c_p, _i, _i, _c_p, _c_p, _c_p]),
    "lens_build_threshold_codes": (_i, [_c_p, _i, _i, _c_p]),
    "lens_event_capacity_words": (ctypes.c_size_t, [_i, _i]),
    "lens_output_tile_places": (_i, [_i, _i, _i, _i]),
    "lens_output_reads_through_l2": (_i, [_i, _i, _i, _i]),
    "lens_output_layout": (_i, [_i, _i, _i, _i]),
    "lens_output_tail_places": (_i, [_i, _i, _i, _i]),
    "lens_event_scale": (_i, [_i, _i, _i, _i]),
    "lens_build_active_steps": (_i, [_c_p, _i, _i, _c_p]),
    "lens_feature_events": (_i, [_c_p, _i, _i, _c_p, _i, _c_p, _i, _c_p, _i, _i, _c_p, _i, _i,
                                 _c_p, _i, _c_p, _i, _c_p, _c_p, _c_p]),
    "lens_feature_currents": (_i, [_c_p, _i, _i, _c_p, _i, _c_p, _i, _c_p, _i, _i, _c_p, _i, _c_p]),
    "lens_feature_fire_scratch_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "lens_feature_fire": (_i, [_c_p, _i, _i, _i, _i, _c_p, _i, _c_p, _i, _c_p, _i, _c_p, _c_p]),
    "lens_output_counts": (_i, [_c_p, _i, _c_p, _i, _c_p, _i, _i, _i, _c_p, _i, _i, _i, _c_p, _i, _c_p]),
    "lens_forward_counts_static": (_i, [_c_p, _i, _i, _c_p, _i, _c_p, _i, _c_p, _i, _i, _c_p, _i, _i,
                                        _c_p, _c_p, _i, _i, _i, _c_p, _i, _c_p]),
    "lens_seqmatch": (_i, [_c_p, _i, _i, _i, _i, _c_p, _i, _c_p]),
    "lens_seqmatch_topk": (_i, [_c_p, _i, _i, _i, _i, _i, _i, _c_p, _c_p, _c_p]),
    "lens_topk_merge": (_i, [_c_p, _c_p, _i, _i, _i, _c_p, _c_p, _c_p]),
    "lens_output_counts_dense": (_i, [_c_p, _i, _i, _i, _i, _c_p, _i, _i, _c_p, _i, _c_p]),
    "lens_sad_matrix": (_i, [_c_p, _i, _c_p, _i, _i, _c_p, _i, _c_p]),
    "lens_train_in_pad": (_i, [_i]),
    "lens_layer_forward_clamped": (_i, [_c_p, _i, _i, _c_p, _i, _c_p, _i, _i, _c_p, _i, _c_p]),
    "lens_train_layer": (_i, [_c_p, _i, _c_p, _c_p, _c_p, _i, _c_p, _i, _c_p, _c_p, _i, _c_p, _c_p, _c_p, _c_p,
                              _i, _i, _i, _c_p]),
}

_lib = None
ABI_VERSION = 11


class LensAmdError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile lens_amd/csrc/lens_kernels.hip -> lens_amd/liblens_amd.so (gfx950)."""
    newest = max(os.path.getmtime(f) for f in [SRC, HEADER] + INCLUDES)
    if not force and os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= newest:
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", SO_PATH, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


def lib():
    """Load the shared library (once) and attach the prototypes. Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise LensAmdError(
                f"{SO_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        got = L.lens_amd_abi_version()
        if got != ABI_VERSION:
            raise LensAmdError(f"liblens_amd.so ABI version {got}, expected {ABI_VERSION}")
        _lib = L
    return _lib


def check(status, what):
    """Turn a C-ABI status into a Python exception (LensAmdError, a RuntimeError)."""
    if status == 0:
        return
    if status < 0:
        msg = lib().lens_amd_strerror(status).decode()
        raise LensAmdError(f"{what}: {msg} (status {status})")
    raise LensAmdError(f"{what}: HIP error {status}")
