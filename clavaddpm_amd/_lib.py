"""ctypes binding of libclava_hip.so (the C ABI declared in include/clava_hip.h).

The library handle lives in this module, never on a model object, so diffusion /
denoiser modules stay picklable (the reference pickles them whole,
`complex_pipeline.py:64,78`).  There is no CPU fallback: a missing library or a
failed call raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libclava_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "clava_hip.h")

_lib = None

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SILU = 0, 1, 2, 3


class ClavaHipError(RuntimeError):
    pass


class SplitOverflowError(ClavaHipError):
    """A tensor of the split-f16 training path left the float16 range since the last calibration (Trainer.run_loop
    catches it and re-runs the window on the float32 kernels)."""


def declared_symbols():
    """Names of every function include/clava_hip.h declares."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clava_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ClavaHipError(
                f"{LIB_PATH} is missing: build it with `python clavaddpm_amd/csrc/build.py` "
                "(there is no CPU fallback for the HIP path)")
        # torch ships its own libamdhip64 / libhsa-runtime64 (same sonames as /opt/rocm's).  Import torch first so that
        # this library binds to the HIP runtime torch already loaded; the other order puts two HSA runtimes into one
        # process and the second one sees no device.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.clava_last_error.restype = ctypes.c_char_p
        L.clava_version.restype = ctypes.c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().clava_last_error().decode(errors="replace")
        raise ClavaHipError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or NULL for None) as a ctypes void*."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise ClavaHipError("the HIP path needs device tensors (no CPU fallback): got a CPU tensor")


def c_int(v):
    return ctypes.c_int(int(v))


def c_float(v):
    return ctypes.c_float(float(v))


def c_long(v):
    return ctypes.c_long(int(v))


def c_double(v):
    return ctypes.c_double(float(v))


def c_u64(v):
    return ctypes.c_uint64(int(v) & 0xFFFFFFFFFFFFFFFF)
