"""ctypes loader for oracle/libmc_oracle.so (CPU marching-cubes oracle).
TEST INFRASTRUCTURE ONLY -- see oracle/mc_lewiner.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libmc_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE])
        lib = ctypes.CDLL(so)
        lib.mc_lewiner.restype = ctypes.c_int
        lib.mc_lewiner.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
        lib.mc_free.argtypes = [ctypes.c_void_p]
        _lib = lib
    return _lib


def default_level(vol):
    """skimage: 0.5 * (volume.min() + volume.max()), the sum rounded in float32."""
    return 0.5 * float(np.float32(vol.min()) + np.float32(vol.max()))


def marching_cubes(vol, level=None):
    """Lewiner marching cubes, 'ascent' orientation; returns (verts f32 [V,3] in
    array-axis order, faces i32 [F,3], level)."""
    lib = _load()
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    if level is None:
        level = default_level(vol)
    vp, fp = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf = ctypes.c_int(), ctypes.c_int()
    rc = lib.mc_lewiner(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], float(level),
                        ctypes.byref(vp), ctypes.byref(nv), ctypes.byref(fp), ctypes.byref(nf))
    verts = np.ctypeslib.as_array(ctypes.cast(vp, ctypes.POINTER(ctypes.c_float)), (nv.value, 3)).copy() if nv.value else np.zeros((0, 3), np.float32)
    faces = np.ctypeslib.as_array(ctypes.cast(fp, ctypes.POINTER(ctypes.c_int)), (nf.value, 3)).copy() if nf.value else np.zeros((0, 3), np.int32)
    lib.mc_free(vp)
    lib.mc_free(fp)
    if rc != 0:
        raise RuntimeError("No surface found at the given iso value.")
    return verts, faces, level
