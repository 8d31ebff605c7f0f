"""ctypes signatures of the model-side entry points of include/py4cast_hip.h (see _lib.py)."""

from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

P, I, L, F = c_void_p, c_int, c_int64, c_float

SIGNATURES = {}
OTHER = {}
