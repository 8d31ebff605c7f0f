"""
ctypes binding of ``libpy4cast_hip.so`` (C ABI: ``include/py4cast_hip.h``).

The product path has NO CPU fallback: if the shared library is missing or a symbol is
absent, import-time / call-time errors are raised.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C py4cast_amd/csrc``.
"""

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P4C_LIB_PATH") or os.path.join(_HERE, "libpy4cast_hip.so")  # override: diagnostic builds

F32, BF16 = 0, 1
PROF_CONV3X3_C64, PROF_WGRAD3X3_C64, PROF_CONV3X3_C64_BWD = 1, 2, 4
LOSS_MSE, LOSS_L1 = 0, 1
MASK_NONE, MASK_FROM_NAN, MASK_F32, MASK_U8 = 0, 1, 2, 3

P = c_void_p
I = c_int
L = c_int64
F = c_float

# name -> argtypes (restype is int unless noted).  Mirrors include/py4cast_hip.h one to one.
SIGNATURES = {
    "p4c_build_x": [P, L, L, P, L, P, L, P, I, I, I, I, L, I, I, I, I, I, P],
    "p4c_build_x_bwd": [P, I, I, P, I, I, L, I, P],
    "p4c_build_x_masked": [P, L, L, P, L, P, L, P, I, I, I, I, L, I, I, I, I, I, P, I, I, I, I, P],
    "p4c_build_x_bwd_masked": [P, I, I, P, I, I, L, I, P, I, I, I, I, P],
    "p4c_ar_update_fwd": [P, L, P, I, I, P, L, P, P, P, P, P, L, I, L, I, F, I, P],
    "p4c_ar_update_bwd": [P, L, P, P, P, I, I, P, L, I, L, I, F, P],
    "p4c_mask_all_zero_count": [P, I, L, L, I, I, L, I, P, P],
    "p4c_weighted_loss_fwd": [P, L, L, P, L, L, P, I, P, P, F, P, I, P, P, I, I, L, I, P],
    "p4c_weighted_loss_map": [P, L, L, P, L, L, P, I, P, I, P, I, I, L, I, P],
    "p4c_weighted_loss_bwd": [P, P, L, L, P, L, L, P, I, P, P, F, P, I, P, L, L, I, I, L, I, P],
    "p4c_scaled_loss_fwd": [P, L, L, P, L, L, P, I, P, P, F, P, I, P, P, I, I, L, I, P],
    "p4c_acc_sums": [P, L, L, P, L, L, P, I, P, P, P, I, I, L, I, P],
    "p4c_unnormalize": [P, P, P, P, L, I, P],
    "p4c_unnormalize_planes": [P, P, P, P, L, L, I, P],
    "p4c_adamw_step": [P, P, P, P, L, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, L, P],
    "p4c_nan_moments": [P, P, L, P, P, I, L, I, P],
    "p4c_pack_standardize": [P, L, P, P, P, L, I, P],
    "p4c_ar_update_loss_fwd": [P, L, P, I, I, P, L, P, P, P, P, P, L, P, F, P, I, I, P, L, P, I, L, I, F, P],
    "p4c_ar_update_loss_fwd_next": [P, L, P, I, I, P, L, P, P, P, P, P, L, P, F, P, I, I, P, L, P, I, L, I, F,
                                    P, I, P, L, I, P, L, I, P],
    "p4c_ar_update_loss_bwd": [P, L, P, I, I, P, L, P, L, P, L, P, P, I, P, F, P, I, I, P, I, I, P, L, I, L, I, F, P],
    # a, a_scale, a_shift, wout, cout | prev, prev_bs, target, tgt_bs, std, mean, border, interior, new_state, new_bs, weights, num_interior,
    # masked_count, kind, loss_out, loss_stride, workspace, B, N, F, keep_prev | x_next, c_pad, statics, statics_bs, Fs, forcing, forcing_bs, Ff,
    # lgrad, lgrad_bs, stream
    "p4c_out_conv_update_loss_fwd": [P, P, P, P, I, P, L, P, L, P, P, P, P, P, L, P, F, P, I, P, L, P, I, L, I, F,
                                     P, I, P, L, I, P, L, I, P, L, P],
    "p4c_ar_update_loss_fwd_next_saved": [P, L, P, I, I, P, L, P, P, P, P, P, L, P, F, P, I, I, P, L, P, I, L, I, F,
                                          P, I, P, L, I, P, L, I, P, L, P],
    "p4c_ar_update_loss_bwd_saved": [P, L, P, I, I, P, L, P, L, P, P, I, P, F, P, I, I, P, I, I, P, L, I, L, I, F, P],
}
OTHER = {
    "p4c_version": ([], c_int),
    "p4c_graph_replace_memsets": ([P, ctypes.POINTER(c_int), ctypes.POINTER(c_int)], c_int),
    "p4c_last_error": ([], c_char_p),
    "p4c_num_cus": ([], c_int),
    "p4c_loss_workspace_bytes": ([I, I, L, I], c_size_t),
    "p4c_set_side_stream": ([P, P, I], c_int),
    "p4c_ghost_dw_fwd": ([P, P, P, I, I, I, I, P], c_int),
    "p4c_ghost_dw_bwd_data": ([P, P, P, I, I, I, I, P], c_int),
    "p4c_ghost_dw_wgrad_blocks": ([I, I, I], c_int),
    "p4c_ghost_dw_wgrad": ([P, P, P, I, I, I, I, P], c_int),
    "p4c_inorm_blocks": ([L, I], c_int),
    "p4c_inorm_reduce": ([P, P, P, P, P, F, P, I, I, L, I, P], c_int),
    "p4c_inorm_apply": ([P, P, P, P, P, P, P, P, P, P, F, P, P, I, I, L, I, P], c_int),
    "p4c_inorm_apply_mul": ([P, P, P, P, P, P, P, P, P, P, F, P, P, I, I, L, I, P, L, F, P], c_int),
    "p4c_inorm_reduce_mul": ([P, P, P, P, P, F, P, I, I, L, I, P, L, F, P], c_int),
    "p4c_inorm_finalize_fwd": ([P, I, I, L, I, I, P, P, F, P, P, P, P, P], c_int),
    "p4c_inorm_reduce_finalize_fwd": ([P, P, P, P, P, F, P, P, P, P, I, I, L, I, P], c_int),
    "p4c_inorm_reduce_finalize_bwd": ([P, P, P, P, P, F, P, P, P, P, P, P, I, I, L, I, P], c_int),
    "p4c_inorm_finalize_bwd": ([P, I, I, L, I, I, P, P, P, P, P, P, P], c_int),
    "p4c_ts_gram_splits": ([L], c_int),
    "p4c_ts_gram": ([P, I, L, L, L, P, I, L, L, L, P, I, I, L, I, I, P], c_int),
    "p4c_ts_apply_wide_ok": ([I, I, I, I], c_int),
    "p4c_ts_gram_wide_ok": ([I, I, I, I], c_int),
    "p4c_ts_apply_softmax": ([P, L, L, L, P, L, P, L, L, L, I, I, L, I, I, I, P, L, L, L, P], c_int),
    "p4c_ts_apply": ([P, I, L, L, L, P, L, P, I, L, L, L, I, I, L, I, I, I, P], c_int),
    "p4c_ts_apply_mt": ([P, I, L, L, L, P, L, P, I, L, L, L, I, I, L, I, I, I, P], c_int),
    "p4c_ts_apply_mt_ok": ([P, I, L, L, L, P, L, P, I, L, L, L, I, I], c_int),
    "p4c_epa_small_fwd": ([P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P], c_int),
    "p4c_epa_small_bwd": ([P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P], c_int),
    "p4c_ts_gram_norms": ([P, I, L, L, L, P, I, L, L, L, P, I, I, L, I, I, P], c_int),
    "p4c_ts_reduce_splits": ([P, I, I, I, I, I, P, P, P, I, I, P], c_int),
    "p4c_ts_reduce_transpose": ([P, I, L, I, P, I, P], c_int),
    "p4c_ts_colsums": ([I, P, P, P, P, P], c_int),
    "p4c_ts_merge_published": ([P, P, I, L, I, I, I, P], c_int),
    "p4c_prof_enable": ([I, I], c_int),
    "p4c_prof_filter": ([L], c_int),
    "p4c_prof_collect": ([I, L, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double)], c_int),
}

_lib = None

# bumped whenever a kernel updates parameters through raw pointers (FlatAdamW): caches of re-laid parameters key on it as well
# as on tensor._version
PARAM_EPOCH = [0]


def invalidate_param_caches():
    """Tell the per-parameter caches (bf16 casts, re-laid weight images) that parameter storage was written behind autograd's
    back: ``p.data.copy_`` (an EMA / SWA swap), a third-party optimizer that writes through ``.data``, a collective into
    ``p.data``.  ``tensor._version`` does not move for such writes; FlatAdamW, FlatDDP.broadcast_parameters and
    FlatDDP.all_gather_params call this themselves ("Parameter caches")."""
    PARAM_EPOCH[0] += 1


# Gradients written IN PLACE (ops_gemm.GRADS_IN_PLACE, graphlam.GRADS_IN_PLACE: a reduction kernel adds dW into the parameter's .grad
# buffer, autograd gets None) pass no AccumulateGrad node, so no per-parameter hook fires for them.  A gradient exchange that wants to
# start inside the backward (trainer.FlatDDP(overlap=True)) listens here instead:
#   * grad_sink_taken(view)  -- a forward pass took `view` (a region of some parameter's .grad) as the destination of a later backward;
#   * grad_written(*views)   -- the kernels that ADD into these regions have just been enqueued on the current stream (called from
#                               inside the backward pass, after the launch: a stream that waits for the current one sees the sums).
GRAD_SINK_LISTENERS = []


def grad_sink_taken(view):
    for listener in GRAD_SINK_LISTENERS:
        listener.sink_taken(view)


def grad_written(*views):
    if GRAD_SINK_LISTENERS:
        for listener in GRAD_SINK_LISTENERS:
            listener.written(views)


# Values derived from the parameters (re-laid weight images, bf16 copies of weight blocks) are cached per parameter version in
# eager mode.  Inside a HIP-graph capture they must be produced by kernels of THAT graph (a replay has to see the current
# parameters), but once per capture is enough: the capturing code (trainer.GraphedTrainingStep) opens a scope, the ops keep what
# they derived in it -- which also keeps those tensors allocated, hence intact, until the capture ends.
CAPTURE_SCOPE = [None]


def _base(t):
    t = t._base if t._base is not None else t
    return getattr(t, "_p4c_owner", t)      # a rollout's per-step stand-in of a parameter (trainer.RolloutParamProxies): the parameter


def owner_refs(tensors):
    """Weak references to the tensors that own the storage of `tensors` (parameters, for slices of parameters).  A cache keyed by
    addresses and version counters alone can hit a DIFFERENT parameter that the allocator placed at a freed one's address."""
    import weakref

    return tuple(None if t is None else weakref.ref(_base(t)) for t in tensors)


def owners_alive(refs, tensors) -> bool:
    return all((r is None and t is None) or (r is not None and t is not None and r() is _base(t)) for r, t in zip(refs, tensors))


def capture_cache():
    """The dict of the capture in progress, or None (not capturing, or a capture nobody opened a scope for: derive every time)."""
    import torch

    if CAPTURE_SCOPE[0] is not None and torch.cuda.is_current_stream_capturing():
        return CAPTURE_SCOPE[0]
    return None


class P4CError(RuntimeError):
    pass


def _declare(lib, signatures, other):
    for name, argtypes in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, (argtypes, restype) in other.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype


def all_symbols():
    from . import _lib_model  # model-side signatures live next to their wrappers

    names = list(SIGNATURES) + list(OTHER) + list(_lib_model.SIGNATURES) + list(_lib_model.OTHER)
    return names


def lib():
    """Load (once) and return the ctypes handle.  Raises if the extension was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P4CError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)."
            )
        handle = ctypes.CDLL(LIB_PATH)
        _declare(handle, SIGNATURES, OTHER)
        from . import _lib_model

        _declare(handle, _lib_model.SIGNATURES, _lib_model.OTHER)
        _lib = handle
    return _lib


def kernel_sources_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources of this tree (csrc/*.hip, *.cpp, *.hpp, the C header): what a committed
    PMC traffic figure under profiles/ was measured on.  bench.py reports such a figure as `traffic` only when it matches the running
    tree; a changed kernel keeps the pointer to the file and a null number."""
    import glob
    import hashlib

    h = hashlib.sha256()
    root = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.cpp")) + glob.glob(os.path.join(root, "*.hpp")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "py4cast_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(names, key=None):
    """(value, source note) from the first of the profiles/ JSON files `names` that exists: the value only if the file records the
    kernel sources of THIS tree (kernel_sources_sha16), else None with a note that names the file as stale."""
    import json

    root = os.path.dirname(_HERE)
    for name in names:
        f = os.path.join(root, "profiles", name)
        if not os.path.exists(f):
            continue
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        sha = d.get("kernel_sources_sha16")
        val = d
        for k in (key or ()):
            val = val.get(k, {}) if isinstance(val, dict) else None
        if not isinstance(val, (int, float)):
            continue
        if sha == kernel_sources_sha16():
            return val, f"profiles/{name} (committed rocprofv3 --pmc passes on this tree's kernel sources, sha {sha})"
        return None, (f"profiles/{name} holds {val:.4g} B measured on other kernel sources (sha {sha}; this tree: "
                      f"{kernel_sources_sha16()}): not reported as this run's traffic")
    return None, None


DIAG_LIB_PATH = os.path.join(_HERE, "libpy4cast_hip_diag.so")
_diag = None


class use_diagnostic_library:
    """Context manager: route every call of this process to libpy4cast_hip_diag.so (``make -C py4cast_amd/csrc diag``) -- the same
    sources built with -DP4C_DIAG_BUILD, where the P4C_* environment switches that select an older kernel / another geometry are
    live.  The product library reads none of them.  For the A/B parity tests and tools/diagnostics only."""

    def __enter__(self):
        global _lib, _diag
        if not os.path.exists(DIAG_LIB_PATH):
            raise P4CError(f"{DIAG_LIB_PATH} not found: run `make -C py4cast_amd/csrc diag` (or __graft_entry__.build())")
        lib()
        if _diag is None:
            _diag = ctypes.CDLL(DIAG_LIB_PATH)
            _declare(_diag, SIGNATURES, OTHER)
            from . import _lib_model

            _declare(_diag, _lib_model.SIGNATURES, _lib_model.OTHER)
        self._saved = _lib
        _lib = _diag
        invalidate_param_caches()      # derived tensors of the other library's plans are not this one's
        return _diag

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        invalidate_param_caches()
        return False


def diag_switch(name: str):
    """The value of a P4C_* ROUTE switch (an older kernel / a library route as the A/B reference of the native one), or None.  Like the C
    library's diag_env (csrc/common.hpp) these switches exist in DIAGNOSTIC mode only -- inside ``use_diagnostic_library()``, which the
    A/B parity tests (tests/conftest.py::diag_library) and tools/diagnostics enter: the product package reads no route switch from the
    environment, so a deployment cannot differ silently from the measured configuration.  The settings that ARE read from
    the environment are documented ones and bench.py echoes them (config.environment_settings): P4C_NO_AFFINITY, P4C_NO_TUNED_GEMMS,
    P4C_LIB_PATH."""
    if _diag is not None and _lib is _diag:
        return os.environ.get(name)
    return None


def environment_settings():
    """the documented environment settings in force (bench.py: config.environment_settings)"""
    return {k: os.environ[k] for k in ("P4C_NO_AFFINITY", "P4C_NO_TUNED_GEMMS", "P4C_LIB_PATH", "PYTORCH_TUNABLEOP_ENABLED", "MIOPEN_FIND_MODE")
            if k in os.environ}


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().p4c_last_error()
        raise P4CError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


# ---- optional per-entry-point timing with HIP events on the launch stream (bench.py roofline leg)
_TIMED = None


def enable_kernel_timing(names):
    """Record a HIP event pair around every call of the named entry points (None disables)."""
    global _TIMED
    _TIMED = None if names is None else {n: [] for n in names}


def kernel_times():
    """{name: (calls, avg_ms)} for the calls recorded since enable_kernel_timing (synchronises)."""
    out = {}
    if _TIMED:
        torch.cuda.synchronize()
        for n, evs in _TIMED.items():
            if evs:
                out[n] = (len(evs), sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs))
    return out


def kernel_bytes():
    """{name: algorithmic HBM bytes summed over the recorded calls} for the entry points whose wrappers state them."""
    out = {}
    if _TIMED:
        for n, evs in _TIMED.items():
            if evs and all(nb is not None for _, _, nb in evs):
                out[n] = float(sum(nb for _, _, nb in evs))
    return out


# bench.py's step-level roofline of the widened models: with WORK[0] a dict, every native call adds the algorithmic HBM bytes / matrix
# flops its wrapper states ({"bytes": .., "flops": .., "calls": .., "unstated": ..}); None = off (no cost on the product path)
WORK = [None]


def call(name: str, *args, alg_bytes=None, alg_flops=None):
    """alg_bytes / alg_flops: the call's algorithmic HBM bytes and matrix flops (roofline bookkeeping of bench.py; ignored unless
    timing / work counting is enabled)."""
    w = WORK[0]
    if w is not None:
        w["calls"] += 1
        if alg_bytes is None:
            w["unstated"] += 1
        else:
            w["bytes"] += alg_bytes
        if alg_flops:
            w["flops"] += alg_flops
    if _TIMED is not None and name in _TIMED:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()  # torch's current stream == the stream handed to the library
        check(getattr(lib(), name)(*args), name)
        b.record()
        _TIMED[name].append((a, b, alg_bytes))
        return
    check(getattr(lib(), name)(*args), name)


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def stream(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise P4CError(f"unsupported dtype {dt}")


_TUNED_GEMMS_LOADED = [False]


def _load_tuned_gemms():
    """Once per process, at the first native call on a GPU tensor: hand the shipped GEMM selections (package __init__) to TunableOp."""
    _TUNED_GEMMS_LOADED[0] = True
    path = os.environ.get("P4C_TUNED_GEMMS_FILE")
    if not path:
        return
    try:
        import torch.cuda.tunable as tunable

        # lookup on, tuning and result files off: only the shipped selections are consulted, nothing is measured or written,
        # and a stray tunableop_results*.csv in the working directory is not picked up by a file-name default
        tunable.enable(True)
        tunable.tuning_enable(False)
        if hasattr(tunable, "write_file_on_exit"):
            tunable.write_file_on_exit(False)
        ok = tunable.read_file(path)
        if ok is False:   # validator lines of the file (torch / ROCm / hipBLASLt / GPU) do not match this stack
            import warnings

            warnings.warn("py4cast_amd: the shipped GEMM selections were rejected by TunableOp's validators (another torch / "
                          "ROCm / hipBLASLt build): library-default GEMM kernels are used")
    except Exception as exc:  # noqa: BLE001  (no TunableOp in this build / unreadable file: the library's default selection stays)
        import warnings

        warnings.warn(f"py4cast_amd: tuned GEMM selections not loaded ({type(exc).__name__}: {exc})")


def require_cuda(*tensors):
    if not _TUNED_GEMMS_LOADED[0] and any(t is not None and t.is_cuda for t in tensors):
        _load_tuned_gemms()
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise P4CError(
                "py4cast_amd kernels run on the GPU only (no CPU fallback); got a tensor on " + str(t.device)
            )
