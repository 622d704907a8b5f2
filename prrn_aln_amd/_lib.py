"""Loader for libg2g.so.  There is no fallback: if the HIP library is missing or no MI355X is usable the
product raises -- nothing here (or anywhere in the package) computes an alignment on the CPU."""
from __future__ import annotations

import ctypes as C
import os

from . import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("G2G_LIB") or os.path.join(HERE, "libg2g.so")      # (G2G_LIB: an experiment's build of the same library)

_lib = None


class G2GError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise G2GError("libg2g.so is not built (python -m prrn_aln_amd.build); there is no CPU fallback")
    # the persistent launches of a sweep run side by side on streams of their own; HIP multiplexes streams onto this many
    # hardware queues (4 by default) -- read when the HIP runtime initialises, so it only helps if nothing has touched the GPU yet
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    L = C.CDLL(LIB)
    L.g2g_create.restype = C.c_void_p
    L.g2g_create.argtypes = [C.c_int]
    L.g2g_destroy.argtypes = [C.c_void_p]
    L.g2g_last_error.restype = C.c_char_p
    L.g2g_device_ok.argtypes = [C.c_void_p]
    L.g2g_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.g2g_reset_options.argtypes = [C.c_void_p]
    L.g2g_get_option.restype = C.c_char_p
    L.g2g_get_option.argtypes = [C.c_void_p, C.c_char_p]
    PP = C.POINTER(C.POINTER(_abi.Problem))
    L.g2g_forward_batch.argtypes = [C.c_void_p, C.c_int, PP, C.POINTER(_abi.Result)]
    L.g2g_batch_prepare.argtypes = [C.c_void_p, C.c_int, PP, C.POINTER(C.c_void_p)]
    L.g2g_batch_run.argtypes = [C.c_void_p]
    L.g2g_batch_fetch.argtypes = [C.c_void_p, C.POINTER(_abi.Result)]
    L.g2g_batch_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.g2g_batch_cells.restype = C.c_longlong
    L.g2g_batch_cells.argtypes = [C.c_void_p]
    L.g2g_batch_arena_bytes.restype = C.c_size_t
    L.g2g_batch_arena_bytes.argtypes = [C.c_void_p]
    L.g2g_batch_free.argtypes = [C.c_void_p]
    L.g2g_batch_recovery.restype = None
    L.g2g_batch_recovery.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.g2g_ctx_counters.restype = None
    L.g2g_ctx_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.g2g_ctx_last_timeout.restype = C.c_char_p
    L.g2g_ctx_last_timeout.argtypes = [C.c_void_p]
    L.g2g_align2_score_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(_abi.Skl)), C.POINTER(C.c_int), C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(C.POINTER(_abi.Skl)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(_abi.Fstat), C.POINTER(_abi.Fstat)]
    L.g2g_process_counters.restype = None
    L.g2g_process_counters.argtypes = [C.POINTER(C.c_longlong)]
    L.g2g_process_last_timeout.restype = C.c_size_t
    L.g2g_process_last_timeout.argtypes = [C.c_char_p, C.c_size_t]
    L.g2g_ctx_mem_counters.restype = None
    L.g2g_ctx_mem_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.g2g_ctx_wait_gaps.restype = None
    L.g2g_ctx_wait_gaps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
    L.g2g_batch_spscore.argtypes = [C.c_void_p, C.POINTER(_abi.SpParams), C.POINTER(C.POINTER(_abi.Skl)), C.POINTER(C.c_int), C.POINTER(_abi.Fstat)]
    L.g2g_free.argtypes = [C.c_void_p]
    L.g2g_stdskl.restype = C.POINTER(_abi.Skl)
    L.g2g_stdskl.argtypes = [C.POINTER(_abi.Skl), C.c_int, C.POINTER(C.c_int)]
    L.g2g_alnscored_batch.argtypes = [C.c_void_p, C.POINTER(_abi.Params), C.c_int, C.POINTER(_abi.DSeq), C.c_int,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), _abi.c_f64p, C.POINTER(C.c_int32)]
    L.g2g_alignb_ng_batch.argtypes = [C.c_void_p, C.POINTER(_abi.Params), C.c_int, C.POINTER(_abi.DSeq), C.c_int,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), _abi.c_f64p, C.POINTER(C.POINTER(_abi.Skl)),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int32)]
    L.g2g_refine.argtypes = [C.c_void_p, C.POINTER(_abi.Params), C.c_int, C.c_int, _abi.c_u8p, C.POINTER(_abi.Tree),
                             C.POINTER(_abi.RefineOpts), C.POINTER(_abi.c_u8p), C.POINTER(C.c_int),
                             C.POINTER(C.POINTER(_abi.RefineStep)), C.POINTER(C.c_int), C.POINTER(_abi.RefineStats)]
    L.g2g_pairsum.argtypes = [C.c_void_p, C.POINTER(_abi.Params), C.c_int, C.c_int, _abi.c_u8p, C.POINTER(_abi.Tree), C.c_int,
                              C.POINTER(C.c_double)]
    bind_level1(L)
    _lib = L
    return L


def last_error() -> str:
    return lib().g2g_last_error().decode()


def bind_level1(L):
    L.g2g_group_create.restype = C.c_void_p
    L.g2g_group_create.argtypes = [C.c_void_p, C.POINTER(_abi.Params), C.c_int, C.c_int, _abi.c_u8p, _abi.c_f64p]
    L.g2g_group_free.argtypes = [C.c_void_p]
    L.g2g_pwdm_create.restype = C.c_void_p
    L.g2g_pwdm_create.argtypes = [C.c_void_p, C.POINTER(_abi.Params), C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.g2g_pwdm_create_batch.argtypes = [C.c_void_p, C.POINTER(_abi.Params), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    L.g2g_pwdm_free.argtypes = [C.c_void_p]
    L.g2g_pwdm_problem.restype = C.POINTER(_abi.Problem)
    L.g2g_pwdm_problem.argtypes = [C.c_void_p]
    L.g2g_align2.argtypes = [C.c_void_p, C.c_void_p, _abi.c_f64p, C.POINTER(C.POINTER(_abi.Skl)), C.POINTER(C.c_int)]
    L.g2g_homscore.argtypes = [C.c_void_p, C.c_void_p, _abi.c_f64p, C.POINTER(C.c_int64)]
    L.g2g_pwdm_spparams.argtypes = [C.c_void_p, C.POINTER(_abi.SpParams)]
    L.g2g_spscore_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(_abi.Skl)), C.POINTER(C.c_int), C.POINTER(_abi.Fstat)]
    L.g2g_align2_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), _abi.c_f64p,
                                   C.POINTER(C.POINTER(_abi.Skl)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
