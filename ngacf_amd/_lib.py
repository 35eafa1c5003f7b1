"""ctypes binding of the C ABI declared in include/ngacf.h.

There is no CPU fallback: if libngacf_hip.so is missing the import of any product
module fails here, loudly.  PyTorch only provides device memory and the stream.
"""
import ctypes
import os

import torch  # noqa: F401  -- must come first: the library has to bind to the HIP runtime torch already
              # loaded (torch bundles its own libamdhip64); two runtimes in one process cannot share pointers

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NGACF_LIB_PATH: another build of the same library, e.g. an instrumented variant for an A/B measurement -- never a fallback:
#  a missing file fails the import just the same)
LIB_PATH = os.environ.get("NGACF_LIB_PATH") or os.path.join(_HERE, "lib", "libngacf_hip.so")

F32, BF16 = 0, 1
FLAG_ELU, FLAG_RESIDUAL, FLAG_SCORE_ROW, FLAG_SKIP_NODE, FLAG_GOUT_F32 = 1, 2, 4, 8, 16
FLAG_ROOF = (0, 32, 64)      # NGACF_ROOF_ADD / NGACF_ROOF_XOR: the measurement-only instantiations of the traversal kernels
ADAM_MAX_TENSORS = 24            # NGACF_ADAM_MAX_TENSORS
TICK_MAX = 48                      # NGACF_TICK_MAX
TBWD_BLOCKS = 1024
MAX_PACK = 8                       # NGACF_MAX_PACK

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_u64 = ctypes.c_uint64

# name -> argtypes, exactly the prototypes of include/ngacf.h
SIGNATURES = {
    "ngacf_transform_fwd": [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _f, _u64, _p, _p],
    "ngacf_prop_fwd": [_p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                       _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _i, _p],
    "ngacf_prop_fwd_next": [_p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                            _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _i, _p, _p],
    "ngacf_transform_fwd_pack": [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _f, _u64, _p, _i, _p, _p, _p, _p, _p, _p],
    "ngacf_prop_bwd": [_p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p,
                       _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _i, _p],
    "ngacf_prop_bwd_dx": [_p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p,
                          _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _u64, _p, _i, _p, _p],
    "ngacf_transform_dw_batch": [_p, _i, _p, _i, _i, _i, _p, _p],
    "ngacf_transform_dx_node": [_p, _p, _p, _i, _i, _f, _u64, _p, _p, _p],
    "ngacf_adam_step": [_p, _p, _p, _p, _p, _p, _p, _i, _f, _f, _f, _f, _f, _p],
    "ngacf_bpr_fwd_bwd": [_p, _p, _p, _p, _p, _i, _p],
    "ngacf_tick": [_p, _p, _i, _p, _i, _p],
    "ngacf_mark_rows": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ngacf_transform_bwd": [_p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _u64, _p, _p],
    "ngacf_transform_fwd_rows": [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _p, _p],
    "ngacf_transform_bwd_rows": [_p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                 _f, _u64, _p, _p],
    "ngacf_prop_bwd_node": [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _p],
    "ngacf_score_fwd": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ngacf_score_bwd": [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ngacf_bpr_fwd": [_p, _p, _p, _i, _p],
    "ngacf_bpr_bwd": [_p, _p, _p, _p, _p, _i, _p],
    "ngacf_spmm_fwd": [_p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p],
    "ngacf_sample_pairs": [_p, ctypes.c_longlong, ctypes.c_longlong, _p, _p, _i, _p, _p, _i, _u64, _p, _p, _p, _p, _p, _i, _p],
    "ngacf_eval_topk": [_p, _i, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p],
    "ngacf_dropout_bits": [_i, _u64, _p, _p, _p, _p, _i, _p],
    "ngacf_shard_rows": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "ngacf_rows_scatter": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ngacf_rows_scatter_scored": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ngacf_timing_begin": [],
    "ngacf_timing_end": [_p, _p, _i],
    "ngacf_peer_alloc": [_u64, _p], "ngacf_peer_alloc_ctrl": [_u64, _p], "ngacf_peer_free": [_p], "ngacf_peer_export": [_p, _p], "ngacf_peer_import": [_p, _p],
    "ngacf_peer_close": [_p], "ngacf_peer_push": [_p, _i, _i, _p, _p, _p, _p], "ngacf_peer_wait": [_p, _i, _i, _p, _p, _u64, _p],
    "ngacf_mha_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _p, _p],
    "ngacf_mha_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _p, _p],
}


class NextStage(ctypes.Structure):
    """ngacf_next_stage of include/ngacf.h"""
    _fields_ = [("panel", _p), ("avec", _p), ("H", _p), ("S", _p), ("dout", ctypes.c_int32), ("heads", ctypes.c_int32),
                ("feat_drop_p", _f), ("seed", _u64)]


DW_BATCH_MAX = 6                   # NGACF_DW_BATCH_MAX


class DwJob(ctypes.Structure):
    """ngacf_dw_job of include/ngacf.h"""
    _fields_ = [("X", _p), ("H", _p), ("dH", _p), ("dS", _p), ("dW", _p), ("davec", _p), ("x_dtype", ctypes.c_int32), ("h_dtype", ctypes.c_int32),
                ("din", ctypes.c_int32), ("dout", ctypes.c_int32), ("heads", ctypes.c_int32), ("feat_drop_p", _f), ("seed", _u64)]


class DxPrev(ctypes.Structure):
    """ngacf_dx_prev of include/ngacf.h"""
    _fields_ = [("out", _p), ("agg", _p), ("R", _p), ("S", _p), ("Gn", _p), ("ST", _p), ("dS", _p), ("alpha", _f), ("flags", ctypes.c_int32)]


class BwdDx(ctypes.Structure):
    """ngacf_bwd_dx of include/ngacf.h"""
    _fields_ = [("panel", _p), ("dX", _p), ("din", ctypes.c_int32), ("dx_dtype", ctypes.c_int32), ("feat_drop_p", _f), ("seed", _u64)]


MAX_PEERS, PEER_MAX_REGIONS, PEER_HANDLE_BYTES = 15, 2, 64       # NGACF_MAX_PEERS, NGACF_PEER_MAX_REGIONS, NGACF_PEER_HANDLE_BYTES


class PeerRegion(ctypes.Structure):
    """ngacf_peer_region of include/ngacf.h"""
    _fields_ = [("src", _p), ("bytes", _u64), ("dst", _p * MAX_PEERS)]


class NgacfError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ngacf_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.ngacf_version.restype = ctypes.c_char_p
    lib.ngacf_version.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = ctypes.c_int
        fn.argtypes = args
    return lib


lib = _load()


def version():
    return lib.ngacf_version().decode()


_ERRS = {-1: "NGACF_E_ARG (null pointer or bad size)", -2: "NGACF_E_SHAPE (unsupported d/heads/dtype)"}


def check(rc, what):
    if rc != 0:
        raise NgacfError("%s failed: %s" % (what, _ERRS.get(rc, "hipError_t %d" % rc)))
