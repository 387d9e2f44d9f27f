"""ctypes binding of the C ABI in include/kmpc.h (libkmpc_hip.so, built for gfx950).

There is no CPU fallback: if the shared library is missing or cannot be loaded this module
raises, and every compute entry point fails when no MI355X is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkmpc_hip.so")

KMPC_F64, KMPC_F32 = 0, 1
STATUS_NAMES = {0: "Optimal", 1: "UserLimit", 2: "Infeasible", 3: "Error"}  # cf. JuMP status symbols


class Config(C.Structure):
    """struct kmpc_config (include/kmpc.h) = constants of MKZMPCPathFollower.jl:28-48 + solver options."""
    _fields_ = [("N", C.c_int32), ("dtype", C.c_int32),
                ("dt", C.c_double), ("dt_control", C.c_double), ("L_a", C.c_double), ("L_b", C.c_double),
                ("steer_max", C.c_double), ("steer_dmax", C.c_double), ("a_max", C.c_double),
                ("a_dmax", C.c_double), ("v_min", C.c_double), ("v_max", C.c_double),
                ("max_iter", C.c_int32), ("hessian", C.c_int32), ("tol", C.c_double),
                ("mu_init", C.c_double), ("bound_relax", C.c_double), ("warm_push", C.c_double),
                ("warm_mu", C.c_double), ("max_ls", C.c_int32), ("kernel_variant", C.c_int32),
                ("mu_strategy", C.c_int32), ("indef_strategy", C.c_int32), ("schedule", C.c_int32), ("model", C.c_int32),
                ("start", C.c_int32)]


EXPORTS = ["kmpc_abi_version", "kmpc_config_default", "kmpc_create", "kmpc_destroy", "kmpc_set_cost",
           "kmpc_get_cost", "kmpc_solve_batch", "kmpc_solve_batch_host", "kmpc_last_error",
           "kmpc_debug_condense", "kmpc_debug_mfma_probe",
           "kmpc_path_create", "kmpc_path_destroy", "kmpc_waypoints_batch", "kmpc_path_last_error",
           "kmpc_sim_advance_batch", "kmpc_solve_batch_frenet", "kmpc_debug_kkt", "kmpc_command_batch",
           "kmpc_record_bytes", "kmpc_pack_records", "kmpc_solve_batch_packed"]

_lib = None


def load():
    """Load libkmpc_hip.so; raise (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    L.kmpc_abi_version.restype = i32
    L.kmpc_config_default.argtypes = [C.POINTER(Config), i32, i32]
    L.kmpc_create.argtypes = [C.POINTER(Config), i32, C.POINTER(vp)]
    L.kmpc_destroy.argtypes = [vp]
    L.kmpc_set_cost.argtypes = [vp, C.POINTER(C.c_double)]
    L.kmpc_get_cost.argtypes = [vp, C.POINTER(C.c_double)]
    sig = [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.kmpc_solve_batch.argtypes = sig + [vp]
    L.kmpc_solve_batch_host.argtypes = sig
    L.kmpc_solve_batch_frenet.argtypes = sig + [vp]
    L.kmpc_last_error.argtypes = [vp]
    L.kmpc_last_error.restype = C.c_char_p
    L.kmpc_debug_condense.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.kmpc_debug_mfma_probe.argtypes = [vp, vp, vp, vp, vp]
    L.kmpc_debug_kkt.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, C.c_double, C.c_double, i32, vp, vp, vp, vp, vp]
    dp = C.POINTER(C.c_double)
    L.kmpc_path_create.argtypes = [i32, i32, dp, dp, dp, dp, dp, C.POINTER(vp)]
    L.kmpc_path_destroy.argtypes = [vp]
    L.kmpc_waypoints_batch.argtypes = [vp, i32, i32, C.c_double, vp, vp, vp, vp, vp, vp]
    L.kmpc_path_last_error.argtypes = [vp]
    L.kmpc_path_last_error.restype = C.c_char_p
    L.kmpc_sim_advance_batch.argtypes = [i32, i32, vp, vp, i32, vp]
    L.kmpc_command_batch.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp]
    L.kmpc_record_bytes.argtypes = [i32, i32]
    L.kmpc_record_bytes.restype = C.c_int64
    L.kmpc_pack_records.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.kmpc_solve_batch_packed.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp]
    for name in EXPORTS:
        getattr(L, name)
    _lib = L
    return L


class KmpcError(RuntimeError):
    pass


def check(rc, handle=None):
    if rc != 0:
        msg = load().kmpc_last_error(handle)
        raise KmpcError("kmpc error %d: %s" % (rc, msg.decode() if msg else "?"))
