"""Batched device-side solver: thin Python host above the C ABI (include/kmpc.h).

torch is used only for device memory and streams.  One BatchMPC = one handle = one GPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import KMPC_F32, KMPC_F64, Config

DEFAULT_WEIGHTS = (9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0)  # MKZMPCPathFollower.jl:51-59


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


class _Out(dict):
    """output dict of a solve call; remembers which tensor objects were validated for which (B, dtype, ...) so that handing it back as `out=`
    skips the checks (the entries themselves are only tensors: callers iterate over them)"""
    __slots__ = ("fits", "refs", "ptrs")


class BatchMPC:
    """Solves B independent kinematic-bicycle MPC problems per call on one MI355X.

    Arguments mirror the reference module (scripts/mpc_utils/MKZMPCPathFollower.jl):
    N, dt, ... are its constants (:28-48); `weights` is update_cost()'s argument list (:158-169).
    """

    def __init__(self, N=8, dtype=torch.float64, device=None, weights=None, **options):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("BatchMPC needs an MI355X (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.dtype = dtype
        self.cfg = Config()
        _lib.check(self.lib.kmpc_config_default(C.byref(self.cfg), int(N), KMPC_F64 if dtype == torch.float64 else KMPC_F32))
        for k, v in options.items():
            if not hasattr(self.cfg, k):
                raise TypeError("unknown option %r" % k)
            setattr(self.cfg, k, v)
        self.N = int(N)
        h = C.c_void_p()
        _lib.check(self.lib.kmpc_create(C.byref(self.cfg), self.device.index, C.byref(h)))
        self.h = h
        if weights is not None:
            self.update_cost(*weights)

    def close(self):
        if getattr(self, "h", None):
            self.lib.kmpc_destroy(self.h)
            self.h = None

    __del__ = close

    def update_cost(self, cx, cy, cp, cv, cda, cdd, ca, cd):
        """update_cost of MKZMPCPathFollower.jl:158-169 (same argument order)."""
        w = (C.c_double * 8)(cx, cy, cp, cv, cda, cdd, ca, cd)
        _lib.check(self.lib.kmpc_set_cost(self.h, w), self.h)

    def _dev(self, a, shape):
        # (a device tensor of the right type passes straight through: the B = 1 latency path spends its time here otherwise)
        if type(a) is torch.Tensor and a.dtype == self.dtype and a.device == self.device and a.is_contiguous():
            t = a
        else:
            t = torch.as_tensor(a, dtype=self.dtype, device=self.device).contiguous()
        if t.shape != shape:
            raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(t.shape)))
        return t

    def _outputs(self, out, B, want_U, want_X):
        """output tensors of one call: those in `out` are reused when they fit (B, N, dtype, device), anything else is (re)allocated --
        a buffer left over from a call with another batch size, horizon or element type must never reach the kernel as a raw pointer"""
        N = self.N
        key = (B, N, want_U, want_X, self.dtype, self.device)  # U [B,N,2] and X [B,N+1,4] depend on the horizon
        if type(out) is _Out and out.fits == key and all(out.get(k) is t for k, t in out.refs):
            return out  # the dict of the previous call with the very same tensor objects: checked then
        o = _Out(out) if out is not None else _Out()
        spec = {"u0": ((B, 2), self.dtype), "status": ((B,), torch.int32), "cost": ((B,), self.dtype), "viol": ((B,), self.dtype),
                "iters": ((B,), torch.int32)}
        if want_U:
            spec["U"] = ((B, N, 2), self.dtype)
        if want_X:
            spec["X"] = ((B, N + 1, 4), self.dtype)
        for k, (shape, dt) in spec.items():
            t = o.get(k)
            if not (isinstance(t, torch.Tensor) and tuple(t.shape) == shape and t.dtype == dt and t.device == self.device and t.is_contiguous()):
                o[k] = torch.empty(shape, dtype=dt, device=self.device)
        o.fits = key
        o.refs = [(k, o[k]) for k in spec]
        o.ptrs = tuple(_ptr(o.get(k)) for k in ("u0", "status", "cost", "viol", "iters", "U", "X"))  # valid while `refs` match
        return o

    def _solve(self, entry, z0, second, second_shape, v_target, u_prev, warm_U, warm, want_U, want_X, out):
        N = self.N
        z0 = self._dev(z0, (len(z0), 4))
        B = z0.shape[0]
        second = self._dev(second, (B,) + second_shape)
        v_target = self._dev(v_target, (B,))
        u_prev = self._dev(u_prev, (B, 2))
        if warm_U is not None:
            warm_U = self._dev(warm_U, (B, N, 2))
        o = self._outputs(out, B, want_U, want_X)
        stream = C.c_void_p(_raw_stream(self.device.index))
        pu0, pst, pco, pvi, pit, pU, pX = o.ptrs
        rc = entry(self.h, B, _ptr(z0), _ptr(second), _ptr(v_target), _ptr(u_prev),
                   _ptr(warm_U), 1 if (warm and warm_U is not None) else 0,
                   pu0, pst, pco, pvi, pit, pU if want_U else None, pX if want_X else None, stream)
        _lib.check(rc, self.h)
        if warm_U is not None:
            o["warm_U"] = warm_U
        return o

    def solve(self, z0, ref, v_target, u_prev, warm_U=None, warm=False, want_U=False, want_X=False, out=None):
        """z0[B,4], ref[B,N+1,3], v_target[B], u_prev[B,2] (acc, steer) -> dict of device tensors.

        Asynchronous on torch's current stream.  `out` may carry the output dict of a previous call: its tensors are
        reused when batch size, dtype and device match (no allocation in the timed path), otherwise replaced.
        """
        return self._solve(self.lib.kmpc_solve_batch, z0, ref, (self.N + 1, 3), v_target, u_prev, warm_U, warm, want_U, want_X, out)

    def solve_frenet(self, z0, k_poly, v_target, u_prev, warm_U=None, warm=False, want_U=False, want_X=False, out=None):
        """Frenet-frame variant (handle created with model=1): z0[B,4] = (s, e_y, e_psi, v), k_poly[B,4] curvature polynomial,
        highest degree first; X[B,N+1,4] = (s, e_y, e_psi, v).  Otherwise as solve()."""
        return self._solve(self.lib.kmpc_solve_batch_frenet, z0, k_poly, (4,), v_target, u_prev, warm_U, warm, want_U, want_X, out)

    # ---- packed records (ABI v8, kmpc_solve_batch_packed; include/kmpc.h) -------------------------
    def record_scalars(self):
        """scalars per input record (kmpc_record_bytes / element size): z0[4], v_target, u_prev[2], pad, ref[(N+1)*3], zero padding to whole 64-B lines"""
        return int(self.lib.kmpc_record_bytes(self.N, self.cfg.dtype)) // self.dtype.itemsize

    def pack(self, z0, ref, v_target, u_prev, out=None):
        """the four input arrays of solve() -> records [B, record_scalars()] (device kernel kmpc_pack_records)"""
        z0 = self._dev(z0, (len(z0), 4))
        B = z0.shape[0]
        ref = self._dev(ref, (B, self.N + 1, 3))
        v_target = self._dev(v_target, (B,))
        u_prev = self._dev(u_prev, (B, 2))
        rs = self.record_scalars()
        rec = out if (out is not None and tuple(out.shape) == (B, rs) and out.dtype == self.dtype and out.device == self.device) else \
            torch.empty((B, rs), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.kmpc_pack_records(self.h, B, _ptr(z0), _ptr(ref), _ptr(v_target), _ptr(u_prev), _ptr(rec),
                                              C.c_void_p(_raw_stream(self.device.index))), self.h)
        return rec

    def solve_packed(self, records, warm_U=None, warm=False, want_U=False, want_X=False, out=None):
        """records [B, record_scalars()] -> dict with `orec` (the [B, 64 bytes] output records) and views into it: u0 [B,2], cost [B], viol [B] (strided views),
        status [B], iters [B] (int32 views); U / X as in solve().  Asynchronous on torch's current stream."""
        N, rs = self.N, self.record_scalars()
        if not (records.dtype == self.dtype and records.device == self.device and records.is_contiguous() and records.dim() == 2 and records.shape[1] == rs):
            raise ValueError("records: expected a contiguous [B, %d] %s tensor on %s" % (rs, self.dtype, self.device))
        B = records.shape[0]
        per = 64 // self.dtype.itemsize
        o = out if (out is not None and out.get("orec") is not None and tuple(out["orec"].shape) == (B, per)) else {}
        if "orec" not in o:
            o["orec"] = torch.zeros((B, per), dtype=self.dtype, device=self.device)
            o["u0"], o["cost"], o["viol"] = o["orec"][:, 0:2], o["orec"][:, 2], o["orec"][:, 3]
            ints = o["orec"].view(torch.int32).view(B, 64 // 4)
            k = 4 * self.dtype.itemsize // 4
            o["status"], o["iters"] = ints[:, k], ints[:, k + 1]
        if want_U and o.get("U") is None:
            o["U"] = torch.empty((B, N, 2), dtype=self.dtype, device=self.device)
        if want_X and o.get("X") is None:
            o["X"] = torch.empty((B, N + 1, 4), dtype=self.dtype, device=self.device)
        if warm_U is not None:
            warm_U = self._dev(warm_U, (B, N, 2))
        _lib.check(self.lib.kmpc_solve_batch_packed(self.h, B, _ptr(records), _ptr(warm_U), 1 if (warm and warm_U is not None) else 0, _ptr(o["orec"]),
                                                    _ptr(o["U"]) if want_U else None, _ptr(o["X"]) if want_X else None,
                                                    C.c_void_p(_raw_stream(self.device.index))), self.h)
        if warm_U is not None:
            o["warm_U"] = warm_U
        return o

    # ---- diagnostics for tests ------------------------------------------------------------------
    def debug_condense(self, z0, ref, v_target, U, hessian=1):
        N = self.N
        z0 = self._dev(z0, (len(z0), 4))
        B = z0.shape[0]
        ref = self._dev(ref, (B, N + 1, 3))
        v_target = self._dev(v_target, (B,))
        U = self._dev(U, (B, N, 2))
        H = torch.zeros((B, 2 * N, 2 * N), dtype=self.dtype, device=self.device)
        g = torch.zeros((B, 2 * N), dtype=self.dtype, device=self.device)
        J = torch.zeros((B,), dtype=self.dtype, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.kmpc_debug_condense(self.h, B, _ptr(z0), _ptr(ref), _ptr(v_target), _ptr(U),
                                                int(hessian), _ptr(H), _ptr(g), _ptr(J), stream), self.h)
        return H, g, J

    def debug_kkt(self, z0, ref, v_target, u_prev, U, w, b, sc=1.0, reg=0.0, hessian=1):
        """kmpc_debug_kkt: K = sc*H(U) + A^T diag(w) A + reg*I as the solve kernel assembles it, gradient, K^-1 (b - sc*g), PD flag"""
        N = self.N
        z0 = self._dev(z0, (len(z0), 4))
        B = z0.shape[0]
        ref = self._dev(ref, (B, N + 1, 3))
        v_target = self._dev(v_target, (B,))
        u_prev = self._dev(u_prev, (B, 2))
        U = self._dev(U, (B, N, 2))
        w = self._dev(w, (B, 5 * N - 2))
        b = self._dev(b, (B, 2 * N))
        K = torch.zeros((B, 2 * N, 2 * N), dtype=self.dtype, device=self.device)
        g = torch.zeros((B, 2 * N), dtype=self.dtype, device=self.device)
        x = torch.zeros((B, 2 * N), dtype=self.dtype, device=self.device)
        ok = torch.zeros((B,), dtype=torch.int32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.kmpc_debug_kkt(self.h, B, _ptr(z0), _ptr(ref), _ptr(v_target), _ptr(u_prev), _ptr(U), _ptr(w), _ptr(b),
                                           float(sc), float(reg), int(hessian), _ptr(K), _ptr(g), _ptr(x), _ptr(ok), stream), self.h)
        return K, g, x, ok

    def debug_mfma_probe(self, a, b):
        a = self._dev(a, (64,))
        b = self._dev(b, (64,))
        d = torch.zeros((64, 4), dtype=self.dtype, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.kmpc_debug_mfma_probe(self.h, _ptr(a), _ptr(b), _ptr(d), stream), self.h)
        return d


def solve_host(N, z0, ref, v_target, u_prev, weights=None, dtype=np.float64, warm_U=None, device=0, **options):
    """Host-pointer convenience path (kmpc_solve_batch_host): numpy in, numpy out."""
    lib = _lib.load()
    cfg = Config()
    _lib.check(lib.kmpc_config_default(C.byref(cfg), int(N), KMPC_F64 if dtype == np.float64 else KMPC_F32))
    for k, v in options.items():
        if not hasattr(cfg, k):
            raise TypeError("unknown option %r" % k)
        setattr(cfg, k, v)
    h = C.c_void_p()
    _lib.check(lib.kmpc_create(C.byref(cfg), device, C.byref(h)))
    try:
        if weights is not None:
            _lib.check(lib.kmpc_set_cost(h, (C.c_double * 8)(*weights)), h)
        z0 = np.ascontiguousarray(z0, dtype=dtype).reshape(-1, 4)
        B = z0.shape[0]
        ref = np.ascontiguousarray(ref, dtype=dtype).reshape(B, N + 1, 3)
        vt = np.ascontiguousarray(v_target, dtype=dtype).reshape(B)
        up = np.ascontiguousarray(u_prev, dtype=dtype).reshape(B, 2)
        u0 = np.empty((B, 2), dtype)
        st = np.empty(B, np.int32)
        cost = np.empty(B, dtype)
        viol = np.empty(B, dtype)
        iters = np.empty(B, np.int32)
        U = np.empty((B, N, 2), dtype)
        X = np.empty((B, N + 1, 4), dtype)
        wu = None
        if warm_U is not None:
            wu = np.ascontiguousarray(warm_U, dtype=dtype).reshape(B, N, 2).copy()
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        _lib.check(lib.kmpc_solve_batch_host(h, B, p(z0), p(ref), p(vt), p(up), p(wu), 1 if wu is not None else 0,
                                             p(u0), p(st), p(cost), p(viol), p(iters), p(U), p(X)), h)
        return dict(u0=u0, status=st, cost=cost, viol=viol, iters=iters, U=U, X=X)
    finally:
        lib.kmpc_destroy(h)
