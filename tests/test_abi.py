"""The C-ABI library loads on a CPU-only box and exports every symbol include/kmpc.h declares.
No compute calls are made here; creating a handle must fail loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "kmpc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kmpc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mkz_mpc_path_follower_amd import _lib
    L = _lib.load()
    names = _declared()
    assert len(names) >= 11
    for n in names:
        assert hasattr(L, n), "libkmpc_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names
    assert L.kmpc_abi_version() == 8
    assert L.kmpc_record_bytes(20, _lib.KMPC_F64) == 576 and L.kmpc_record_bytes(20, _lib.KMPC_F32) == 320 and L.kmpc_record_bytes(8, _lib.KMPC_F64) == 320   # 72 scalars at N = 20
    assert L.kmpc_record_bytes(100, _lib.KMPC_F64) == -1


def test_config_defaults_are_the_reference_constants():
    """MKZMPCPathFollower.jl:28-48"""
    from mkz_mpc_path_follower_amd import _lib
    L = _lib.load()
    c = _lib.Config()
    assert L.kmpc_config_default(C.byref(c), 8, _lib.KMPC_F64) == 0
    assert (c.N, c.dt, c.dt_control, c.L_a, c.L_b) == (8, 0.20, 0.10, 1.108, 1.742)
    assert (c.steer_max, c.steer_dmax, c.a_max, c.a_dmax, c.v_min, c.v_max) == (0.5, 0.5, 1.0, 1.5, 0.0, 20.0)
    assert c.tol == 1e-8 and c.bound_relax == 1e-8  # Ipopt defaults
    assert c.mu_init == 1.0  # deliberately not Ipopt's 0.1


def test_create_fails_loudly_without_gpu_or_with_bad_config():
    import torch
    from mkz_mpc_path_follower_amd import _lib
    L = _lib.load()
    c = _lib.Config()
    L.kmpc_config_default(C.byref(c), 100, _lib.KMPC_F64)  # horizon out of range
    h = C.c_void_p()
    assert L.kmpc_create(C.byref(c), 0, C.byref(h)) == -1
    assert b"horizon" in L.kmpc_last_error(None)
    if not torch.cuda.is_available():
        L.kmpc_config_default(C.byref(c), 8, _lib.KMPC_F64)
        rc = L.kmpc_create(C.byref(c), 0, C.byref(h))
        assert rc in (-2, -3) and L.kmpc_last_error(None)
        from mkz_mpc_path_follower_amd import BatchMPC
        with pytest.raises(RuntimeError):
            BatchMPC(N=8)  # no CPU fallback


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mkz_mpc_path_follower_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for bad in ("import oracle", "from oracle", "libkmpc_oracle", "oracle/", "kmpc_condensed_solve", "kmpc_nlp.h"):
                    assert bad not in src, (f, bad)


def test_makefile_lists_every_included_header():
    """An object file must be rebuilt when a header it includes changes: the generic kernel once kept running an older kmpc_ipm.h because
    its Makefile rule did not list it (round 3; caught by the kernels' iteration counts parting from the CPU checker's)."""
    csrc = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()

    def includes(path, seen):
        for inc in re.findall(r'#include\s+"([^"]+)"', open(path).read()):
            f = os.path.normpath(os.path.join(os.path.dirname(path), inc))
            if f not in seen and os.path.exists(f):
                seen.add(f)
                includes(f, seen)
        return seen

    rules = dict(re.findall(r"^(kmpc_\w+\.o):\s*(.*)$", mk, flags=re.M))
    assert len(rules) >= 8
    for obj, deps in rules.items():
        src = os.path.join(csrc, obj[:-2] + ".hip")
        listed = {os.path.normpath(os.path.join(csrc, d)) for d in deps.split()}
        missing = [os.path.relpath(f, csrc) for f in includes(src, set()) if f not in listed]
        assert not missing, (obj, missing)
