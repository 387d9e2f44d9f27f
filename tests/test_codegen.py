"""Code-generation guard for the compile-time-horizon solver kernels (CPU only: hipcc cross-compiles gfx950 assembly, nothing runs).

Round 3 found spill code of this toolchain's register allocator inside a divergent region -- scratch stores ahead of the `s_or_b64 exec`
that re-opens the masked lanes, reloads after it -- in kmpc_solve_fast_kernel<double, 28>: lanes that were off came back with stale
slot contents and the solver converged, by its own measure, on points that are not KKT points (tools/spill_exec_check.py, DESIGN.md
section 9).  The kernels that were hit now run at an occupancy that needs no scratch; this test keeps every shipped instantiation of
the FOUR translation units that instantiate the solve free of the pattern.  Round 4: the generic kernel (kmpc_kernels.hip) is checked too --
its reductions now hand their results to the compiler as wave-uniform values (uniform_(), kmpc_common.h), so the state machine's branches
are scalar branches there as well and the 16 sites the checker had found in it (phi copies and spills between nested mask restores) are gone;
at run time ipm::solve refuses to report an iterate Optimal whose slacks have parted from b - a_f^T U (tests/test_gpu_parity.py).
The checker cannot tell a spill from a phi copy into an AGPR-resident variable: whatever it flags has to be removed, not argued away."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "csrc")


def test_checker_recognises_the_pattern():
    import spill_exec_check as S
    flow_block = """
.LBB1_3:
\tscratch_store_dwordx2 off, v[2:3], off offset:8 ; 8-byte Folded Spill
\tv_mov_b32_e32 v9, v8
\ts_or_b64 exec, exec, s[18:19]
\tv_add_f64 v[4:5], v[4:5], v[6:7]
""".splitlines()
    n, bad = S.check_kernel(flow_block)
    assert n == 1 and len(bad) == 1 and bad[0][0] == ".LBB1_3"
    after_restore = """
.LBB1_3:
\ts_or_b64 exec, exec, s[18:19]
\tscratch_store_dwordx2 off, v[2:3], off offset:8 ; 8-byte Folded Spill
\ts_and_saveexec_b64 s[2:3], vcc
\tscratch_load_dwordx2 v[2:3], off, off offset:8 ; 8-byte Folded Reload
\ts_or_b64 exec, exec, s[2:3]
""".splitlines()
    n, bad = S.check_kernel(after_restore)
    assert n == 2 and not bad


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_no_spill_stores_ahead_of_an_exec_restore():
    import spill_exec_check as S

    def one(name):
        return name, S.device_asm(os.path.join(CSRC, name))
    with ThreadPoolExecutor(4) as ex:
        asms = list(ex.map(one, ["kmpc_fast.hip", "kmpc_wide.hip", "kmpc_quad.hip", "kmpc_kernels.hip"]))
    flagged = []
    for name, asm in asms:
        kernels = list(S.kernels(asm))
        assert len(kernels) >= 2, name
        for k, body in kernels:
            _, bad = S.check_kernel(body)
            if bad:
                flagged.append((name, subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:80], bad[:3]))
    assert not flagged, flagged
