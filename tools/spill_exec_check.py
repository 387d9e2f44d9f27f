"""Guard against a register-allocator hazard of this toolchain (ROCm 7.2 hipcc, gfx950): VGPR spill code placed INSIDE a divergent region.

Found in round 3 (kmpc_solve_fast_kernel<double, 28>): the allocator put a group of `scratch_store` spills at the top of a
flow block BEFORE the `s_or_b64 exec, exec, sX` that re-opens the lanes the region had masked off (there: `if (vid < n)`, lanes 56..63 off), and the
reloads after it, under the full mask.  Lanes that were off keep whatever the slot held before: per-lane iterates of those lanes (slacks and
multipliers of forms 56..63) and wave-uniform scalars (the shift `reg`, stored to LDS by all lanes, last lane wins) came back wrong -- the
solver then converged, by its own measure, on a point that is not a KKT point.  Codegen-dependent: the same source passed or failed
depending on unrelated edits.

The check walks the gfx950 assembly of a translation unit block by block and reports every spill store (scratch_store, or v_accvgpr_write
where AGPRs serve as spill space) that sits in a block ahead of that block's `s_or_b64 exec, exec, sX` with no narrowing in between.

usage: python tools/spill_exec_check.py file.hip [more.hip ...] [-D MACRO ...]     exit code 1 if anything is flagged
       python tools/spill_exec_check.py --asm file.s
"""
import os, re, subprocess, sys, tempfile

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on"]


def device_asm(src, defs=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-D" + d for d in defs] + ["--cuda-device-only", "-S", "-o", out, src]
    subprocess.check_call(cmd)
    txt = open(out).read()
    os.unlink(out)
    return txt


def kernels(asm):
    """yield (mangled name, lines) for every function body"""
    name, body = None, []
    for l in asm.splitlines():
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if l.startswith("\t.size\t" + name):
            yield name, body
            name = None
            continue
        body.append(l)


SCR = re.compile(r"\s*scratch_(load|store)_(dword|dwordx2|dwordx3|dwordx4|short|ubyte|byte)\S*\s+(.*)")


def check_kernel(lines):
    """-> (n_spill_ops, [(block label, line no, text)]): spill stores that sit in a block AHEAD of that block's `s_or_b64 exec, exec, sX`
    (the mask restore of a divergent region): they run with the region's lanes only, the value is live beyond it."""
    nops, bad = 0, []
    label, pend = "entry", []   # pend: spill stores seen in the current block with no exec change since the label
    narrowed = False            # an exec-narrowing instruction was seen in this block before the pending stores
    for i, l in enumerate(lines):
        t = l.strip()
        if re.match(r"^\.?\w+:", l):          # a label starts a new block
            label, pend, narrowed = l.split(":")[0], [], False
            continue
        if t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            pend, narrowed = [], False
            continue
        if t.startswith(("s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec", "s_andn2_b64 exec", "s_and_b64 exec", "s_xor_b64 exec", "s_mov_b64 exec")):
            pend, narrowed = [], True          # stores before a narrowing ran under the wider mask: fine
            continue
        if re.match(r"s_or_b64 exec, exec,", t):
            if not narrowed:
                bad += [(label, j, lines[j].strip()) for j in pend]
            pend, narrowed = [], False
            continue
        m = SCR.match(l)
        if m:
            nops += 1
            if m.group(1) == "store":
                pend.append(i)
            continue
        m = re.match(r"v_accvgpr_(write|read)_b32\s", t)
        if m:   # AGPRs are spill space in these kernels (the matrix-core accumulators live in VGPRs)
            nops += 1
            if m.group(1) == "write":
                pend.append(i)
    return nops, bad


def demangle(names):
    r = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(KP.*", "", d).replace("void ", "") for d in r]


def check_asm(asm, label, verbose=True):
    rows = [(n, *check_kernel(b)) for n, b in kernels(asm)]
    rows = [r for r in rows if r[1] > 0 or r[2]]
    names = demangle([r[0] for r in rows]) if rows else []
    nbad = 0
    for (n, nops, bad), d in zip(rows, names):
        if verbose or bad:
            print("%-24s %-66s spill ops %4d  stores ahead of an exec restore %3d %s" % (label, d[:66], nops, len(bad), "<-- HAZARD" if bad else ""))
            for (lab, j, txt) in bad[:4]:
                print("        %s +%d: %s" % (lab, j, txt))
        nbad += bool(bad)
    return nbad


if __name__ == "__main__":
    args = sys.argv[1:]
    defs = []
    while "-D" in args:
        k = args.index("-D"); defs.append(args[k + 1]); del args[k:k + 2]
    nbad = 0
    if args and args[0] == "--asm":
        for f in args[1:]:
            nbad += check_asm(open(f).read(), os.path.basename(f))
    else:
        for f in args:
            nbad += check_asm(device_asm(f, defs), os.path.basename(f))
    print("flagged kernels:", nbad)
    sys.exit(1 if nbad else 0)
