"""The C-ABI library builds, loads, and exports every symbol include/mcmcx.h declares (CPU-only check)."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mcmcf90_amd import build, _lib
    build.build()
    hdr = open(os.path.join(ROOT, "include", "mcmcx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mcmcx_[A-Za-z0-9_]+)\s*\(", hdr))
    declared -= {"mcmcx_config", "mcmcx_handle", "mcmcx_engine"}
    assert len(declared) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIBPATH]).decode()
    exported = set(re.findall(r"\bT (mcmcx_[A-Za-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.load()
    assert L.mcmcx_version().startswith(b"mcmcx")


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device mcmcx_create must fail (no CPU path).  On a GPU box this is skipped."""
    import ctypes as C
    import pytest
    from mcmcf90_amd import _lib, make_config
    L = _lib.load()
    cfg = make_config(2, 1, nsimu=10)
    h = C.c_void_p()
    rc = L.mcmcx_create(C.byref(cfg), C.byref(h))
    if rc == 0:
        L.mcmcx_destroy(h)
        pytest.skip("a GPU is present")
    assert rc < 0 and b"no HIP device" in L.mcmcx_last_error()


def test_npar_cap_is_checked_before_anything_is_sized():
    """npar <= 4096 (at the old cap of 8192, 64 P = 2.147e9 exceeded INT_MAX and only the size_t casts of the use sites kept the
    packed indices correct).  One past the cap is refused with the cap in the message -- before any device is looked for, so the check runs
    here without a GPU; AT the cap the configuration passes that check (and then fails for want of a device, or is created on a GPU box:
    the P-sized arrays are allocated at mcmcx_init, not here).  64 P and (2 npar + P) 64 at the cap stay below 2**31."""
    import ctypes as C
    from mcmcf90_amd import _lib, make_config
    L = _lib.load()
    h = C.c_void_p()
    rc = L.mcmcx_create(C.byref(make_config(4097, 1, nsimu=10)), C.byref(h))
    assert rc == -5 and b"1..4096" in L.mcmcx_last_error(), (rc, L.mcmcx_last_error())
    rc = L.mcmcx_create(C.byref(make_config(4096, 1, nsimu=10)), C.byref(h))
    if rc == 0:
        L.mcmcx_destroy(h)
    else:
        assert rc == -10 and b"no HIP device" in L.mcmcx_last_error(), (rc, L.mcmcx_last_error())
    P = 4096 * 4097 // 2
    assert 64 * P < 2 ** 31 and (2 * 4096 + P) * 64 < 2 ** 31


def test_header_is_plain_c_and_the_c_example_links():
    """include/mcmcx.h must be usable from C (the reference's host language side binds a C ABI): compile the example
    driver as pedantic C99 against it and link it with libmcmcx.so."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_driver")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", "c_driver.c"), "-L" + os.path.join(root, "mcmcf90_amd"), "-lmcmcx",
           "-Wl,-rpath," + os.path.join(root, "mcmcf90_amd"), "-o", exe]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()


@pytest.mark.gpu
def test_c_example_runs():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_driver")
    if not os.path.exists(exe):
        test_header_is_plain_c_and_the_c_example_links()
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and "simuind 2000" in out and "pooled mean over 256 chains" in out, out


@pytest.fixture(scope="module")
def device_disassembly(tmp_path_factory):
    """llvm-objdump -d of the shipped gfx950 code object (one run for the tests below)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "mcmcf90_amd", "libmcmcx.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(lib) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    tmp = tmp_path_factory.mktemp("dis")
    shutil.copy(lib, tmp / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    co = [f for f in os.listdir(tmp) if "amdgcn" in f]
    assert co, os.listdir(tmp)
    return subprocess.run([objdump, "-d", co[0]], cwd=tmp, stdout=subprocess.PIPE, check=True).stdout.decode(errors="replace").splitlines()


def test_group_kernels_hold_no_v_cmpx(device_disassembly):
    """mcx_group.hpp's DPP sequences are inline asm that opens with `s_nop 1`: enough for a DPP operand written by the preceding VALU
    instruction, not for a VALU write of EXEC (five wait states), which the hazard recogniser cannot see across an asm statement.  On
    gfx9 the compiler forms exec masks with v_cmp + s_and_saveexec (SALU), never v_cmpx: the shipped code object is checked for it."""
    ingroup, seen, bad = False, 0, []
    for line in device_disassembly:
        if line.endswith(">:"):
            ingroup = "group_step_kernel" in line or "group_ram_kernel" in line
            seen += ingroup
        elif ingroup and "v_cmpx" in line:
            bad.append(line.strip())
    assert seen >= 10 and not bad, (seen, bad[:3])


def _vregs(op):
    op = op.strip().rstrip(",")
    m = re.match(r"^v\[(\d+):(\d+)\]$", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", op)
    return {int(m.group(1))} if m else set()


def _dpp_hazards(lines):
    """Every *_dpp instruction of a disassembly: its DPP operand (src0) must not have been written by a VALU instruction within the two
    wait states before it (an instruction = one wait state, `s_nop N` = N + 1; a label forgets the history: another path may arrive there)."""
    novdst = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")
    bad, ndpp, hist = [], 0, []
    for ln in lines:
        t = ln.split("//")[0].strip()
        if not t or t.endswith(":"):
            if t.endswith(":"):
                hist = []
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if mn.endswith("_dpp"):
            ndpp += 1
            src0 = _vregs(ops[1].split()[0]) if len(ops) > 1 else set()
            ws = 0
            for m2, d2, w2 in reversed(hist):
                if ws >= 2:
                    break
                if m2.startswith("v_") and d2 & src0:
                    bad.append((t, m2))
                    break
                ws += w2
        w = int(ops[0], 0) + 1 if (mn == "s_nop" and ops) else 1
        dst = _vregs(ops[0].split()[0]) if (mn.startswith("v_") and not mn.startswith(novdst) and ops) else set()
        hist.append((mn, dst, w))
        if len(hist) > 8:
            hist.pop(0)
    return ndpp, bad


def test_dpp_operands_are_two_wait_states_old(device_disassembly):
    """the inline-asm DPP chains of the group kernels rely on `s_nop 1` at the head of every statement and on no statement
    writing a register one of its own DPP operands reads; the compiler's hazard recogniser does not look into them.  Checked on the
    shipped code object, instruction by instruction, for EVERY DPP instruction (the compiler's own included): the DPP-read operand was
    not written by a vector instruction within the two wait states before it.  (And the checker sees a hazard when there is one.)"""
    n, bad = _dpp_hazards(device_disassembly)
    assert n > 5000 and not bad, (n, bad[:5])
    fake = ["\tv_add_f64 v[4:5], v[0:1], v[2:3]", "\tv_fmac_f64_dpp v[8:9], v[4:5], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf"]
    assert _dpp_hazards(fake)[1] and _dpp_hazards([fake[0], "\ts_nop 0", fake[1]])[1] and not _dpp_hazards([fake[0], "\ts_nop 1", fake[1]])[1]
