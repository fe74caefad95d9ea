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


def test_group_kernels_hold_no_v_cmpx(tmp_path):
    """mcx_group.hpp's DPP sequences are inline asm that opens with `s_nop 1`: enough for a DPP operand written by the preceding VALU
    instruction, not for a VALU write of EXEC (five wait states), which the hazard recogniser cannot see across an asm statement.  On
    gfx9 the compiler forms exec masks with v_cmp + s_and_saveexec (SALU), never v_cmpx: the shipped code object is checked for it."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "mcmcf90_amd", "libmcmcx.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(lib) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    co = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert co, os.listdir(tmp_path)
    dis = subprocess.run([objdump, "-d", co[0]], cwd=tmp_path, stdout=subprocess.PIPE, check=True).stdout.decode(errors="replace")
    ingroup, seen, bad = False, 0, []
    for line in dis.splitlines():
        if line.endswith(">:"):
            ingroup = "group_step_kernel" in line
            seen += ingroup
        elif ingroup and "v_cmpx" in line:
            bad.append(line.strip())
    assert seen >= 10 and not bad, (seen, bad[:3])
