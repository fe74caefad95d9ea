"""oracle/gen_golden_io.py -- TEST INFRASTRUCTURE. Not part of the product path.

Byte fixture of the reference's ASCII writers: feeds a table of doubles through the REAL reference's
writedata_mat / writedata_vec / writedata_scal (matutils.F90:841-967; oracle/_ref/wd_ref = ref/writedata_probe.F90
linked with the reference library, dev container only) and stores the bytes it wrote.  The committed fixture is data:
the doubles and the files' bytes.  tests/test_fortran_shim_cpu.py runs the same program linked against the engine's shim
(mcmcf90_amd/fortran/demo_writedata) and compares byte for byte.

    python oracle/gen_golden_io.py
"""
import os
import struct
import subprocess
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "io", "writedata.npz")


def table():
    """7 x 6 doubles: what MCMC_writechains meets (chain values, repeat counts, covariances, sigma2) and the corners of G0."""
    rng = np.random.default_rng(20161020)
    special = [0.0, -0.0, 1.0, -1.0, 0.1, 1.0 / 3.0, 9.33, 2.5e10, -2.5e10, 1e15, 1e16, 123456789012345678.0,
               1e-300, 1.7976931348623157e308, 5e-324, 2.2250738585072014e-308, np.pi, -np.e * 1e-7, 1e-5, 0.5e-4, 100.0, 17.0,
               1e22, 1e23, 0.001, 12345.678, 6.02214076e23, 1.0 + 2.0 ** -52]
    vals = special + list(rng.standard_normal(8) * 10.0 ** rng.integers(-12, 12, 8)) + list(rng.integers(1, 500, 6).astype(float))
    return np.array(vals, dtype=np.float64).reshape(7, 6)


def run_probe(exe, a):
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "wd_in.bin"), "wb") as f:
            f.write(struct.pack("<2i", *a.shape))
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        if p.returncode != 0:
            raise RuntimeError(p.stdout.decode(errors="replace"))
        out = {k: open(os.path.join(d, "wd_%s.dat" % k), "rb").read() for k in ("mat", "lock", "vec", "scal")}
        raw = open(os.path.join(d, "wd_back.bin"), "rb").read()
        m, n = struct.unpack_from("<2i", raw, 0)
        out["back"] = np.frombuffer(raw, dtype="<f8", offset=8).reshape(m, n).copy()
        out["leftover"] = sorted(f for f in os.listdir(d) if f.endswith(".lock"))
        return out


if __name__ == "__main__":
    exe = os.path.join(HERE, "_ref", "wd_ref")
    if not os.path.exists(exe):
        sys.exit("oracle/_ref/wd_ref not built (make -C oracle ref; needs /root/reference)")
    a = table()
    r = run_probe(exe, a)
    assert not r["leftover"] and r["mat"] == r["lock"]
    assert np.array_equal(r["back"].view(np.uint64), a.view(np.uint64)), "the reference's own G0 output does not round-trip"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, values=a, mat=np.frombuffer(r["mat"], dtype=np.uint8), vec=np.frombuffer(r["vec"], dtype=np.uint8),
                        scal=np.frombuffer(r["scal"], dtype=np.uint8))
    print(r["mat"].decode()[:400])
    print("wrote", OUT, len(r["mat"]), len(r["vec"]), len(r["scal"]))
