"""tools/hostcb_breakeven.py -- what the single-chain user of the drop-in sees, and where it turns around.  GPU box.

The reference's own unmodified testcases/mcmcrun.F90 (BASELINE configuration 0: the user's Fortran ssfunction on the HOST) linked against
libmcmcxf.a + libmcmcx.so, with `&mcmcx nchains = N` (and `hostbatch = 1`), and the same model device-resident (`devtarget = 'expdata'`), against the
reference program itself on one host core: microseconds per iteration and per chain-iteration."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_fortran_shim as T
from golden_util import load
from oracle import pyoracle as oracle, refrun as rr

z, cfg, prob = load("c1_shipped_nml", oracle)
tc = os.path.join(ROOT, "oracle", "_ref", "tc_mcmcrun")
dm = os.path.join(ROOT, "mcmcf90_amd", "fortran", "demo_main")
base = T.NML.split("&mcmcx")[0].replace("verbosity   = 1", "verbosity   = 0").replace("printint    = 100", "printint    = 100000000")


def wall(exe, nml, nsimu):
    with tempfile.TemporaryDirectory() as d:
        T._write_inputs(d, z, nml.replace("nsimu       = 1000", "nsimu       = %d" % nsimu))
        open(os.path.join(d, "lower.dat"), "w").write("0 0\n")
        t0 = time.perf_counter()
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            print(p.stdout.decode(errors="replace")[-800:])
        return dt


def per_it(exe, nml, guess_us):
    """us per iteration from two run lengths (the difference removes process start, device initialisation and MCMC_init); the longer run is
    sized to ~2 s of sampling by `guess_us`, each length is run twice and the quicker run counts"""
    n2 = int(max(2000, min(400000, 2.0e6 / guess_us)))
    n1 = max(400, n2 // 5)
    wall(exe, nml, n1)                                       # warm: binary, libraries, device
    t1 = min(wall(exe, nml, n1), wall(exe, nml, n1))
    t2 = min(wall(exe, nml, n2), wall(exe, nml, n2))
    return (t2 - t1) / (n2 - n1) * 1e6, n2


# the reference itself, one core: two run lengths, the difference is the sampling loop
def ref_us():
    out = []
    for n in (200000, 2000000):
        cfgn = oracle.make_cfg(nsimu=n, adaptint=200, burnintime=1000, doburnin=1, drscale=0.0, updatesigma=1, N0=1.0, S02=0.0)
        rr.run_reference(cfgn, prob, timing_only=True)
        out.append(rr.run_reference(cfgn, prob, timing_only=True).seconds)
    return (out[1] - out[0]) / 1800000 * 1e6
r = ref_us() if rr.available() else float("nan")
print("reference (flang -O2 + MKL), one host core, one chain: %.2f us per iteration" % r, flush=True)
print("%-58s %10s %14s %16s" % ("engine form", "nchains", "us/iteration", "us/chain-iter"), flush=True)
for label, exe, extra, counts, guess in (
        ("host ssfunction, one call per chain (the drop-in default)", tc, "", (1, 16, 64, 256, 1024, 4096), lambda n: 40.0 + 0.15 * n),
        ("host ssfunction_batch (&mcmcx hostbatch = 1)", tc, " hostbatch = 1\n", (64, 1024, 4096), lambda n: 40.0 + 0.12 * n),
        ("device-resident target (&mcmcx devtarget = 'expdata')", dm, " devtarget = 'expdata'\n datafile  = 'data.dat'\n lowerfile = 'lower.dat'\n", (1, 64, 4096, 65536, 262144), lambda n: 12.0 + 0.0004 * n)):
    for n in counts:
        nml = base + "&mcmcx\n%s nchains   = %d\n/\n" % (extra, n)
        us, n2 = per_it(exe, nml, guess(n))
        print("%-58s %10d %14.2f %16.4f   (x the reference core's chain-iterations/s: %.2f; nsimu %d)" % (label, n, us, us / n, (r * n) / us, n2), flush=True)
