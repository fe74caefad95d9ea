"""The drop-in boundary end to end: a Fortran driver program shaped like the reference's
testcases/mcmcrun.F90 (`call mcmc_main()`), linked against the ISO_C_BINDING shim
(mcmcf90_amd/fortran/mcmcx_mod.F90) + libmcmcx.so, reads the reference's own shipped namelist
(testcases/mcmcinit.nml values) and .dat inputs and must write the chain the real reference wrote
(fixture c1_shipped_nml): identical run-length column, theta to BLAS/libm rounding."""
import os
import subprocess
import tempfile
import numpy as np
import pytest
from golden_util import load

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "mcmcf90_amd", "fortran")

NML = """!! Run time parameters for the mcmc run (values of the reference's testcases/mcmcinit.nml)
&mcmc
 method = 'dram'
 nsimu       = 1000
 verbosity   = 1
 doadapt     = 1
 adaptint    = 200
 burnintime  = 1000
 doburnin    = 1
 drscale     = 0
 printint    = 100
 updatesigma = 1
 N0          = 1
 S02         = 0
 chainfile   = 'chain.dat'
 ssfile      = 'sschain.dat'
 s2file      = 's2chain.dat'
/
&mcmcx
 devtarget = 'expdata'
 datafile  = 'data.dat'
 lowerfile = 'lower.dat'
 nchains   = 64
/
"""


def test_fortran_driver_reproduces_reference_chain(oracle):
    exe = os.path.join(FDIR, "demo_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("c1_shipped_nml", oracle)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write(NML)
        with open(os.path.join(d, "data.dat"), "w") as f:
            f.write("% example data set\n")
            for x, y in zip(z["prob_xdata"], z["prob_ydata"]):
                f.write("  %g   %.2f\n" % (x, y))
        open(os.path.join(d, "mcmcpar.dat"), "w").write("10 0.1 \n")
        open(os.path.join(d, "mcmccov.dat"), "w").write("0.2 0 \n0 0.001 \n")
        open(os.path.join(d, "mcmcsigma2.dat"), "w").write("0.5\n11\n")
        open(os.path.join(d, "lower.dat"), "w").write("0 0\n")
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        out = p.stdout.decode(errors="replace")
        assert p.returncode == 0, out
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        s2 = np.loadtxt(os.path.join(d, "s2chain.dat"), ndmin=1)
        ss = np.loadtxt(os.path.join(d, "sschain.dat"), ndmin=2)
        last = np.loadtxt(os.path.join(d, "mcmclaststates.dat"), ndmin=2)        # nchains = 64: engine extension files
        pmean = np.loadtxt(os.path.join(d, "mcmcpooledmean.dat"), ndmin=1)
        pcov = np.loadtxt(os.path.join(d, "mcmcpooledcov.dat"), ndmin=2)
    assert last.shape == (64, 2)
    np.testing.assert_array_equal(last[0], chain[-1, :2])
    np.testing.assert_allclose(pmean, last.mean(axis=0), rtol=1e-12)
    np.testing.assert_allclose(pcov, np.cov(last.T), rtol=1e-9)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), z["runlen"])
    k = z["rows_head"].shape[0]
    np.testing.assert_allclose(chain[:k, :-1], z["rows_head"], rtol=1e-9)
    np.testing.assert_allclose(chain[-k:, :-1], z["rows_tail"], rtol=1e-9)
    np.testing.assert_allclose(s2[-k:], z["s2_tail"], rtol=1e-9)
    np.testing.assert_allclose(ss[-k:, 0], z["ss_tail"], rtol=1e-9)
    # and bit for bit against the oracle (ES24.16 round-trips a double)
    o = oracle.run_chain(cfg, prob, chain_id=0)
    np.testing.assert_array_equal(chain, o.chain)
    np.testing.assert_array_equal(s2, o.s2chain)


def test_unmodified_user_program_with_fortran_callbacks(oracle):
    """demo_user.F90 is a complete mcmcf90-style user program (own Fortran ssfunction reading data.dat, own
    checkbounds) with the reference's shipped namelist and NO engine-specific input: the shim routes the
    callbacks through the host-callback path.  The user's ssfunction uses the Fortran runtime's exp, so it is
    compared with the real reference's chain at the libm rounding level; the accept sequence is identical."""
    exe = os.path.join(FDIR, "demo_user")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("c1_shipped_nml", oracle)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write(NML.split("&mcmcx")[0])
        with open(os.path.join(d, "data.dat"), "w") as f:
            f.write("% example data set\n")
            for x, y in zip(z["prob_xdata"], z["prob_ydata"]):
                f.write("  %g   %.2f\n" % (x, y))
        open(os.path.join(d, "mcmcpar.dat"), "w").write("10 0.1 \n")
        open(os.path.join(d, "mcmccov.dat"), "w").write("0.2 0 \n0 0.001 \n")
        open(os.path.join(d, "mcmcsigma2.dat"), "w").write("0.5\n11\n")
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        s2 = np.loadtxt(os.path.join(d, "s2chain.dat"), ndmin=1)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), z["runlen"])
    k = z["rows_head"].shape[0]
    np.testing.assert_allclose(chain[-k:, :-1], z["rows_tail"], rtol=1e-9)
    np.testing.assert_allclose(s2[-k:], z["s2_tail"], rtol=1e-9)


def test_mat4_output_and_restart_namelist(oracle):
    """chainfile='chain.mat' selects the MAT-v4 writer (MCMC_aux.F90:25-29, matfiles.F90:66-126) and nmlffile the restart
    namelist with initcmatn += nsimu, burnintime = 0 (MCMC_aux.F90:78-83)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle.refrun import read_mat4
    exe = os.path.join(FDIR, "demo_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("c1_shipped_nml", oracle)
    nml = NML.replace("chainfile   = 'chain.dat'", "chainfile   = 'chain.mat'").replace("s2file      = 's2chain.dat'", "s2file = 's2chain.mat'\n nmlffile = 'mcmcinitf.nml'\n covnfile = 'mcmccovn.dat'")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write(nml)
        with open(os.path.join(d, "data.dat"), "w") as f:
            for x, y in zip(z["prob_xdata"], z["prob_ydata"]):
                f.write("  %g   %.2f\n" % (x, y))
        open(os.path.join(d, "mcmcpar.dat"), "w").write("10 0.1 \n")
        open(os.path.join(d, "mcmccov.dat"), "w").write("0.2 0 \n0 0.001 \n")
        open(os.path.join(d, "mcmcsigma2.dat"), "w").write("0.5\n11\n")
        open(os.path.join(d, "lower.dat"), "w").write("0 0\n")
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = read_mat4(os.path.join(d, "chain.mat"))["chain"]
        s2 = read_mat4(os.path.join(d, "s2chain.mat"))["s2chain"][:, 0]
        restart = open(os.path.join(d, "mcmcinitf.nml")).read().upper()
    o = oracle.run_chain(cfg, prob, chain_id=0)
    np.testing.assert_array_equal(chain, o.chain)
    np.testing.assert_array_equal(s2, o.s2chain)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), z["runlen"])
    import re
    assert re.search(r"INITCMATN\s*=\s*1000", restart) and re.search(r"BURNINTIME\s*=\s*0\b", restart)


def _write_inputs(d, z, nml):
    open(os.path.join(d, "mcmcinit.nml"), "w").write(nml)
    with open(os.path.join(d, "data.dat"), "w") as f:
        f.write("% example data set\n")
        for x, y in zip(z["prob_xdata"], z["prob_ydata"]):
            f.write("  %g   %.2f\n" % (x, y))
    open(os.path.join(d, "mcmcpar.dat"), "w").write("10 0.1 \n")
    open(os.path.join(d, "mcmccov.dat"), "w").write("0.2 0 \n0 0.001 \n")
    open(os.path.join(d, "mcmcsigma2.dat"), "w").write("0.5\n11\n")


@pytest.mark.parametrize("prog", ["tc_mcmcrun", "tc_mcmcrun2", "tc_mcmcrun3"])
def test_reference_example_programs_run_unmodified(oracle, prog):
    """oracle/_ref/tc_* are the reference's OWN example programs (testcases/mcmcrun.F90: file-based initialize;
    mcmcrun2.F90: user-supplied `initialize` with allocatable dummies; mcmcrun3.F90: MCMC_setpar0 / _setcmat0 /
    _setsigma2nobs before mcmc_main), compiled from /root/reference without a change and linked against the
    engine's shim instead of libmcmcrun.a (oracle/Makefile, target testcases).  With the reference's shipped namelist
    and input files all three must write the chain the real reference wrote (fixture c1_shipped_nml)."""
    exe = os.path.join(ROOT, "oracle", "_ref", prog)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s was not built (needs /root/reference in the dev container)" % prog)
    z, cfg, prob = load("c1_shipped_nml", oracle)
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, NML.split("&mcmcx")[0])
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        s2 = np.loadtxt(os.path.join(d, "s2chain.dat"), ndmin=1)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), z["runlen"])
    k = z["rows_head"].shape[0]
    np.testing.assert_allclose(chain[:k, :-1], z["rows_head"], rtol=1e-9)
    np.testing.assert_allclose(chain[-k:, :-1], z["rows_tail"], rtol=1e-9)
    np.testing.assert_allclose(s2[-k:], z["s2_tail"], rtol=1e-9)


def test_config1_verbatim_10000_iterations_with_delayed_rejection(oracle):
    """BASELINE config 1 as written: the reference's UNMODIFIED testcases/mcmcrun.F90:48-122 (oracle/_ref/tc_mcmcrun: its own Fortran
    ssfunction / modelfunction / checkbounds on the host, file-based initialize, one chain) on the bundled data with DRAM -- nsimu =
    10000, drscale = 2, greedy burn-in scaling up to iteration 1000, sigma2 update -- linked against the shim instead of libmcmcrun.a.
    The files it leaves must be the chain the real reference produced with the same namelist (fixture c1_expdata_dram, made by
    oracle/gen_golden.py): identical run-length column (every accept decision of the 10000 iterations and of the delayed-rejection
    tries in between), rows, sigma2 chain, and the final covariance / mean files (mcmccovf.dat, mcmcmean.dat) within north_star's
    1e-6."""
    exe = os.path.join(ROOT, "oracle", "_ref", "tc_mcmcrun")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tc_mcmcrun was not built (needs /root/reference in the dev container)")
    z, cfg, prob = load("c1_expdata_dram", oracle)
    nml = """&mcmc
 method = 'dram'
 nsimu       = %d
 verbosity   = 0
 doadapt     = 1
 adaptint    = %d
 burnintime  = %d
 doburnin    = 1
 greedy      = %d
 scalelimit  = %g
 scalefactor = %g
 drscale     = %g
 printint    = 1000
 updatesigma = 1
 N0          = %g
 S02         = %g
 chainfile   = 'chain.dat'
 ssfile      = 'sschain.dat'
 s2file      = 's2chain.dat'
/
""" % (cfg.nsimu, cfg.adaptint, cfg.burnintime, cfg.greedy, cfg.scalelimit, cfg.scalefactor, cfg.drscale, cfg.N0, cfg.S02)
    assert cfg.nsimu == 10000 and cfg.drscale == 2.0 and cfg.doburnin == 1 and cfg.greedy == 1
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, nml)
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        s2 = np.loadtxt(os.path.join(d, "s2chain.dat"), ndmin=1)
        covf = np.loadtxt(os.path.join(d, "mcmccovf.dat"), ndmin=2)
        meanf = np.loadtxt(os.path.join(d, "mcmcmean.dat"), ndmin=1)
    assert chain.shape[0] == int(z["chainind"]) and int(chain[:, -1].sum()) == cfg.nsimu
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), z["runlen"])
    k = z["rows_head"].shape[0]
    np.testing.assert_allclose(chain[:k, :-1], z["rows_head"], rtol=1e-9)
    np.testing.assert_allclose(chain[-k:, :-1], z["rows_tail"], rtol=1e-9)
    np.testing.assert_allclose(s2[:k], z["s2_head"], rtol=1e-9)
    np.testing.assert_allclose(s2[-k:], z["s2_tail"], rtol=1e-9)
    assert np.max(np.abs(covf - z["chaincmat"])) / np.max(np.abs(z["chaincmat"])) < 1e-6
    assert np.max(np.abs(meanf.ravel() - z["chainmean"]) / np.abs(z["chainmean"])) < 1e-6


def test_user_initialize_and_dump_hooks(oracle):
    """demo_hooks.F90 overrides the optional link-time hooks: its `initialize` supplies par0 / cmat0 / sigma2 / nobs
    without any input file, and dump_init / dump / dump_end trace the chain every dumpint iterations."""
    exe = os.path.join(FDIR, "demo_hooks")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("c1_shipped_nml", oracle)
    nml = NML.split("&mcmcx")[0].replace("&mcmc", "&mcmc\n dumpint = 250", 1)
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, nml)
        for f in ("mcmcpar.dat", "mcmccov.dat", "mcmcsigma2.dat"):
            os.remove(os.path.join(d, f))                         # the user's initialize needs none of them
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        trace = open(os.path.join(d, "dump_trace.dat")).read().splitlines()
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), z["runlen"])
    # the progress line of MCMC_adapt.F90:22-37 every printint = 100 iterations, with chain 1's stayed / bounds shares
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith(" simu i =")]
    assert len(lines) == 10
    stayed_at_1000 = 1000 - len(chain)
    assert "%5.1f" % (stayed_at_1000 / 1000.0 * 100.0) in lines[-1] and lines[-1].split()[3] == "1000,"
    assert trace[0] == "# dump_init" and trace[-1] == "# dump_end"
    rows = np.array([[float(v) for v in ln.split()] for ln in trace[1:-1]])
    assert rows.shape == (4, 2)                                   # iterations 250, 500, 750, 1000
    np.testing.assert_array_equal(rows[-1], chain[-1, :2])        # the last dump is the last row of the chain
    starts = np.cumsum(chain[:, -1]) - chain[:, -1] + 1           # iteration at which each row was accepted
    for it, r in zip((250, 500, 750), rows):
        np.testing.assert_array_equal(r, chain[np.searchsorted(starts, it, side="right") - 1, :2])


def test_seed_file_and_signal(oracle):
    """gfortran_seed.dat holds the stream key (mcmcrand.F90:214-238) and is rewritten at the end of the job (:317-342);
    SIGUSR1 during the run saves the chain "upto simuind" and stops (MCMC_signal_handler.F90:95-107)."""
    import signal, time
    exe = os.path.join(FDIR, "demo_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("c1_shipped_nml", oracle)
    chains = []
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, NML)
        open(os.path.join(d, "lower.dat"), "w").write("0 0\n")
        for rep in range(3):
            if rep < 2:
                open(os.path.join(d, "gfortran_seed.dat"), "w").write(" 424242\n")
            p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            assert p.returncode == 0, p.stdout.decode(errors="replace")
            chains.append(np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2))
            nxt = int(open(os.path.join(d, "gfortran_seed.dat")).read().split()[0])
            assert nxt != 424242
    assert np.array_equal(chains[0], chains[1])                   # same key, same chain
    assert not np.array_equal(chains[0][:50], chains[2][:50])     # the rewritten key starts a new stream
    assert not np.array_equal(chains[0][:, -1].astype(np.int32), z["runlen"])   # and 424242 is not the default key
    # signal: a long run, interrupted
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, NML.replace("nsimu       = 1000", "nsimu       = 2000000").replace("adaptint    = 200", "adaptint    = 1000"))
        open(os.path.join(d, "lower.dat"), "w").write("0 0\n")
        p = subprocess.Popen([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        time.sleep(4.0)
        p.send_signal(signal.SIGUSR1)
        out, _ = p.communicate(timeout=600)
        text = out.decode(errors="replace")
        assert "Saving chain upto" in text, text[-2000:]
        upto = int(text.split("Saving chain upto")[1].split()[0])
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        assert 1 <= upto < 2000000 and int(chain[:, -1].sum()) == upto


def test_reference_gaussian_example_mcmcrun4():
    """testcases/mcmcrun4.F90 unmodified: MCMC_setpar0(5, 0.0), MCMC_setcmat0(0.1), the user's dense Gaussian
    ssfunction reading mcmctest_mu.dat / mcmctest_lam.dat (files the reference does not ship).  The likelihood runs in
    the user's Fortran, so the check is statistical: the chain must sample N(mu, inv(lam))."""
    exe = os.path.join(ROOT, "oracle", "_ref", "tc_mcmcrun4")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/tc_mcmcrun4 was not built (needs /root/reference in the dev container)")
    d5 = 5
    S = 0.5 ** np.abs(np.subtract.outer(np.arange(d5), np.arange(d5))) * 0.25
    mu = np.linspace(-1.0, 1.0, d5)
    nml = NML.split("&mcmcx")[0].replace("nsimu       = 1000", "nsimu       = 40000").replace("burnintime  = 1000", "burnintime  = 0") \
        .replace("doburnin    = 1", "doburnin    = 0").replace("updatesigma = 1", "updatesigma = 0").replace("adaptint    = 200", "adaptint    = 100")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write(nml)
        np.savetxt(os.path.join(d, "mcmctest_mu.dat"), mu[None, :])
        np.savetxt(os.path.join(d, "mcmctest_lam.dat"), np.linalg.inv(S))
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
    w = chain[:, -1]
    assert int(w.sum()) == 40000
    th = np.repeat(chain[:, :-1], w.astype(int), axis=0)[5000:]
    assert np.max(np.abs(th.mean(axis=0) - mu)) < 0.08
    assert np.max(np.abs(np.cov(th.T) - S)) < 0.08
    assert 0.1 < len(chain) / 40000.0 < 0.6


def test_user_program_with_method_er(oracle):
    """The unmodified user program (own Fortran ssfunction / checkbounds, the library's default ssfunction_er and
    priorfun with a priorsfile) with method = 'er' in the namelist: fixture e6_expdata_er_priors' accept sequence."""
    exe = os.path.join(FDIR, "demo_user")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("e6_expdata_er_priors", oracle)
    nml = """&mcmc
 method = 'er'
 nsimu = %d
 adaptint = %d
 updatesigma = 1
 priorsfile = 'priors.dat'
/
""" % (cfg.nsimu, cfg.adaptint)
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, nml)
        np.savetxt(os.path.join(d, "priors.dat"), np.vstack([z["prob_pri_mu"], z["prob_pri_sig"]]))
        open(os.path.join(d, "gfortran_seed.dat"), "w").write(" 1835232611\n")
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
    # the fixture's stream is chain id 41 of the default key; the shim's chain 0 has another stream: compare with the oracle
    o = oracle.run_chain(cfg, prob, chain_id=0)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), o.chain[:, -1].astype(np.int32))
    np.testing.assert_allclose(chain[:, :-1], o.chain[:, :-1], rtol=1e-9)


@pytest.mark.parametrize("seed", range(12))
def test_random_namelists_through_the_shim(oracle, seed):
    """Random values for the namelist's numeric control variables, written as mcmcinit.nml, through the Fortran shim
    (demo_main, device-resident expdata target): chain.dat must be the oracle's chain for the same settings -- checks
    the namelist -> mcmcx_config mapping variable by variable."""
    exe = os.path.join(FDIR, "demo_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, _, _ = load("c1_shipped_nml", oracle)
    r = np.random.default_rng(21000 + seed)
    method = str(r.choice(["dram", "dram", "ram", "er", "scam"]))
    ckw = dict(nsimu=int(r.integers(300, 900)), method=method, adaptint=int(r.choice([50, 100])), updatesigma=int(r.integers(0, 2)),
               N0=float(r.choice([1.0, 3.0])), S02=float(r.choice([0.0, 0.6])), initcmatn=int(r.choice([0, 25])))
    if method == "dram":
        ckw.update(drscale=float(r.choice([0.0, 2.0])), doburnin=int(r.integers(0, 2)), burnintime=int(r.choice([0, 150])),
                   scalelimit=float(r.choice([0.05, 0.3])), scalefactor=float(r.choice([2.5, 1.5])), greedy=int(r.integers(0, 2)),
                   adapthist=int(r.choice([0, 120])), adaptend=int(r.choice([0, 250])), badaptint=int(r.choice([-1, 30])))
    if method == "ram":
        ckw.update(alphatarget=float(r.choice([0.234, 0.35])), nuparam=float(r.choice([0.7, 0.55])))
    if method == "scam":
        ckw.update(condmax=float(r.choice([0.0, 1e12])))
    pkw = dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5, nobs=11,
               xdata=z["prob_xdata"], ydata=z["prob_ydata"], lo=[0, 0])
    nml = "&mcmc\n" + "".join(" %s = %s\n" % (k, ("'%s'" % v) if isinstance(v, str) else repr(v)) for k, v in ckw.items()) + \
          " printint = 0\n/\n&mcmcx\n devtarget = 'expdata'\n datafile = 'data.dat'\n lowerfile = 'lower.dat'\n nchains = 3\n/\n"
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, nml)
        open(os.path.join(d, "lower.dat"), "w").write("0 0\n")
        o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=0, continue_on_downdate_fail=True)
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        if o.ram_downdate_fail:
            # a failed choldowndate stops the reference (matutils.F90:719-722): so does the drop-in, with its message
            # and without chain files (the engine itself only flags the chain)
            assert p.returncode != 0 and b"error in coldowndate" in p.stdout and not os.path.exists(os.path.join(d, "chain.dat")), nml
            return
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        cov = np.loadtxt(os.path.join(d, "mcmccovf.dat"), ndmin=2)
    np.testing.assert_array_equal(chain, o.chain, err_msg=nml)
    if method != "ram":
        np.testing.assert_array_equal(np.triu(cov), np.triu(o.chaincmat), err_msg=nml)


@pytest.mark.parametrize("method,extra", [("scam", ""), ("dram", " condmax = 1e10\n drscale = 2.0\n")])
def test_user_program_with_the_svd_paths(oracle, method, extra):
    """The unmodified user program (its own Fortran ssfunction / checkbounds on the host) with method = 'scam', and with
    the SVD proposal factor (condmax > 0) plus delayed rejection: every componentwise proposal / every DR stage makes its
    round trip to the user's functions, and the chain is the oracle's."""
    exe = os.path.join(FDIR, "demo_user")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, _, _ = load("s4_expdata_scam_s2", oracle)
    ckw = dict(nsimu=400, method=method, adaptint=100, updatesigma=1)
    if method == "dram":
        ckw.update(condmax=1e10, drscale=2.0)
    pkw = dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5, nobs=11,
               xdata=z["prob_xdata"], ydata=z["prob_ydata"], lo=[0, 0])
    nml = "&mcmc\n method = '%s'\n nsimu = 400\n adaptint = 100\n updatesigma = 1\n%s/\n" % (method, extra)
    with tempfile.TemporaryDirectory() as d:
        _write_inputs(d, z, nml)
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
    o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=0)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), o.chain[:, -1].astype(np.int32))
    np.testing.assert_allclose(chain[:, :-1], o.chain[:, :-1], rtol=1e-8)


def test_user_program_with_two_response_columns(oracle):
    """demo_cols.F90: nycol = 2 from MCMC_setsigma2nobs((/s1, s2/), (/n1, n2/)), the user's ssfunction returns ss(1:2);
    chain / sschain (three columns) / s2chain (two columns) / mcmcsigma2f.dat (2 x 2) against the oracle on fixture m1's problem."""
    exe = os.path.join(FDIR, "demo_cols")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("m1_expdata2_dram_dr_s2", oracle)
    nml = """&mcmc
 nsimu = %d
 adaptint = %d
 updatesigma = 1
 drscale = 2.0
 N0 = 1
 S02 = 0
 printint = 0
/
""" % (cfg.nsimu, cfg.adaptint)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write(nml)
        x, Y = z["prob_xdata"], z["prob_ydata"]
        with open(os.path.join(d, "data2.dat"), "w") as f:
            f.write("% x y1 y2\n")
            for i in range(len(x)):
                f.write("  %r   %r   %r\n" % (float(x[i]), float(Y[0, i]), float(Y[1, i])))
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        ss = np.loadtxt(os.path.join(d, "sschain.dat"), ndmin=2)
        s2 = np.loadtxt(os.path.join(d, "s2chain.dat"), ndmin=2)
        sf = np.loadtxt(os.path.join(d, "mcmcsigma2f.dat"), ndmin=2)
    o = oracle.run_chain(cfg, prob, chain_id=0)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), o.chain[:, -1].astype(np.int32))
    np.testing.assert_allclose(chain[:, :-1], o.chain[:, :-1], rtol=1e-8)
    assert ss.shape == o.sschain.shape == (len(chain), 3) and s2.shape == o.s2chain.shape == (cfg.nsimu, 2)
    np.testing.assert_allclose(ss, o.sschain, rtol=1e-8)
    np.testing.assert_allclose(s2, o.s2chain, rtol=1e-8)
    np.testing.assert_allclose(sf, np.vstack([o.s2chain[-1], [11.0, 13.0]]), rtol=1e-8)


def test_two_response_columns_with_the_device_resident_target(oracle):
    """The same two-column problem with `&mcmcx devtarget = 'expcols'`: the response-column model evaluated on the
    device (dev_eval_kernel between the phase kernels), 96 chains; chain 1's files against the oracle."""
    exe = os.path.join(FDIR, "demo_cols")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("m1_expdata2_dram_dr_s2", oracle)
    nml = """&mcmc
 nsimu = %d
 adaptint = %d
 updatesigma = 1
 drscale = 2.0
 N0 = 1
 S02 = 0
 printint = 0
/
&mcmcx
 devtarget = 'expcols'
 datafile = 'data2.dat'
 lowerfile = 'lower.dat'
 nchains = 96
/
""" % (cfg.nsimu, cfg.adaptint)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "mcmcinit.nml"), "w").write(nml)
        open(os.path.join(d, "lower.dat"), "w").write("0.0 0.0 0.0\n")
        x, Y = z["prob_xdata"], z["prob_ydata"]
        with open(os.path.join(d, "data2.dat"), "w") as f:
            for i in range(len(x)):
                f.write("  %r   %r   %r\n" % (float(x[i]), float(Y[0, i]), float(Y[1, i])))
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        chain = np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2)
        ss = np.loadtxt(os.path.join(d, "sschain.dat"), ndmin=2)
        s2 = np.loadtxt(os.path.join(d, "s2chain.dat"), ndmin=2)
        last = np.loadtxt(os.path.join(d, "mcmclaststates.dat"), ndmin=2)
    o = oracle.run_chain(cfg, prob, chain_id=0)
    np.testing.assert_array_equal(chain[:, -1].astype(np.int32), o.chain[:, -1].astype(np.int32))
    np.testing.assert_allclose(chain[:, :-1], o.chain[:, :-1], rtol=1e-13)
    np.testing.assert_allclose(ss, o.sschain, rtol=1e-13)
    np.testing.assert_allclose(s2, o.s2chain, rtol=1e-13)
    assert last.shape == (96, 3)
    o95 = oracle.run_chain(cfg, prob, chain_id=95)
    np.testing.assert_allclose(last[95], o95.theta, rtol=1e-13)


def test_ngpus_namelist_variable():
    """&mcmcx ngpus: one engine per GPU under one RCCL communicator (mcmcx_comm_create_all / mcmcx_run_all).  The test
    box has one GPU: ngpus = 1 must equal the default run, ngpus = 2 must stop with the engine's message instead of
    quietly running on one device."""
    exe = os.path.join(FDIR, "demo_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    import ctypes
    n = ctypes.c_int(0)
    ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n))
    nml = "&mcmc\n method = 'dram'\n nsimu = 300\n updatesigma = 0\n verbosity = 0\n/\n&mcmcx\n devtarget = 'banana'\n nchains = 128\n ngpus = %d\n pooled = %d\n usecomm = %d\n/\n"
    res = {}
    for ng, pooled, usecomm in ((1, 0, 0), (2, 0, 0), (2, 1, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)):
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "mcmcinit.nml"), "w").write(nml % (ng, pooled, usecomm))
            open(os.path.join(d, "mcmcpar.dat"), "w").write("0 0 0 0\n")
            open(os.path.join(d, "mcmccov.dat"), "w").write("1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n")
            p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            out = p.stdout.decode(errors="replace")
            if ng > n.value:
                assert p.returncode != 0 and "HIP device(s) are visible" in out, out
                assert not os.path.exists(os.path.join(d, "chain.dat"))
                continue
            assert p.returncode == 0, out
            res[(ng, pooled, usecomm)] = (np.loadtxt(os.path.join(d, "chain.dat"), ndmin=2), np.loadtxt(os.path.join(d, "mcmclaststates.dat"), ndmin=2),
                                          np.loadtxt(os.path.join(d, "mcmcpooledcov.dat"), ndmin=2))
    assert res[(1, 0, 0)][1].shape == (128, 4)
    if (2, 0, 0) in res:                                # a multi-GPU box: the chains do not depend on the GPU count
        for a, b in zip(res[(1, 0, 0)], res[(2, 0, 0)]):
            np.testing.assert_array_equal(a, b)
    # usecomm = 1: the N-GPU code path end to end on the one GPU -- ncclCommInitAll(1), mcmcx_set_comm, mcmcx_run_all with the
    # pooled ticks' all-gathers on the communicator, mcmcx_allreduce_moments_all -- must reproduce the run without it
    for pooled in (0, 1):
        for a, b in zip(res[(1, pooled, 0)], res[(1, pooled, 1)]):
            np.testing.assert_array_equal(a, b)
    assert not np.array_equal(res[(1, 0, 0)][1], res[(1, 1, 0)][1])          # (the pooled run is a different run)


def test_user_program_batched_and_module_targets(oracle, tmp_path):
    """The two GPU-speed forms of the user's functions from a Fortran program's namelist.
    (1) &mcmcx hostbatch = 1: the unmodified demo_user program (own Fortran ssfunction) evaluated through
        ssfunction_batch (default member: a loop over ssfunction), several chains, one and three threads -- the chain
        files equal those of the one-call-per-chain path byte for byte.
    (2) &mcmcx devtarget = 'module': device code built with include/mcmcx_target.h, loaded by the shim, its data read
        from moduledatafile -- chain 1 equals the same module driven through the Python mirror."""
    exe = os.path.join(FDIR, "demo_user")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    z, cfg, prob = load("c1_shipped_nml", oracle)
    outs = []
    for extra in ("", "&mcmcx\n nchains = 70\n hostbatch = 1\n hostthreads = 1\n/\n", "&mcmcx\n nchains = 70\n hostbatch = 1\n hostthreads = 3\n/\n",
                  "&mcmcx\n nchains = 70\n/\n"):
        d = tmp_path / ("b%d" % len(outs)); d.mkdir()
        (d / "mcmcinit.nml").write_text(NML.split("&mcmcx")[0] + extra)
        with open(d / "data.dat", "w") as f:
            for x, y in zip(z["prob_xdata"], z["prob_ydata"]):
                f.write("  %g   %.2f\n" % (x, y))
        (d / "mcmcpar.dat").write_text("10 0.1 \n"); (d / "mcmccov.dat").write_text("0.2 0 \n0 0.001 \n"); (d / "mcmcsigma2.dat").write_text("0.5\n11\n")
        p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode(errors="replace")
        outs.append(((d / "chain.dat").read_bytes(), (d / "s2chain.dat").read_bytes(),
                     (d / "mcmclaststates.dat").read_bytes() if extra else None))
    assert outs[0][0] == outs[1][0] == outs[2][0] == outs[3][0] and outs[0][1] == outs[1][1] == outs[2][1]
    assert outs[1][2] == outs[2][2] == outs[3][2]                        # all 70 chains' last states
    # ---- (2) module target through the namelist
    src = tmp_path / "poly.hip"
    src.write_text('''#include "mcmcx_target.h"
__device__ void p_ss(const double *th, int npar, int ny, const void *data, double *ss)
{ const double *w = (const double *)data; double s = 0.0; for (int k = 0; k < npar; ++k) { double q = th[k] - w[k]; s = s + w[npar + k] * (q * q); } ss[0] = s; }
__device__ double p_prior(const double *th, int npar, const void *data) { return 0.0; }
__device__ int p_bounds(const double *th, int npar, const void *data) { return th[0] > -4.0 ? 1 : 0; }
MCMCX_DEFINE_TARGET(poly_target, p_ss, p_prior, p_bounds)
''')
    hsaco = tmp_path / "poly.hsaco"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O2", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(hsaco)])
    data = np.array([0.5, -0.25, 1.0, 2.0, 0.7, 1.3])                    # centres (3), weights (3)
    d = tmp_path / "mod"; d.mkdir()
    (d / "mcmcinit.nml").write_text("&mcmc\n method = 'dram'\n nsimu = 400\n adaptint = 50\n drscale = 2\n updatesigma = 0\n verbosity = 0\n/\n"
                                    "&mcmcx\n devtarget = 'module'\n modulefile = '%s'\n modulekernel = 'poly_target'\n moduledatafile = 'w.dat'\n nchains = 64\n/\n" % hsaco)
    (d / "w.dat").write_text(" ".join(repr(float(v)) for v in data) + "\n")
    (d / "mcmcpar.dat").write_text("0 0 0\n"); (d / "mcmccov.dat").write_text("0.1 0 0\n0 0.1 0\n0 0 0.1\n")
    p = subprocess.run([os.path.join(FDIR, "demo_main")], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode(errors="replace")
    chain = np.loadtxt(d / "chain.dat", ndmin=2)
    from mcmcf90_amd import Engine, make_config
    e = Engine(make_config(3, 64, nsimu=400, adaptint=50, drscale=2.0, updatesigma=0, record_chain=1))
    e.setpar0(np.zeros(3)); e.setcmat0(0.1 * np.eye(3)); e.setsigma2nobs(1.0, 1)
    e.set_target_module(str(hsaco), "poly_target", data)
    e.init(); e.run()
    ch, _, _ = e.chain(0)
    e.close()
    np.testing.assert_array_equal(chain, ch)
