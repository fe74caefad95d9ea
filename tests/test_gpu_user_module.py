"""User likelihoods at GPU speed (external_inc.h:12-33): the user's ssfunction / priorfun /
checkbounds compiled as a device code object in the test (hipcc --genco, include/mcmcx_target.h) and loaded with
mcmcx_set_target_module must give, bit for bit, what the SAME C functions give through the host-callback path
(mcmcx_set_target_host) and through the batched host path (mcmcx_set_target_host_batch, threaded)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# one source, two compilers: a quartic "double banana" with data-dependent weights, a Gaussian prior on some
# components and box bounds -- arithmetic only (+, -, *, /), so that gcc and hipcc with -ffp-contract=off agree exactly
USER_SRC = r"""
#ifdef __HIPCC__
#include "mcmcx_target.h"
#define FN __device__
#else
#define FN
#endif
FN void user_ss(const double *th, int npar, int ny, const void *data, double *ss)
{
    const double *w = (const double *)data;                   /* w[0..npar-1] weights, w[npar] = b */
    const double b = w[npar];
    double s = 0.0;
    for (int k = 0; k + 1 < npar; ++k) {
        double q = th[k + 1] - b * (th[k] * th[k] - 1.0);
        s = s + w[k] * (q * q) + (th[k] * th[k]) / 9.0;
    }
    s = s + w[npar - 1] * (th[npar - 1] * th[npar - 1]);
    ss[0] = s;
    for (int j = 1; j < ny; ++j) ss[j] = s / (double)(j + 1) + th[0] * th[0];
}
FN double user_prior(const double *th, int npar, const void *data)
{
    double p = 0.0;
    for (int k = 0; k < npar; k += 2) { double q = (th[k] - 0.25) / 2.0; p = p + q * q; }
    return p;
}
FN int user_bounds(const double *th, int npar, const void *data)
{
    for (int k = 0; k < npar; ++k) if (!(th[k] > -3.5 && th[k] < 3.0)) return 0;
    return 1;
}
#ifdef __HIPCC__
MCMCX_DEFINE_TARGET(user_target, user_ss, user_prior, user_bounds)
#else
static const void *g_data;
void set_data(const void *d) { g_data = d; }
void host_ss(const double *th, int npar, int ny, double *ss, void *user) { user_ss(th, npar, ny, g_data, ss); }
double host_prior(const double *th, int npar, void *user) { return user_prior(th, npar, g_data); }
int host_bounds(const double *th, int npar, void *user) { return user_bounds(th, npar, g_data); }
void host_ss_batch(const double *th, int npar, int n, int ny, double *ss, void *user)
{ for (int i = 0; i < n; ++i) user_ss(th + (long)i * npar, npar, ny, g_data, ss + (long)i * ny); }
#endif
"""


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    d = tmp_path_factory.mktemp("usermod")
    src = d / "user_target.hip"
    src.write_text(USER_SRC)
    (d / "user_target.c").write_text(USER_SRC)
    hsaco = d / "user_target.hsaco"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O2", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(hsaco)])
    so = d / "libuser_host.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", str(d / "user_target.c"), "-o", str(so)])
    return str(hsaco), C.CDLL(str(so))


CONFIGS = [
    dict(method="dram", drscale=0.0, adaptint=40),
    dict(method="dram", drscale=2.0, adaptint=40),
    dict(method="dram", drscale=3.0, adaptint=30, doburnin=1, burnintime=40, scalelimit=0.3),
    dict(method="ram"),
    dict(method="er", adaptint=40),
    dict(method="scam", adaptint=25),
    dict(method="dram", drscale=2.0, adaptint=40, ny=2),
]


@pytest.mark.parametrize("ci", range(len(CONFIGS)))
def test_user_module_equals_host_callbacks(built, ci):
    from mcmcf90_amd import Engine, make_config, _lib
    hsaco, H = built
    kw = dict(CONFIGS[ci])
    ny = kw.pop("ny", 1)
    npar, nch = 5, 70
    nsimu = 60 if kw["method"] == "scam" else 150
    rng = np.random.default_rng(ci)
    data = np.concatenate([rng.uniform(0.5, 2.0, npar), [0.3]])
    H.set_data(data.ctypes.data_as(C.c_void_p))

    def engine():
        e = Engine(make_config(npar, nch, nsimu=nsimu, updatesigma=1, record_accept=1, chain_id0=11, **kw))
        e.setpar0(np.full(npar, 0.1)); e.setcmat0(0.05 * np.eye(npar))
        e.setsigma2nobs(np.full(ny, 0.8), np.full(ny, 15))
        return e

    runs = {}
    # (1) device module
    e = engine(); e.set_target_module(hsaco, "user_target", data); e.init(); e.run()
    runs["module"] = (e.theta(), e.accept_masks(), e.scalars(), [e.rng(c)[0] for c in (0, 64, 69)]); e.close()
    # (2) host callbacks, one chain at a time
    e = engine()
    ss_t, pri_t, cb_t = _lib.SSFUN_T, _lib.PRIORFUN_T, _lib.CHECKBOUNDS_T
    keep = (C.cast(H.host_ss, ss_t), C.cast(H.host_prior, pri_t), C.cast(H.host_bounds, cb_t))
    assert e.L.mcmcx_set_target_host(e.h, keep[0], keep[1], keep[2], None) == 0
    e.init(); e.run()
    runs["host"] = (e.theta(), e.accept_masks(), e.scalars(), [e.rng(c)[0] for c in (0, 64, 69)]); e.close()
    # (3) batched host callback on three threads
    e = engine()
    keepb = (C.cast(H.host_ss_batch, _lib.SSFUN_BATCH_T), keep[1], keep[2])
    assert e.L.mcmcx_set_target_host_batch(e.h, keepb[0], keepb[1], keepb[2], None, 3) == 0
    e.init(); e.run()
    runs["batch"] = (e.theta(), e.accept_masks(), e.scalars(), [e.rng(c)[0] for c in (0, 64, 69)]); e.close()
    lastmask = np.uint64((1 << (nch - 64)) - 1)                # the ragged tile's unused lanes never see a callback
    for other in ("host", "batch"):
        a, b = runs["module"], runs[other]
        assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)), other
        assert np.array_equal(a[1][:, 0], b[1][:, 0]) and np.array_equal(a[1][:, 1] & lastmask, b[1][:, 1] & lastmask), other
        assert np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64)), other
        assert a[3] == b[3], other
    assert runs["module"][1].any()                             # something was accepted: the comparison is not vacuous


def test_module_errors_are_loud(built, tmp_path):
    from mcmcf90_amd import Engine, make_config, McmcError
    hsaco, _ = built
    e = Engine(make_config(5, 64, nsimu=10))
    with pytest.raises(McmcError, match="cannot load"):
        e.set_target_module(str(tmp_path / "nope.hsaco"), "user_target")
    with pytest.raises(McmcError, match="no kernel named"):
        e.set_target_module(hsaco, "not_there")
    e.close()
    e = Engine(make_config(100, 64, nsimu=10))                 # npar above the module's MCMCX_TARGET_MAX_NPAR (64)
    with pytest.raises(McmcError, match="MCMCX_TARGET_MAX_NPAR"):
        e.set_target_module(hsaco, "user_target")
    e.close()
