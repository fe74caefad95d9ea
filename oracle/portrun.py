"""oracle/portrun.py -- TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): run one chain of the C restatement of a
BASELINE configuration in a process of its own and print its delayed-rejection count.

    python -m oracle.portrun <workload c2..c5> <nsimu> <adaptint> <chain_id>
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from oracle import pyoracle as po
    from mcmcf90_amd.workloads import problem
    wl, nsimu, adaptint, cid = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ckw, pkw, _ = problem(wl, nsimu, adaptint=adaptint)
    if len(sys.argv) > 5:
        ckw = dict(ckw, method=sys.argv[5])
    o = po.run_chain(po.make_cfg(**ckw), po.Problem(**pkw), chain_id=cid)
    print(o.drtries)
