import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, util, cases
hip, oracle = util.Impl("hip"), util.Impl("oracle")
for dims in [(24, 13, 11), (24, 16, 16), (32, 13, 11), (20, 13, 11), (44, 30, 18), (48, 30, 18), (40, 24, 16)]:
    flags, A, _ = cases.system_inputs(dims, 5)
    rhs = cases.cg_rhs(dims, flags, 5)
    for env in ({}, ):
        x, st = cases.run_cg_impl(hip, dims, flags, A, rhs, 2, 1e-3, 60, 0)
        xo, sto = cases.run_cg_impl(oracle, dims, flags, A, rhs, 2, 1e-3, 60, 0)
        print(dims, "iters", st[0], sto[0], "relerr %.3e" % util.rel_err(x, xo), flush=True)
