#!/usr/bin/env python3
"""Debug aid: the goldens' pairs through g2g_pwdm_create_batch one parameter set at a time, names printed as they go."""
import faulthandler, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.dump_traceback_later(150, exit=True)
import numpy as np
from prrn_aln_amd import engine, operator as op
from test_host_builders import groups_from_golden, params_from_golden
from test_gpu_builders import same_problem
ctx = engine.Context()
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    d = dict(np.load(path)); alp = params_from_golden(d)
    print(os.path.basename(path), "tgapf", alp.tgapf, "molc", alp.molc, flush=True)
    h = op.PwdM(list(groups_from_golden(d, alp)), alp)
    g = op.PwdM.batch(ctx, [list(groups_from_golden(d, alp))], alp)[0]
    same_problem(g.problem, h.problem, path)
print("all goldens identical")
