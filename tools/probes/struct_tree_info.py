import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, structured_problems as sp
from miosqp_amd import bnb, problems
for name in ["milp", "equality_rows", "one_sided_rows", "A_5pct", "badly_scaled", "power_converter_K10","low_rank_P","A_1pct","power_converter_K20"]:
    pr=sp.make(name)
    st=dict(problems.BNB_SETTINGS, max_iter_bb=40, device_tree=False)
    m=bnb.MIOSQP(); m.setup(pr["P"],pr["q"],pr["A"],pr["l"],pr["u"],pr["i_idx"],pr["i_l"],pr["i_u"],dict(st,device_search=True),dict(problems.QP_SETTINGS))
    r=m.solve()
    print(name, r.status, m.work.iter_num, m.work.osqp_iter, r.upper_glob, m.work.lower_glob)
