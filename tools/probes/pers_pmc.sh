# Memory-side traffic and L2 hit rate of the persistent streaming solver on config 2 (run on the GPU box):
#   gpurun --timeout 900 -- 'bash tools/probes/pers_pmc.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${PERS_ARGS:-"500 1000 250 0.7 1"}
cat > /tmp/pers_one.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from miosqp_amd import qp, problems
n, m, p, dens, fold = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
pr = problems.random_miqp(n, m, p, density=dens, seed=0)
A, l, u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr["P"], pr["q"], A, l, u, fold=fold, coop=0, resident=0, pers=1, **problems.QP_SETTINGS)
g.warm_start(x=np.zeros(n), y=np.zeros(A.shape[0]))
print("us/it", g.time_kernel(4, 1000)[0])
PY
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  D=/tmp/pp_$(echo $C | tr ' ' '_')
  rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $D -o c -- python /tmp/pers_one.py $ARGS 2>&1 | grep "us/it"
  python $R/tools/rocpd_pmc.py $(find $D -name "*.db" | head -1) | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'k_pers' in k:
        for c,r in v.items(): print(k[:40], c, 'dispatches', r['dispatches'], 'min %.4g max %.4g' % (r['min'], r['max']))
"
done
