cd $GRAFT_REPO_ROOT
for v in 0 11 22 33 44; do echo "== MIOSQP_BM_VAR=$v"; MIOSQP_BM_VAR=$v python tools/probes/kbm_time.py ${1:-256}; done
