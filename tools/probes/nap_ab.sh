# the resident search at config 2 (rho 0.1 and rho chosen at set-up) over the poll delay (MIOSQP_COOP_NAP, 64-clock units; "" = the table's)
cd $GRAFT_REPO_ROOT
for nap in ${NAPS:-14 15 16 17 18 19 20}; do
echo "NAP=$nap"
MIOSQP_COOP_NAP=$nap timeout 300 python - <<'PY'
import sys
sys.path.insert(0, "tools/probes")
sys.argv = ["run_ab.py"]
import run_ab
for rho in (0.1, "auto"):
    r = run_ab.one(rho, 1, 300)
    print({k: r[k] for k in ("rho", "nodes_per_s", "iters_per_s", "usec_per_node", "usec_per_node_outside_iterations", "usec_iter_back_to_back")}, flush=True)
PY
done
