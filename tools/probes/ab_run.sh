# current library against miosqp_amd/libmiosqp_hip_base.so on ONE box: the hosted search at config 2 with the resident grid
# (tools/probes/run_ab.py, both settings of rho), alternating, REPS times (default 3)
cd $GRAFT_REPO_ROOT
one() { timeout 300 python - "$@" <<'PY'
import sys
sys.path.insert(0, "tools/probes")
sys.argv = ["run_ab.py"]
import run_ab
for rho in (0.1, "auto"):
    r = run_ab.one(rho, 1, 300)
    print({k: r[k] for k in ("rho", "nodes_per_s", "iters_per_s", "usec_per_node", "usec_per_node_outside_iterations", "usec_iter_back_to_back", "node_us_per_iter_min_med_max")}, flush=True)
PY
}
for rep in $(seq ${REPS:-3}); do
  echo "== current"; one
  cp miosqp_amd/libmiosqp_hip.so /tmp/cur.so; cp miosqp_amd/libmiosqp_hip_base.so miosqp_amd/libmiosqp_hip.so
  echo "== base"; one
  cp /tmp/cur.so miosqp_amd/libmiosqp_hip.so
done
