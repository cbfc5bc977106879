# A/B two builds of the library on the same box: usage ab_lib.sh <probe.py> [args]; compares
# miosqp_amd/libmiosqp_hip.so with miosqp_amd/libmiosqp_hip_base.so (swapped in place, then restored)
set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== current"; timeout 300 python "$@" 2>&1 | tail -2
  cp miosqp_amd/libmiosqp_hip.so /tmp/cur.so; cp miosqp_amd/libmiosqp_hip_base.so miosqp_amd/libmiosqp_hip.so
  echo "== base"; timeout 300 python "$@" 2>&1 | tail -2
  cp /tmp/cur.so miosqp_amd/libmiosqp_hip.so
done
