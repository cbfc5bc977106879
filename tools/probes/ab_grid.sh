# the reference grid (examples/random_miqp.py, set-up inside the timing) with the current library and with
# miosqp_amd/libmiosqp_hip_base.so (another build, e.g. the previous round's) on the same box
cd $GRAFT_REPO_ROOT
run() { timeout 600 python examples/random_miqp.py --out /tmp/grid.csv 2>&1 | grep "t_avg" | cut -c1-60; }
for rep in 1 2; do
echo "== current"; run
cp miosqp_amd/libmiosqp_hip.so /tmp/cur.so; cp miosqp_amd/libmiosqp_hip_base.so miosqp_amd/libmiosqp_hip.so
echo "== base"; run
cp /tmp/cur.so miosqp_amd/libmiosqp_hip.so
done
