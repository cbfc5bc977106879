# current vs base library: hosted search at config 2 with the per-node device stamps (MIOSQP_SEARCH_STAMPS=1)
cd $GRAFT_REPO_ROOT
run() { MIOSQP_SEARCH_STAMPS=1 timeout 300 python tools/probes/hosted_rate.py 200 1 2>&1 | tail -3; }
echo "== current AHEAD=0"; MIOSQP_COOP_AHEAD=0 run
echo "== current AHEAD=1"; run
cp miosqp_amd/libmiosqp_hip.so /tmp/cur.so; cp miosqp_amd/libmiosqp_hip_base.so miosqp_amd/libmiosqp_hip.so
echo "== base"; run
cp /tmp/cur.so miosqp_amd/libmiosqp_hip.so
