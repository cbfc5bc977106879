cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 150 --warmup 10 --legs none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('nodes_per_s'))"; }
for rep in 1 2; do
  echo "== current AHEAD=1"; run
  echo "== current AHEAD=0"; MIOSQP_COOP_AHEAD=0 run
  cp miosqp_amd/libmiosqp_hip.so /tmp/cur.so; cp miosqp_amd/libmiosqp_hip_base.so miosqp_amd/libmiosqp_hip.so
  echo "== base"; run
  cp /tmp/cur.so miosqp_amd/libmiosqp_hip.so
done
