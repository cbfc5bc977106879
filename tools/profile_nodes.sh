# Section 3 of tools/profile_round.sh alone (node launches only): gpurun -- 'bash tools/profile_nodes.sh r02'
set -x
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
db() { find $1 -name "*.db" | head -1; }

cd /tmp
# 3. node launches only: every k_coop dispatch is one node relaxation of the timed workload (no calibration,
#    no back-to-back probes, no other leg); avg duration / iterations per launch follow from this file + its JSON
export MIOSQP_COOP_NAP=18
NODES="--steps 150 --warmup 10 --legs none --no-probes"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_nodes -o n -- python $R/bench.py $NODES > $O/bench_nodes_only.json 2> $O/ks_nodes.err
python $R/tools/rocpd_stats.py $(db /tmp/p_nodes) "MIOSQP_COOP_NAP=18 rocprofv3 --kernel-trace --stats -- python bench.py $NODES   (MI355X, $TAG)" "every k_coop dispatch = one node relaxation (10 warm-up + 150 timed); iterations per launch: see the JSON of the same run" > $O/rocprofv3_kernel_stats_nodes_only.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_nodes_$C -o c -- python $R/bench.py $NODES > $O/pmc_nodes_${C}_bench.json 2> $O/pmc_nodes_$C.err
  python $R/tools/rocpd_pmc.py $(db /tmp/p_nodes_$C) > $O/pmc_nodes_$C.json
done
python $R/tools/pmc_merge.py $O/pmc_nodes_FETCH_SIZE.json $O/pmc_nodes_WRITE_SIZE.json "MIOSQP_COOP_NAP=18 python bench.py $NODES" > $O/pmc_traffic.json
unset MIOSQP_COOP_NAP

