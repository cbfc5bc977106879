# Round profile: run on the GPU box as  gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
# Writes gpurun_out/<tag>/...; the summaries worth keeping are copied into profiles/ (committed) by hand.
# Counters are collected in their own passes (--kernel-trace + --pmc only), as the pool requires.
set -x
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
db() { find $1 -name "*.db" | head -1; }

# 1. the line the driver records
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err

cd /tmp
# 2. kernel trace of the same default command (all legs; CPU baseline skipped: no GPU work in it).
#    MIOSQP_POOL_NOGRAPH=1: the streaming leg launches its chunk kernel by kernel (rocprofv3 dies inside
#    hipGraphLaunch after ~230 replays of that graph); same kernels, same device time
export MIOSQP_POOL_NOGRAPH=1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_all -o all -- python $R/bench.py --no-cpu-baseline > $O/ks_all_bench.json 2> $O/ks_all.err
python $R/tools/rocpd_stats.py $(db /tmp/p_all) "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (MI355X, $TAG, all legs)" "durations in ns; profiled runs are slower than un-profiled ones" > $O/rocprofv3_kernel_stats.txt

unset MIOSQP_POOL_NOGRAPH

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

# 4. the HBM-streaming form of the same workload (two launches per iteration)
export MIOSQP_COOP=0
STREAM="--steps 40 --warmup 5 --legs none --no-probes"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_str -o s -- python $R/bench.py $STREAM > $O/bench_two_kernel_form.json 2> $O/ks_str.err
python $R/tools/rocpd_stats.py $(db /tmp/p_str) "MIOSQP_COOP=0 rocprofv3 --kernel-trace --stats -- python bench.py $STREAM   (MI355X, $TAG, HBM-streaming form)" > $O/rocprofv3_kernel_stats_two_kernel_form.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_str_$C -o c -- python $R/bench.py $STREAM > /dev/null 2> $O/pmc_str_$C.err
  python $R/tools/rocpd_pmc.py $(db /tmp/p_str_$C) > $O/pmc_str_$C.json
done
python $R/tools/pmc_merge.py $O/pmc_str_FETCH_SIZE.json $O/pmc_str_WRITE_SIZE.json "MIOSQP_COOP=0 python bench.py $STREAM" > $O/pmc_traffic_two_kernel_form.json
unset MIOSQP_COOP

# 5. config 5 (n=5000: the bandwidth-bound case), factor form, four launches per iteration
C5="--config cfg5 --steps 12 --warmup 2 --legs none --no-probes"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o s -- python $R/bench.py $C5 > $O/bench_cfg5.json 2> $O/ks_c5.err
python $R/tools/rocpd_stats.py $(db /tmp/p_c5) "rocprofv3 --kernel-trace --stats -- python bench.py $C5   (MI355X, $TAG, config 5)" "launches queued behind a decided test are listed as [early exit]" > $O/rocprofv3_kernel_stats_cfg5.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_c5_$C -o c -- python $R/bench.py $C5 > /dev/null 2> $O/pmc_c5_$C.err
  python $R/tools/rocpd_pmc.py $(db /tmp/p_c5_$C) > $O/pmc_c5_$C.json
done
python $R/tools/pmc_merge.py $O/pmc_c5_FETCH_SIZE.json $O/pmc_c5_WRITE_SIZE.json "python bench.py $C5" > $O/pmc_traffic_cfg5.json

# 6. the batched leg alone (config 3: waves, then the stream on the leaf pool)
export MIOSQP_POOL_NOGRAPH=1
B3="--steps 20 --warmup 5 --legs batched --no-probes"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_b3 -o s -- python $R/bench.py $B3 > $O/bench_batched.json 2> $O/ks_b3.err
python $R/tools/rocpd_stats.py $(db /tmp/p_b3) "rocprofv3 --kernel-trace --stats -- python bench.py $B3   (MI355X, $TAG, config 3: 256 leaves in flight, waves then stream; MIOSQP_POOL_NOGRAPH=1)" > $O/rocprofv3_kernel_stats_batched.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_b3_$C -o c -- python $R/bench.py $B3 > /dev/null 2> $O/pmc_b3_$C.err
  python $R/tools/rocpd_pmc.py $(db /tmp/p_b3_$C) > $O/pmc_b3_$C.json
done
python $R/tools/pmc_merge.py $O/pmc_b3_FETCH_SIZE.json $O/pmc_b3_WRITE_SIZE.json "python bench.py $B3" > $O/pmc_traffic_batched.json
ls -la $O
