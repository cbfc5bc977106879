# Round profile: run on the GPU box as  gpurun --timeout 2400 -- 'bash tools/profile_round.sh r03'
# Writes gpurun_out/<tag>/...; the summaries worth keeping are copied into profiles/ (committed) by hand.
# Counters are collected in their own passes (--kernel-trace + --pmc only), as the pool requires.
set -x
TAG=${1:-r04}
ONLY=${2:-}   # optional: one of default nodes_only persistent two_kernel_form cfg5 cfg5x batched all_legs (everything when empty)
want() { [ -z "$ONLY" ] || [ "$ONLY" = "$1" ]; }
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
db() { find $1 -name "*.db" | head -1; }
# kernel table + the two counter passes of one bench command:  prof <name> "<bench args>" "<header>"
prof() {
  NAME=$1; ARGS=$2; HDR=$3
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_$NAME -o s -- python $R/bench.py $ARGS > $O/bench_$NAME.json 2> $O/ks_$NAME.err
  python $R/tools/rocpd_stats.py $(db /tmp/p_$NAME) "$HDR rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (MI355X, $TAG)" "durations in ns; a traced run is slower than an untraced one: the same command untraced is in bench_${NAME}_untraced.json" > $O/rocprofv3_kernel_stats_$NAME.txt
  timeout 400 python $R/bench.py $ARGS > $O/bench_${NAME}_untraced.json 2> /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_${NAME}_$C -o c -- python $R/bench.py $ARGS > $O/pmc_${NAME}_$C.bench.json 2> $O/pmc_${NAME}_$C.err
    python $R/tools/rocpd_pmc.py $(db /tmp/p_${NAME}_$C) > $O/pmc_${NAME}_$C.json
  done
  python $R/tools/pmc_merge.py $O/pmc_${NAME}_FETCH_SIZE.json $O/pmc_${NAME}_WRITE_SIZE.json "$HDR python bench.py $ARGS" $O/pmc_${NAME}_FETCH_SIZE.bench.json > $O/pmc_traffic_$NAME.json
}

# 1. the line the driver records, and the BASELINE.md node budget (999 nodes)
if want default; then
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --steps 999 --legs none > $O/bench_999_nodes.json 2> $O/bench_999.err
# the driver's command (BENCH_rNN.json): 20 timed nodes after 3
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_driver_slice.json 2> $O/bench_driver_slice.err
fi

cd /tmp
# 2. kernel trace of the default command (all legs; CPU baseline skipped: no GPU work in it).
#    MIOSQP_POOL_NOGRAPH=1: the streaming leg launches its chunk kernel by kernel (rocprofv3 dies inside
#    hipGraphLaunch after ~230 replays of that graph); same kernels, same device time
if want all_legs; then
export MIOSQP_POOL_NOGRAPH=1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_all -o all -- python $R/bench.py --no-cpu-baseline > $O/ks_all_bench.json 2> $O/ks_all.err
python $R/tools/rocpd_stats.py $(db /tmp/p_all) "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (MI355X, $TAG, all legs)" "durations in ns; profiled runs are slower than un-profiled ones" > $O/rocprofv3_kernel_stats.txt
unset MIOSQP_POOL_NOGRAPH
fi

# 3. node launches only: every k_coop dispatch is one node relaxation of the timed workload
if want nodes_only; then
prof nodes_only "--steps 150 --warmup 10 --legs none --no-probes" ""
fi

# 4. the streaming forms of the same workload: persistent (one launch per node) and two launches per iteration
if want persistent || want two_kernel_form; then
export MIOSQP_COOP=0
export MIOSQP_PERS=1
prof persistent "--steps 40 --warmup 5 --legs none --no-probes" "MIOSQP_COOP=0 MIOSQP_PERS=1"
export MIOSQP_PERS=0
prof two_kernel_form "--steps 40 --warmup 5 --legs none --no-probes" "MIOSQP_COOP=0 MIOSQP_PERS=0"
unset MIOSQP_COOP MIOSQP_PERS
fi

# 5. config 5 (n=5000: the bandwidth-bound case): the engine's own choice (persistent, tail as S^-1), the persistent
#    form with the two triangular sweeps, four launches per iteration
if want cfg5; then
prof cfg5 "--config cfg5 --steps 12 --warmup 2 --legs none --no-probes" ""
export MIOSQP_PERS=1
prof cfg5_sweeps "--config cfg5 --steps 12 --warmup 2 --legs none --no-probes" "MIOSQP_PERS=1"
export MIOSQP_PERS=0
prof cfg5_four_launches "--config cfg5 --steps 12 --warmup 2 --legs none --no-probes" "MIOSQP_PERS=0"
unset MIOSQP_PERS
fi

# 6. beyond the Infinity Cache: n = 8000 (2 x 512 MB of dense tail per iteration)
if want cfg5x; then
prof cfg5x "--config cfg5x --steps 6 --warmup 1 --legs none --no-probes" ""
export MIOSQP_PERS=0
prof cfg5x_four_launches "--config cfg5x --steps 6 --warmup 1 --legs none --no-probes" "MIOSQP_PERS=0"
unset MIOSQP_PERS
fi

# 7. the batched leg alone (config 3: waves, then the stream on the leaf pool)
if want batched; then
export MIOSQP_POOL_NOGRAPH=1
# ONE pool only (--pools 1): with the two-pool leg in the same table the test kernels' maxima are contention, not the kernel
prof batched "--steps 20 --warmup 5 --legs batched --no-probes --pools 1" "MIOSQP_POOL_NOGRAPH=1"
# ... and the stream by itself (no waves of the wave form in the table: what is not the persistent kernel is the stream's own)
prof batched_stream "--steps 20 --warmup 5 --legs batched --no-probes --pools 1 --batch-waves 0" "MIOSQP_POOL_NOGRAPH=1"
unset MIOSQP_POOL_NOGRAPH
fi
ls -la $O
