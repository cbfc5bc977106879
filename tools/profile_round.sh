set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r01b
mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/ks_bench.json 2> $O/ks.err
DB=$(find /tmp/prof_ks -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (MI355X, round 1, cooperative solver)" "durations in ns; profiled runs are slower than un-profiled ones" > $O/rocprofv3_kernel_stats.txt
CMD="--steps 40 --warmup 5 --no-cpu-baseline --batch-waves 2 --no-large-leg"
# the poll delay is pinned to the value the calibration picks at this size, so that every k_coop dispatch
# counted below is a node relaxation (the calibration launches would dilute the per-dispatch mean)
export MIOSQP_COOP_NAP=16
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $GRAFT_REPO_ROOT/bench.py $CMD > $O/pmc_f_bench.json 2> $O/pmc_f.err
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find /tmp/prof_f -name "*.db" | head -1) > $O/pmc_fetch.json
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $GRAFT_REPO_ROOT/bench.py $CMD > $O/pmc_w_bench.json 2> $O/pmc_w.err
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find /tmp/prof_w -name "*.db" | head -1) > $O/pmc_write.json
python $GRAFT_REPO_ROOT/tools/pmc_merge.py $O/pmc_fetch.json $O/pmc_write.json "MIOSQP_COOP_NAP=16 python bench.py $CMD" > $O/pmc_traffic.json
ls -la $O
