# PMC counters of the batched sweep kernels: gpurun -- 'bash tools/probes/pmc_kbm.sh [cols]'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
COLS=${1:-256}
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr TD_BUSY_avr"; do
  rm -rf /tmp/pk
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pk -o p -- python $R/tools/probes/kbm_time.py $COLS > /dev/null 2> /tmp/pk.err || tail -3 /tmp/pk.err
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pk/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%kbm_fwd%' or kernel_name like '%kbm_bwd%' group by kernel_name, counter_name").fetchall()
for r in rows:
    print("%-10s %-34s n=%4d mean=%.4g" % (r[0].split('::')[-1][:10], r[1], r[2], r[3]))
PY
done
