#!/bin/bash
# kernel-level timeline of one streaming chunk (eager launches: MIOSQP_POOL_NOGRAPH=1; rocprofv3 --kernel-trace, csv):
# start offsets, durations, gaps of the kernels between the last sweep of one chunk and the first sweep of the next
cd /tmp && export TMPDIR=/tmp
export MIOSQP_POOL_NOGRAPH=1
rm -rf /tmp/ctl && rocprofv3 --kernel-trace --output-format csv -d /tmp/ctl -- python $GRAFT_REPO_ROOT/bench.py --legs batched --no-probes --steps 5 --warmup 2 --pools 1 --stream-warmup 120 --stream-chunks 60 > /tmp/ctl.log 2>&1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('/tmp/ctl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]))
rows.sort()
rel=[i for i,r in enumerate(rows) if 'kp_release' in r[2]]
i0=rel[-12]; i1=rel[-11]
prev=None
t0=rows[i0-3][0]
tot_gap=0; tot_dur=0
for r in rows[i0-3:i1+1]:
    gap = (r[0]-prev) if prev else 0
    if 'kbm_fwd' in r[2] or 'kbm_bwd' in r[2]:
        tag='sweep'
    else:
        tag=''
    print('%9.1f us  dur %8.1f  gap %6.1f  %s' % ((r[0]-t0)/1e3, (r[1]-r[0])/1e3, gap/1e3, r[2]))
    prev=r[1]
PY
