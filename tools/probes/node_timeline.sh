#!/bin/bash
# kernel-level timeline of a few consecutive node relaxations (rocprofv3 --kernel-trace, csv): start offsets, durations, gaps
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ntl && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ntl -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --legs none --no-probes > /tmp/ntl.log 2>&1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('/tmp/ntl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
for f in glob.glob('/tmp/ntl/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy '+r.get('Direction','')[:30]))
rows.sort()
coop=[i for i,r in enumerate(rows) if 'k_coop' in r[2]]
i0=coop[-6]; i1=coop[-3]
t0=rows[i0-8][0]
prev=None
for r in rows[i0-8:i1+8]:
    gap = (r[0]-prev) if prev else 0
    print('%9.1f us  dur %8.1f  gap %6.1f  %s' % ((r[0]-t0)/1e3, (r[1]-r[0])/1e3, gap/1e3, r[2]))
    prev=r[1]
PY
