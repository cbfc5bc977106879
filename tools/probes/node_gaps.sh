#!/bin/bash
# kernel-level accounting of the hosted search at config 2 (rocprofv3 --kernel-trace): per node, the cooperative launch's
# duration, the iterations it reports, the gap to the next launch; usage: node_gaps.sh <label> [ENV=VAL ...]
label=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/ng_$label
rocprofv3 --kernel-trace --output-format csv -d /tmp/ng_$label -- python $GRAFT_REPO_ROOT/tools/probes/hosted_rate.py 200 2 > /tmp/ng_$label.log 2>&1
tail -2 /tmp/ng_$label.log
python3 - "$label" <<'PY'
import csv, glob, sys
label = sys.argv[1]
rows = []
for f in glob.glob('/tmp/ng_%s/**/*kernel_trace.csv' % label, recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
coop = [i for i, r in enumerate(rows) if 'k_coop' in r[2]]
coop = coop[len(coop) // 2:]  # the second half: steady state
dur = [(rows[i][1] - rows[i][0]) / 1e3 for i in coop]
gaps, others = [], []
for a, b in zip(coop[:-1], coop[1:]):
    gaps.append((rows[b][0] - rows[a][1]) / 1e3)
    others.append(sum((rows[k][1] - rows[k][0]) / 1e3 for k in range(a + 1, b)))
import statistics as st
print("%s: %d launches  k_coop mean %.1f us  between two launches: mean %.1f us (median %.1f), of which other kernels %.1f us"
      % (label, len(dur), st.mean(dur), st.mean(gaps), st.median(gaps), st.mean(others)))
names = {}
for a, b in zip(coop[:-1], coop[1:]):
    for k in range(a + 1, b):
        nm = rows[k][2].split('(')[0][-40:]
        names.setdefault(nm, []).append((rows[k][1] - rows[k][0]) / 1e3)
for nm, v in names.items():
    print("   %-42s x%.2f per node, %.1f us each" % (nm, len(v) / max(1, len(gaps)), st.mean(v)))
PY
