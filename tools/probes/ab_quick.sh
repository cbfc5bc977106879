#!/bin/bash
# quick pass/fail of several builds: tools/probes/ab_quick.sh name1 name2 ...
cp miosqp_amd/libmiosqp_hip.so /tmp/cur.so
for n in "$@"; do
  cp tools/probes/_bin/$n.so miosqp_amd/libmiosqp_hip.so
  echo "== $n: $(timeout 300 python -m pytest tests -m gpu -x -q -k 'cooperative_solver_equals' 2>&1 | tail -1)"
  echo "   bench: $(timeout 300 python bench.py --legs none --no-probes --steps 100 --warmup 10 2>&1 | tail -1 | cut -c88-110)"
done
cp /tmp/cur.so miosqp_amd/libmiosqp_hip.so
