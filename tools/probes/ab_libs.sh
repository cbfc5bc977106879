#!/bin/bash
# A/B of several builds of the library on one box: tools/probes/ab_libs.sh name1 name2 ...  (tools/probes/_bin/<name>.so)
cp miosqp_amd/libmiosqp_hip.so /tmp/cur.so
for n in "$@"; do
  cp tools/probes/_bin/$n.so miosqp_amd/libmiosqp_hip.so
  echo "== $n"
  python tools/probes/coop_phases.py 2>&1 | tail -3 | grep -v max
  for nap in ${NAPS:-16 18 20}; do echo "nap $nap $(MIOSQP_COOP_NAP=$nap python bench.py --legs none --no-probes --steps 150 --warmup 20 2>/dev/null | cut -c88-110)"; done
done
cp /tmp/cur.so miosqp_amd/libmiosqp_hip.so
