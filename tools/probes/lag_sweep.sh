#!/bin/bash
# hosted search rate at config 2 against the pick-up lag of the testers' decision (MIOSQP_COOP_LAG), with node stamps
for lag in 10 12 14 16 18; do
  echo "== lag $lag"
  MIOSQP_COOP_LAG=$lag MIOSQP_SEARCH_STAMPS=1 python $GRAFT_REPO_ROOT/tools/probes/hosted_rate.py 300 2 2>&1 | tail -3
done
