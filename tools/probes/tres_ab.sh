#!/bin/bash
# the testers' rows of Kc in registers (MIOSQP_COOP_TRES) against rows from memory, and the pick-up lag that goes with it
for cfg in "0 16" "1 16" "1 10" "1 8" "1 6"; do
  set -- $cfg
  echo "== resident rows $1, lag $2"
  MIOSQP_COOP_TRES=$1 MIOSQP_COOP_LAG=$2 MIOSQP_SEARCH_STAMPS=1 python $GRAFT_REPO_ROOT/tools/probes/hosted_rate.py 300 2 2>&1 | tail -3 | cut -c1-520
done
