# hosted search at config 2 over the poll delay of the cooperative solver (MIOSQP_COOP_NAP, 64-clock units)
cd $GRAFT_REPO_ROOT
for nap in "" 10 12 14 16 18 20 22 24; do
  echo "NAP=$nap: $(MIOSQP_COOP_NAP=$nap timeout 300 python tools/probes/hosted_rate.py 300 2 2>&1 | tail -1)"
done
