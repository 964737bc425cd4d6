#!/bin/bash
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700+rep)) tests/dp_rank_main.py --backend gloo --only_epoch ${TRACE:---trace2} 2>/dev/null | grep "^trace\|^{" > /tmp/dp_trace_$rep.txt
  if grep -q '"ok": false' /tmp/dp_trace_$rep.txt; then echo "=== rep $rep FAILED"; cat /tmp/dp_trace_$rep.txt | cut -c1-400; else echo "rep $rep ok"; fi
done
