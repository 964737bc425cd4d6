#!/bin/bash
# diagnostics: part C of tests/dp_rank_main.py (two ranks share GPU 0 through gloo), several times per environment setting
i=0
for cfg in "$@"; do
  for rep in 1 2 3 4 5 6 7 8; do
    i=$((i+1))
    out=$(env $(echo $cfg | tr ',' ' ') timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) tests/dp_rank_main.py --backend gloo --only_epoch 2>/dev/null | grep '^{' | tail -1)
    echo "$cfg rep $rep rc=$? $(echo $out | python -c "import sys,json; d=json.loads(sys.stdin.read() or '{}'); e=d.get('epoch') or {}; print(d.get('ok'), e.get('worst_loss_diff_vs_oracle'), e.get('parameters_identical_across_ranks'), d.get('recurrence_exchange_timeouts'))" 2>/dev/null)"
  done
done
