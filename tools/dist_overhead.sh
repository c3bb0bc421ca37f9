# per-step time of the bench under torch.distributed.run (1 rank) for several gather groupings
for g in 1 8 1000; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --gather-every $g 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('gather-every', $g, 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
