T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --no-cpu-baseline"
P="python bench.py --no-cpu-baseline"
f() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', '%.4f'%d['ms_per_step'], '%.4g'%d['value'])"; }
$T --steps 30 --warmup 5 2>/dev/null | f torchrun30
$T --steps 30 --warmup 5 --gather-every 64 2>/dev/null | f torchrun30_G64
$T --steps 300 --warmup 5 2>/dev/null | f torchrun300
$T --steps 300 --warmup 5 --gather-every 64 2>/dev/null | f torchrun300_G64
$P --steps 30 --warmup 5 | f plain30
$P --steps 300 --warmup 5 | f plain300
