# which part of the torch.distributed set-up costs per-step time at one rank?
run() { "$@" 2>/dev/null | tail -1 > /tmp/o.json; python -c "import json,sys; d=json.load(open('/tmp/o.json')); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'])" "$LABEL"; }
LABEL="plain"; run python bench.py --no-cpu-baseline --gather-every 1000
LABEL="env-only (RANK/WORLD_SIZE/MASTER_ADDR, no torchrun)"; RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 run python bench.py --no-cpu-baseline --gather-every 1000
LABEL="torchrun"; run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --gather-every 1000
LABEL="torchrun OMP=8"; OMP_NUM_THREADS=8 run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --no-cpu-baseline --gather-every 1000
LABEL="torchrun NCCL_MAX_NCHANNELS=1"; NCCL_MAX_NCHANNELS=1 run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --no-cpu-baseline --gather-every 1000
