cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
for st in 20 200; do
PASTURE_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps $st --warmup 3 --no-cpu-baseline --no-north-star --no-configs3 --collective capi 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('capi steps $st', d['ms_per_step'], d['roofline']['frac'])"
done
