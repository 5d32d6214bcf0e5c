# What ONE rank of the strong-scaling run does at N = 2, 4, 8: rows shard_range(1e9, 1, N) of the 1B-row table through bench.py as
# a single RCCL rank with the pipelined two-plan step forced (the code path of the N > 1 runs minus the peers).  The kernels'
# time per step is what an N-GPU strong-scaling step cannot go below; the merge (16 MiB all-reduce) comes on top or hides.
#   bash tools/strong_rehearsal.sh  ->  one line per N
for N in 1 2 4 8; do
  ROWS=$((1000000000 / N)); FIRST=$((ROWS * (N > 1 ? 1 : 0)))
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + N)) HARK_FORCE_PIPELINE=1 \
  python3 bench.py --gpus 1 --rows $ROWS --first-row $FIRST --steps 20 --warmup 5 --cpu-rows 0 --configs 0 --pmc 0 --tolerance-check 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']; h=d['hot_path']
print('N=$N rows/rank', c['rows_this_rank'], 'step %.3f ms' % d['ms_per_step'], 'kernels %.3f' % h['kernel_ms_per_step'], h['by_kernel_ms_per_step'], 'chosen:', c['producer_workgroups'], 'pipelined' if c['pipelined_steps'] else 'serial', c['allreduce'], 'measured', {k: round(v,3) for k,v in (c['measured_at_startup_ms_per_step'] or {}).items()}, 'frac %.3f' % d['roofline']['frac'], d['check'])
"
done
