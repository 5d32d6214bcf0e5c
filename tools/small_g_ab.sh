# A/B of libhark builds on the single-pass path (G = 4096, 1e9 rows): bash tools/small_g_ab.sh libA.so libB.so
for round in 1 2 3; do for lib in "$@"; do
  HARK_LIB=$PWD/$lib python bench.py --groups 4096 --configs 0 --cpu-rows 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-36s step %.3f ms  kernel %.3f' % ('$lib', d['ms_per_step'], d['hot_path']['by_kernel_ms_per_step']['single']))"
done; done
