# A/B of libhark builds on ONE box: bash tools/ab.sh libA.so libB.so ...   (each twice, interleaved; headline only)
for round in 1 2 3; do
for lib in "$@"; do
  HARK_LIB=$PWD/$lib python bench.py --configs 0 --cpu-rows 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['hot_path']['by_kernel_ms_per_step']
print('%-40s step %.3f ms  producer %.3f  consumer %.3f  read-probe %.3f' % ('$lib', d['ms_per_step'], k['producer'], k['consumer'], d['roofline']['probes_ms']['read_3_columns']))"
done; done
