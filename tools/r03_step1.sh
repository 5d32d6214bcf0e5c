set -x
export TMPDIR=/tmp
O=gpurun_out/r03s1; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fgb.py tests/test_gpu_hjoin.py tests/test_gpu_bench_rank.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout -k 10 300 python tools/fgb_ablate.py 1e9 "shift=12;shift=13;pairfmt=3" > $O/ablate.log 2>&1 && cat $O/ablate.log
timeout -k 10 400 bash tools/ab.sh harkdb_amd/libhark.so harkdb_amd/libhark_k96.so > $O/ab_k96.log 2>&1 && cat $O/ab_k96.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03s1/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_measured_in_run'], d['roofline'].get('traffic_note'))
print(d['check'])
for k,v in d.get('configs',{}).items():
    if k.startswith('SWEEP'):
        for kk,vv in v.items(): print(kk, round(vv['ms'],3), round(vv['frac_of_peak'],3), vv['kernel_ms'])
    elif isinstance(v, dict) and 'ms' in v: print(k, round(v['ms'],3), round(v['frac_of_peak'],3))
    else: print(k, v)
PY
