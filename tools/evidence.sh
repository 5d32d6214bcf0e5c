# One box, one run: the bench line, the same command under rocprofv3 (kernel statistics), two PMC passes
# (FETCH_SIZE / WRITE_SIZE, each in its own run), the extra BASELINE configs under rocprofv3, the operator benches.
# Usage (through gpurun):  bash tools/evidence.sh   ; then locally:  python tools/collect_evidence.py r03
set -x
export TMPDIR=/tmp
O=gpurun_out/ev3; rm -rf $O; mkdir -p $O   # gpurun merges into the local copy: stale files of earlier runs are removed below
timeout 600 python bench.py > $O/bench_plain.json 2> $O/bench_plain.err
# headline only (--configs 0): every launch of the fused kernels in these profiles is a headline launch
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --cpu-rows 0 --configs 0 --pmc 0 > $O/bench_rocprof.json 2> $O/bench_rocprof.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-rows 0 --configs 0 --pmc 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-rows 0 --configs 0 --pmc 0 > $O/pmc_write.json 2> $O/pmc_write.err
# the extra configs (small-G single pass, C2, C4, C5) under the profiler: per-kernel time of those pipelines
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_configs -- python3 bench.py --steps 2 --warmup 1 --cpu-rows 0 --pmc 0 > $O/bench_configs_rocprof.json 2> $O/bench_configs_rocprof.err
timeout 600 python tools/ops_bench.py > $O/ops_bench.log 2>&1
timeout 300 python tools/c5_bench.py > $O/c5_bench.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -50
cat $O/bench_plain.json
