# One box, one run: the bench line, the same command under rocprofv3 (kernel statistics), two PMC passes
# (FETCH_SIZE / WRITE_SIZE, each in its own run), the extra BASELINE configs under rocprofv3, the operator benches,
# kernel timelines + PMC byte counters of the join / sort / sparse group-by statements.
# Usage (through gpurun):  bash tools/evidence.sh TAG      (TAG names THIS run, e.g. r04a: output in gpurun_out/ev_TAG)
#        then locally:     python tools/collect_evidence.py r04 gpurun_out/ev_TAG
# gpurun MERGES what a call wrote into the local gpurun_out/: a directory name used twice would hold two runs' files, so
# the script refuses a directory that exists already (on the box nothing does; locally a stale one must not be reused).
set -x
export TMPDIR=/tmp
TAG=${1:?usage: evidence.sh TAG}
O=gpurun_out/ev_$TAG
if [ -e "$O" ]; then echo "$O exists: pick a new TAG" >&2; exit 2; fi
# measure the library of THESE sources or nothing (an experimental build of edited sources once stayed in the tree and an evidence
# run measured it: profiles/r04_notes.md 13); on the box the sources and the library arrive together with the snapshot
python3 harkdb_amd/_srchash.py || exit 2
mkdir -p $O
date -u +%FT%TZ > $O/run_id.txt; sha256sum bench.py harkdb_amd/libhark.so >> $O/run_id.txt
timeout 900 python bench.py > $O/bench_plain.json 2> $O/bench_plain.err
# headline only (--configs 0): every launch of the fused kernels in these profiles is a headline launch
# (under the profiler bench.py launches no nested --pmc children -- --pmc 0 says so explicitly -- and no setup launch: the
# statistics average steps + warm-up full launches per kernel, nothing else)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --cpu-rows 0 --configs 0 --pmc 0 > $O/bench_rocprof.json 2> $O/bench_rocprof.err
# the PMC passes run the --pmc-child mode: warm-up + timed steps only, no setup launch (every launch is a full pass)
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --pmc-child 1 --steps 2 --warmup 1 --cpu-rows 0 --configs 0 --pmc 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --pmc-child 1 --steps 2 --warmup 1 --cpu-rows 0 --configs 0 --pmc 0 > $O/pmc_write.json 2> $O/pmc_write.err
# the extra configs (small-G single pass, C2, C4, C5, sorts, sparse group-by) under the profiler: per-kernel time of those pipelines
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_configs -- python3 bench.py --steps 2 --warmup 1 --cpu-rows 0 --pmc 0 > $O/bench_configs_rocprof.json 2> $O/bench_configs_rocprof.err
timeout 600 python tools/ops_bench.py > $O/ops_bench.log 2>&1
timeout 300 python tools/c5_bench.py > $O/c5_bench.log 2>&1
# bytes moved per kernel of the operators furthest from their roofline (one counter per run)
for w in join_c4 join_u32 sort20 sort32 sort64 sparse_gb sparse_five refgb_hash; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/opmc_${w}_$c -- python3 tools/op_one.py $w > $O/opmc_${w}_$c.log 2>&1
  done
done
# round 6: the join under skew, the reference's own statements on its 7-row table, the no-filter producer, the SQL stress
# (OR / NOT / IN trees against pandas), SQ counters of the join's and the i64 sort's kernels, what one rank of a strong-scaling
# run does (round 5's probes -- LDS side of a hash probe, wide digits, the vendor sort -- are not repeated: profiles/r05_*)
timeout 300 python tools/join_skew_probe.py > $O/join_skew.txt 2>&1
# ... and the tables kept in key order: the join's search path, the GROUP BY's window path, every statement form on a sorted table;
# kernel timelines of the two new paths (the probes' first shapes only)
timeout 300 python tools/join_cluster_probe.py > $O/join_cluster.txt 2>&1
timeout 400 python tools/groupby_cluster_probe.py 1e9 > $O/groupby_cluster.txt 2>&1
timeout 300 python tools/statement_cluster_probe.py > $O/statement_cluster.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_cjoin -- python3 tools/join_cluster_probe.py sorted match_sorted > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_window -- python3 tools/groupby_cluster_probe.py 1e9 sorted > /dev/null 2>&1
{ echo "== tools/join_cluster_probe.py sorted match_sorted (three joins each)"; python tools/ktrace.py $O/trace_cjoin 0.17 | head -14; echo "== tools/groupby_cluster_probe.py 1e9 sorted"; python tools/ktrace.py $O/trace_window 0.05 | head -10; } > $O/cluster_traces.txt 2>&1
timeout 100 python tools/small_latency.py > $O/small_latency.txt 2>&1
timeout 200 python tools/nofilter_ab.py > $O/nofilter_ab.txt 2>&1
timeout 200 python tools/ingest_bench.py > $O/ingest_bench.log 2>&1
timeout 300 bash tools/strong_rehearsal.sh > $O/strong_rehearsal.txt 2>&1
timeout 400 python tools/sql_stress.py 240 6 > $O/sql_stress.txt 2>&1
timeout 400 bash tools/pmc_kernel.sh jorder_kernel,jbucket_kernel,jpart_kernel op_one.py join_c4 1.0 2 > $O/pmc_join_c4.txt 2>&1
timeout 400 bash tools/pmc_kernel.sh msd_final_kernel,msd_part_kernel op_one.py sort64 1.0 2 > $O/pmc_sort64.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -80
cat $O/bench_plain.json
