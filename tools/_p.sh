export TMPDIR=/tmp
O=gpurun_out/joinprof; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/join_one.py > $O/log.txt 2>&1
python tools/kstats.py $O 20
