# kernel times of the C4 join share (and of the experiment builds named on the command line: HARK_JB_EXP variants made with
# tools/ab_build.sh jbN "-DHARK_JB_EXP=N" k_hjoin.hip -- their results are wrong by construction, only the times count)
export TMPDIR=/tmp
for e in base "$@"; do
  if [ $e = base ]; then unset HARK_LIB; else export HARK_LIB=$PWD/harkdb_amd/libhark_$e.so; fi
  timeout 100 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/jbx -- python3 tools/join_c4.py > gpurun_out/jbx.log 2>&1
  echo "== $e"; grep "C4 join" gpurun_out/jbx.log | tail -1; python tools/jtrace.py gpurun_out/jbx | sed -n "/---- per/,\$p" | grep -E "jbucket|jorder|jpart"
  rm -rf gpurun_out/jbx
done
