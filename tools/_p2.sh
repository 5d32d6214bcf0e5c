mkdir -p gpurun_out/r05m
for w in sort64 join_c4; do for e in - HARK_SORT_NO_MSD=1 - HARK_SORT_NO_MSD=1; do
  if [ "$e" != "-" ]; then export $e; fi
  echo "== $w $e" >> gpurun_out/r05m/ab.txt; timeout -k 10 200 python tools/op_one.py $w 1 4 2>&1 | tail -n 2 >> gpurun_out/r05m/ab.txt || exit 1
  if [ "$e" != "-" ]; then unset ${e%%=*}; fi
done; done
cat gpurun_out/r05m/ab.txt
