# A/B build of libhark: bash tools/ab_build.sh NAME "-DFLAG ..." [unit.hip ...]  ->  harkdb_amd/libhark_NAME.so
# Recompiles the named units (default k_fgb.hip) with the extra flags and links them with the product's other objects.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
UNITS=${@:-k_fgb.hip}
cd "$(dirname "$0")/../harkdb_amd/csrc"
make -j8 >/dev/null
OBJ=/tmp/hark_ab_$NAME; mkdir -p $OBJ
EXCL=""
for u in $UNITS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-value -Wno-unused-result $FLAGS -c $u -o $OBJ/${u%.hip}.o
  EXCL="$EXCL ${u%.hip}.o"
done
OTHERS=$(ls *.o | grep -v -x -F "$(echo $EXCL | tr ' ' '\n')")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhark_$NAME.so $OTHERS $(for u in $UNITS; do echo $OBJ/${u%.hip}.o; done)
echo built harkdb_amd/libhark_$NAME.so
