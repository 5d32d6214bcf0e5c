# A/B build of libhark: bash tools/ab_build.sh NAME "-DFLAG ..." [unit.hip | /path/to/edited_copy_of_unit.hip ...]  ->  build/ab/libhark_NAME.so
# (NOT into harkdb_amd/: every file there travels to the GPU box with each call, and HARK_LIB can swap any of them for the product --
# tests/test_abi.py fails when a second libhark*.so sits beside the product library.  Run with HARK_LIB=$PWD/build/ab/libhark_NAME.so)
# Recompiles the named units (default k_fgb.hip) with the extra flags and links them with the product's other objects.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
UNITS=${@:-k_fgb.hip}
cd "$(dirname "$0")/../harkdb_amd/csrc"
# the product's objects as they are: NEVER edit a product source for an experiment and run this (make would rebuild the product
# from the edit) -- copy the unit elsewhere, edit the copy and name it with its path (compiled with -I to these headers)
if ! git diff --quiet -- . ; then echo "harkdb_amd/csrc has uncommitted edits: experiments go into a copy of the unit (see above)" >&2; exit 2; fi
make -j8 >/dev/null
OBJ=/tmp/hark_ab_$NAME; mkdir -p $OBJ
EXCL=""
for u in $UNITS; do
  base=$(basename $u)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-value -Wno-unused-result -I. -I../../include $FLAGS -c $u -o $OBJ/${base%.hip}.o
  EXCL="$EXCL ${base%.hip}.o"
done
OTHERS=$(ls *.o | grep -v -x -F "$(echo $EXCL | tr ' ' '\n')")
mkdir -p ../../build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/libhark_$NAME.so $OTHERS $(for u in $UNITS; do base=$(basename $u); echo $OBJ/${base%.hip}.o; done)
echo built build/ab/libhark_$NAME.so
