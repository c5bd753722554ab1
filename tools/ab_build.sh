#!/bin/bash
# Build named variants of the library for same-box A/B runs: tools/ab_build.sh name1 "DEFINES1" name2 "DEFINES2" ...
# -> dsurftomo_amd/build/ab/lib_<name>.so; run with DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_<name>.so (boxes differ by ~10 %).
set -e
cd "$(dirname "$0")/.."
mkdir -p dsurftomo_amd/build/ab
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  DSA_DEFINES="$defs" python -m dsurftomo_amd.build --force > /dev/null
  cp dsurftomo_amd/libdsurftomo_amd.so dsurftomo_amd/build/ab/lib_$name.so
  echo "built $name ($defs)"
done
python -m dsurftomo_amd.build --force > /dev/null
