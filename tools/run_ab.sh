# A/B of library builds on one box: bash tools/run_ab.sh <tag> "<lib names>" "<windows>" [units] [kind]
set -u
TAG=$1; LIBS=$2; WIN=${3:-1.25}; UNITS=${4:-1024}; KIND=${5:-smooth}
O=gpurun_out/$TAG; mkdir -p $O
first=""
for l in $LIBS; do
  if [ -z "$first" ]; then first=$l; SAVE="DSA_SAVE=$O/ref_$KIND.npy"; else SAVE="DSA_COMPARE=$O/ref_$KIND.npy"; fi
  env $SAVE DSA_LIB_PATH=dsurftomo_amd/build/ab/lib_$l.so python3 tools/perf_probe.py 131 $UNITS $WIN $KIND 256 > $O/$l.$KIND.log 2>&1
  echo "== $l"; grep -v "phase share" $O/$l.$KIND.log | cut -c1-215
done | tee $O/summary_$KIND.txt
