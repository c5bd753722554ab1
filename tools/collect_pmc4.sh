#!/bin/bash
# fourth PMC set (short timeouts: some TA/TCP counters hang on this pool): how busy is the per-CU vector memory path?
set -u
TAG=${1:-pmc4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout 150 rocprofv3 --pmc "$@" -d $OUT/$name -o r -- python3 tools/perf_probe.py 131 512 0.4 smooth 256 > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/rc.txt
}
run a TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum
run b TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
run c TA_BUSY_avr TA_TA_BUSY_sum
for n in a b c; do
  db=$(find $OUT/$n -name "*.db" | head -1)
  echo "== $n"; grep "solves/s" $OUT/$n.log | cut -c1-120
  [ -n "$db" ] && python3 tools/rocpd_pmc.py $db | grep "k_fim" | head -12
done > $OUT/summary.txt 2>&1
cat $OUT/rc.txt $OUT/summary.txt
