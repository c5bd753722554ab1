#!/bin/bash
# second PMC set: where do the waves wait?  usage: bash tools/collect_pmc2.sh <tag> [perf_probe args]
set -u
TAG=${1:-pmc2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $OUT/$name -o r -- python3 tools/perf_probe.py 131 256 3 smooth 256 > $OUT/$name.log 2>&1
}
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL
run b TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run c SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
run d TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_ATOMIC_WITH_RET_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
run e TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum
run f FETCH_SIZE
run g WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
for n in a b c d e f g; do
  db=$(find $OUT/$n -name "*.db" | head -1)
  echo "== $n"; grep "solves/s" $OUT/$n.log | cut -c1-120
  [ -n "$db" ] && python3 tools/rocpd_pmc.py $db | grep "k_fim" | head -12
done > $OUT/summary.txt 2>&1
