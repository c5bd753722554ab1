#!/bin/bash
# rocprofv3 on the bench command itself: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes.
# usage (GPU box, repo root): bash tools/profile_bench.sh <tag>
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- python3 bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o r -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/write -o r -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $OUT/write.log 2>&1
{
  echo "== kernel trace (rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline)"; grep '"metric"' $OUT/trace.log | cut -c1-400
  python3 tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1)
  for n in fetch write; do echo "== pmc $n"; grep '"metric"' $OUT/$n.log | cut -c1-200; python3 tools/rocpd_pmc.py $(find $OUT/$n -name "*.db" | head -1) | grep "k_fim_sorted<256>"; done
} > $OUT/summary.txt 2>&1
