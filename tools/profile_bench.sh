#!/bin/bash
# rocprofv3 on the bench command itself: kernel trace + stats, then the counter groups in their own passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  usage (GPU box, repo root): bash tools/profile_bench.sh <tag>
# Leaves gpurun_out/<tag>/summary.txt and gpurun_out/<tag>/pmc_latest.json: copy them to profiles/ (r0N_rocprof_bench.txt,
# pmc_latest.json) -- bench.py reads profiles/pmc_latest.json and ignores it when the kernel sources have changed since.
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# (round 6: the trace pass profiles the HEADLINE alone -- with the secondary legs in it the dominant-kernel rows mixed the headline's launches with the
# legs' and k_xmarch of the checkerboard legs topped the table: VERDICT r05 weak 6; DSA_PROFILE_SECONDARY=1 adds a second trace of the whole command)
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- python3 bench.py --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1
if [ -n "${DSA_PROFILE_SECONDARY:-}" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_all -o r -- python3 bench.py --no-cpu-baseline > $OUT/trace_all.log 2>&1; fi
for g in "fetch FETCH_SIZE" "write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"; do
  set -- $g; n=$1; shift
  DSA_BENCH_SETTLE=0 timeout 400 rocprofv3 --pmc "$@" -d $OUT/pmc/$n -o r -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 > $OUT/$n.log 2>&1
done
{
  echo "== kernel trace (rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary)"; grep '"metric"' $OUT/trace.log | cut -c1-600
  python3 tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) $OUT/trace.log
  if [ -n "${DSA_PROFILE_SECONDARY:-}" ]; then echo "== kernel trace of the whole bench command (secondary legs included)"; python3 tools/rocpd_summary.py $(find $OUT/trace_all -name "*.db" | head -1) $OUT/trace_all.log | sed -n '1,/^$/p'; fi
  for n in fetch write busy insts wait; do echo "== pmc $n"; grep '"metric"' $OUT/$n.log | cut -c1-200; python3 tools/rocpd_pmc.py $(find $OUT/pmc/$n -name "*.db" | head -1) | grep "k_fim\|k_x"; done
  echo "== pmc_latest.json"
  python3 tools/pmc_to_json.py $OUT/pmc 16000 $OUT/pmc_latest.json k_fim_bundle ${DSA_PMC_STEPS:-1}
} > $OUT/summary.txt 2>&1
tail -30 $OUT/summary.txt
