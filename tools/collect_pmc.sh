#!/bin/bash
# rocprofv3 --pmc passes of one command (counters never share a run with trace domains other than kernel-trace).
#   bash tools/collect_pmc.sh <tag> <groups> [lib] [-- program args...]
# <groups>: comma-separated names from the table below (each is one pass = one run of the command), e.g. "insts,busy,wait"
# [lib]: A/B build to load (DSA_LIB_PATH), or - for the default library; default command: python3 tools/perf_probe.py 131 1024 1.25 smooth 256
# Output: gpurun_out/<tag>/summary.txt (per-kernel counter sums of every pass) -- copy what should be judged into profiles/.
set -u
TAG=${1:-pmc}; GROUPS_=${2:-insts,busy,wait}; shift 2 || true
LIB=""
if [ $# -gt 0 ] && [ "$1" != "--" ]; then LIB=$1; shift; fi
[ $# -gt 0 ] && [ "$1" == "--" ] && shift
if [ $# -gt 0 ]; then CMD=("$@"); else CMD=(python3 tools/perf_probe.py 131 1024 1.25 smooth 256); fi
[ -n "$LIB" ] && [ "$LIB" != "-" ] && export DSA_LIB_PATH=$LIB
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
declare -A SETS=(
  [insts]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
  [busy]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
  [wait]="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_WAVES"
  [lds]="SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
  [vmem]="SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
  [tcc]="TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum"
  [tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"
  [ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum"
  [eastall]="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"
  [eawr]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"
  [tcplat]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
  [tcpstall]="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
  [icache]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"
  [fp64]="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
  [flops64]="SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
  [fp32]="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
)
: > $OUT/rc.txt
IFS=',' read -ra GS <<< "$GROUPS_"
for g in "${GS[@]}"; do
  c=${SETS[$g]:-}
  if [ -z "$c" ]; then echo "unknown group $g" >> $OUT/rc.txt; continue; fi
  timeout 300 rocprofv3 --pmc $c -d $OUT/$g -o r -- "${CMD[@]}" > $OUT/$g.log 2>&1
  echo "$g rc=$?" >> $OUT/rc.txt
done
{
  echo "# rocprofv3 --pmc, one pass per group; command: ${CMD[*]}  lib: ${LIB:-default}"
  for g in "${GS[@]}"; do
    db=$(find $OUT/$g -name "*.db" 2>/dev/null | head -1)
    echo "== $g"; grep -E "solves/s|\"metric\"" $OUT/$g.log | cut -c1-240
    [ -n "$db" ] && python3 tools/rocpd_pmc.py $db | grep -v "^#" | head -40
  done
} > $OUT/summary.txt 2>&1
cat $OUT/rc.txt $OUT/summary.txt
