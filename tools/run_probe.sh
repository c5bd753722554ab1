#!/bin/bash
# One runner for GPU-box jobs (replaces the per-experiment r3_*.sh scripts of round 3).
#   bash tools/run_probe.sh <tag> <job> [job ...]
# Every job's output goes to gpurun_out/<tag>/<job>.log; a one-line tail of each is printed.  Jobs:
#   tests            the GPU test suite (pytest -m gpu)
#   tests:<expr>     ... restricted with -k <expr>
#   smoke            __graft_entry__.smoke()
#   bench            python3 bench.py --steps 5 --warmup 2
#   profile          tools/profile_bench.sh <tag> (rocprofv3 kernel trace + counter passes of the bench command)
#   exact:<args>     tools/exact_probe.py <args with , for spaces>          e.g. exact:131,4096,checker,0,0
#   bundle:<args>    tools/bundle_probe.py <args with , for spaces>
#   py:<path>[:args] any probe script of tools/ or tests/tools/ with , separated arguments
# Each job runs under its own timeout (DSA_JOB_TIMEOUT seconds, default 1500).
TAG=${1:?tag}; shift
O=gpurun_out/$TAG; mkdir -p $O
T=${DSA_JOB_TIMEOUT:-1500}
for job in "$@"; do
  name=${job%%:*}; arg=""; [ "$job" != "$name" ] && arg=${job#*:}
  log=$O/$(echo "$job" | tr ':/, ' '____').log
  case $name in
    tests)   if [ -n "$arg" ]; then timeout $T python3 -m pytest tests -m gpu -x -q -k "$arg" > $log 2>&1; else timeout $T python3 -m pytest tests -m gpu -x -q > $log 2>&1; fi ;;
    smoke)   timeout $T python3 -c "import __graft_entry__ as g; g.smoke()" > $log 2>&1 ;;
    bench)   timeout $T python3 bench.py --steps 5 --warmup 2 > $log 2>&1 ;;
    profile) bash tools/profile_bench.sh $TAG > $log 2>&1 ;;
    exact)   timeout $T python3 tools/exact_probe.py ${arg//,/ } > $log 2>&1 ;;
    bundle)  timeout $T python3 tools/bundle_probe.py ${arg//,/ } > $log 2>&1 ;;
    py)      path=${arg%%:*}; a=""; [ "$arg" != "$path" ] && a=${arg#*:}; timeout $T python3 $path ${a//,/ } > $log 2>&1 ;;
    *)       echo "unknown job $job" > $log ;;
  esac
  echo "== $job rc=$?"; tail -${DSA_JOB_TAIL:-4} $log | cut -c1-300
done
