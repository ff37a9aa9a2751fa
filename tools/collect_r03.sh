#!/bin/bash
# Run on the GPU box (via gpurun): the rocprofv3 evidence behind bench.py's roofline (profiles/<tag>_*).
#   1. kernel-trace stats of the DEFAULT bench.py run (the command the driver runs)
#   2. counter passes, one run each (no tracing), of ONE launch of the headline frame: bench.py --steps 1 --warmup 0
# usage: tools/collect_r03.sh <tag> [extra bench.py arguments for the counter passes...]
set -u
tag=${1:-r03}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profile_$tag
mkdir -p $out
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err || exit 1
pass() { # name, counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 "${EXTRA[@]}" > $out/$name.json 2> $out/$name.err || exit 1
}
EXTRA=("$@")
pass pmc_valu SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
pass pmc_lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
ls $out
