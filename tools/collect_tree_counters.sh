#!/bin/bash
# Run on the GPU box (via gpurun): SQ counter passes of one launch of bench.py with the given arguments (default: the
# headline frame at 256 samples with scene 21 forced through the sphere tree).
# usage: tools/collect_tree_counters.sh <tag> [bench.py arguments...]
set -u
tag=${1:-tree}
shift || true
if [ $# -eq 0 ]; then set -- --spp 64 --tree-min-spheres 1; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profile_$tag
mkdir -p $out
BENCH_ARGS=("$@")
pass() { # name, counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 "${BENCH_ARGS[@]}" > $out/$name.json 2> $out/$name.err || exit 1
}
pass pmc_valu SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
pass pmc_lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
