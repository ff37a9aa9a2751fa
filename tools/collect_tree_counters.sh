#!/bin/bash
# Run on the GPU box (via gpurun): SQ counter passes of one 256-sample launch of the headline frame with scene 21
# forced through the sphere tree (bench.py --tree-min-spheres 1).  usage: tools/collect_tree_counters.sh <tag>
set -u
tag=${1:-tree}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profile_$tag
mkdir -p $out
pass() { # name, counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 64 --tree-min-spheres 1 > $out/$name.json 2> $out/$name.err || exit 1
}
pass pmc_valu SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
pass pmc_lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
