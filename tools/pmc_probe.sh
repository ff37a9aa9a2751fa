#!/bin/bash
# usage: tools/pmc_probe.sh <tag> <perf_probe args...>   (run on the GPU box)
# Collects SQ instruction counters for the megakernel and prints per-wave-step averages.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_$tag -o p -- python3 tools/perf_probe.py "$@" > gpurun_out/pmc_$tag.log 2>&1 || exit 1
python3 tools/pmc_report.py gpurun_out/pmc_$tag/p_results.db gpurun_out/pmc_$tag.log
