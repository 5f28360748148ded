#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_sq
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_sq/sq_counters.txt
cd $R
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq/pass1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/pmc_sq/pass1.json 2> gpurun_out/pmc_sq/pass1.log
tail -3 gpurun_out/pmc_sq/pass1.log
ls gpurun_out/pmc_sq/pass1/*/ | head
