set -x
TAG=${1:-r1_v4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$TAG
cd $R
# 1) kernel-trace stats of the default bench command (steps 4 warmup 2)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG/trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/prof_$TAG/bench_under_rocprof.json 2> gpurun_out/prof_$TAG/trace.log
# 2) PMC passes (own runs, 1 step)
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof_$TAG/pmc_$tag -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/prof_$TAG/pmc_$tag.json 2> gpurun_out/prof_$TAG/pmc_$tag.log
done
# 3) the bench line itself with the CPU baseline
timeout 900 python bench.py > gpurun_out/prof_$TAG/bench.json 2> gpurun_out/prof_$TAG/bench.log
ls -la gpurun_out/prof_$TAG gpurun_out/prof_$TAG/*/* | head -40
du -sh gpurun_out/prof_$TAG
