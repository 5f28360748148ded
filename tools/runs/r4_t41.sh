export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_b8b -- python bench.py --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/trace_b8b.json 2> gpurun_out/trace_b8b.log
python tools/trace_timeline.py gpurun_out/trace_b8b/*/*_kernel_trace.csv 2>&1 | tail -12
