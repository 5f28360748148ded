export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_${1:-x} -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/trace_${1:-x}.json 2> gpurun_out/trace_${1:-x}.log
python tools/trace_timeline.py gpurun_out/trace_${1:-x}/*/*_kernel_trace.csv
