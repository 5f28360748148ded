export TMPDIR=/tmp; mkdir -p gpurun_out
python bench.py --batch 8 --steps 3 --warmup 2 --no-sweep --no-extras --no-cpu-baseline --no-side-stream --profile-all 2> gpurun_out/b8_sites.txt > /dev/null
head -60 gpurun_out/b8_sites.txt
cd /tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_b8 -- python bench.py --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/trace_b8.json 2> gpurun_out/trace_b8.log
python tools/trace_timeline.py gpurun_out/trace_b8/*/*_kernel_trace.csv 2>&1 | tail -40
