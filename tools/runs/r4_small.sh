export TMPDIR=/tmp; mkdir -p gpurun_out
python bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-sweep --no-extras 2>/dev/null | cut -c1-160
python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --no-extras --no-side-stream --profile-all > /dev/null 2> gpurun_out/small_b8_sites.txt
python bench.py --mode infer --batch 1 --graph --steps 50 --warmup 5 --no-cpu-baseline --no-sweep --no-extras 2>/dev/null | cut -c1-200
python bench.py --mode infer --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-extras --profile-all > /dev/null 2> gpurun_out/small_infer1_sites.txt
cd /tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small_trace_b8 -- python bench.py --batch 8 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-extras > /dev/null 2> gpurun_out/small_trace_b8.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small_trace_i1 -- python bench.py --mode infer --batch 1 --graph --steps 20 --warmup 3 --no-cpu-baseline --no-sweep --no-extras > /dev/null 2> gpurun_out/small_trace_i1.log
