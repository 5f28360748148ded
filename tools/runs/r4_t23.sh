export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "splitk" 2>&1 | tail -2
cd /tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_infer2 -- python bench.py --mode infer --batch 1 --graph --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/trace_infer2.json 2> gpurun_out/trace_infer2.log
python tools/trace_infer.py gpurun_out/trace_infer2/*/*_kernel_trace.csv > gpurun_out/trace_infer2.txt 2>&1
tail -3 gpurun_out/trace_infer2.txt
