export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_infer3 -- python bench.py --mode infer --batch 1 --graph --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/trace_infer3.json 2> gpurun_out/trace_infer3.log
python tools/trace_infer.py gpurun_out/trace_infer3/*/*_kernel_trace.csv > gpurun_out/trace_infer3.txt 2>&1
tail -3 gpurun_out/trace_infer3.txt
