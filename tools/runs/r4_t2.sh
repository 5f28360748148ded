export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "mixed_5b_block or split" 2>&1 | grep -v "^  " | tail -30 > gpurun_out/t2_model.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "halo_tile or split_bf16" 2>&1 | tail -30 > gpurun_out/t2_ht3.log
timeout 600 python bench.py --dtype fp32s --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/t2_fp32s_b64.json 2> gpurun_out/t2_fp32s_b64.log
timeout 600 python bench.py --dtype fp32s --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --no-extras --no-side-stream --profile-all > gpurun_out/t2_fp32s_sites.json 2> gpurun_out/t2_fp32s_sites.txt
cd /tmp; R=$GRAFT_REPO_ROOT; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t2_trace_fp32s -- python bench.py --dtype fp32s --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --no-extras > gpurun_out/t2_trace_fp32s.json 2> gpurun_out/t2_trace_fp32s.log
tail -5 gpurun_out/t2_model.log; tail -3 gpurun_out/t2_ht3.log; cut -c1-200 gpurun_out/t2_fp32s_b64.json
