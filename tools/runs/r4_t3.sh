export TMPDIR=/tmp; mkdir -p gpurun_out
(python tools/dbg_split_grad.py fp32s;  python tools/dbg_split_grad.py fp32) 2>&1 | grep -v "amdgpu.ids\|share" > gpurun_out/t3_split_grad.txt
cat gpurun_out/t3_split_grad.txt
