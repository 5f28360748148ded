export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/host_profile.py 8 10 > gpurun_out/host_profile_b8.txt 2>&1
head -80 gpurun_out/host_profile_b8.txt
