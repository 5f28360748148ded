export TMPDIR=/tmp; mkdir -p gpurun_out
for g in 1 4 8 16; do
for b in 8 32; do
  echo -n "group=$g: "; VINET_WGRAD_GROUP=$g VINET_WGRAD_GROUP_CAPTURE=$g python tools/sweep_small.py $b 10 2>/dev/null | tail -1
done; done
