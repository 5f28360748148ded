export TMPDIR=/tmp; mkdir -p gpurun_out
for g in 32 64 128; do
for b in 8 32; do
  echo -n "capture group=$g: "; VINET_WGRAD_GROUP_CAPTURE=$g python tools/sweep_small.py $b 10 2>/dev/null | tail -1
done; done
for b in 8 32; do
  echo -n "graph on one stream: "; SWEEP_GRAPH_ONE_STREAM=1 python tools/sweep_small.py $b 10 2>/dev/null | tail -1
done
