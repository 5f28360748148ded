export TMPDIR=/tmp; mkdir -p gpurun_out
for r in 1 2; do
for v in 200000 800000; do
for b in 16 32; do
  echo -n "train_vox=$v: "; VINET_BRANCH_STREAMS_TRAIN_VOX=$v python tools/sweep_small.py $b 10 2>/dev/null | tail -1
done; done; done
