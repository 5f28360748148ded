cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
for c in 64 96 128 160 256; do
echo "wgrad_cus=$c"; VINET_WGRAD_CUS=$c timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 5 2>>$O/log | cut -c1-100
done
echo "no side stream"; timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 5 --no-side-stream 2>>$O/log | cut -c1-100
echo "defer=0"; VINET_DEFER_DECODER_WGRAD=0 timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 5 2>>$O/log | cut -c1-100
echo "b128"; timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 5 --batch 128 2>>$O/log | cut -c1-100
