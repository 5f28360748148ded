cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
for i in 1 2; do
for opt in "ht_t=0,ht_pre=0" "ht_t=1,ht_pre=0" "ht_t=0,ht_pre=1" "ht_t=1,ht_pre=1"; do
echo $opt; VINET_OPT=$opt timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 5 2>> $O/bench.log | cut -c1-100
done
done
