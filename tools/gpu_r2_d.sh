set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv or exact or halo or pingpong" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python tools/conv_ab.py --ht --batch 64 --rounds 5 --iters 3 --stats > $O/ht_ab_stats.txt 2>&1; cat $O/ht_ab_stats.txt
for i in 1 2; do
timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 6 > $O/bench_$i.json 2>> $O/bench.log
done
timeout 600 python bench.py --no-side-stream --profile-all --no-sweep --no-cpu-baseline --steps 2 --warmup 2 > $O/bench_sites.json 2> $O/sites_b192.txt
grep -h '"value"' $O/*.json | cut -c1-120
