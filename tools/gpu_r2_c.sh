set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for i in 1 2; do
VINET_OPT=ht=0 timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 6 > $O/bench_ht0_$i.json 2>> $O/bench.log
timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 6 > $O/bench_ht1_$i.json 2>> $O/bench.log
done
timeout 600 python bench.py --no-side-stream --profile-all --no-sweep --no-cpu-baseline --steps 2 --warmup 2 > $O/bench_sites.json 2> $O/sites_b192.txt
grep -h '"value"' $O/*.json | cut -c1-120
