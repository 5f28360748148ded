cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 6 2>> $O/bench.log | cut -c1-110
done
timeout 600 python bench.py --no-side-stream --profile-all --no-sweep --no-cpu-baseline --steps 2 --warmup 2 > $O/bench_sites.json 2> $O/sites_b192.txt
grep "sum of bracketed" $O/sites_b192.txt
