set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log
for i in 1 2; do
VINET_DEFER_DECODER_WGRAD=1 timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 6 > $O/bench_defer_$i.json 2>> $O/bench_defer.log
VINET_DEFER_DECODER_WGRAD=0 timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 6 > $O/bench_nodefer_$i.json 2>> $O/bench_defer.log
done
timeout 600 python bench.py --clip 64 --height 256 --width 448 --no-sweep --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.log
timeout 600 python bench.py --no-side-stream --profile-all --no-sweep --no-cpu-baseline --steps 2 --warmup 2 > $O/bench_sites.json 2> $O/sites_b192.txt
cp gpurun_out/parity_report.jsonl $O/ 2>/dev/null
grep -h '"value"' $O/*.json | cut -c1-200
