cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py --model avinet --no-sweep --no-cpu-baseline --steps 4 2>>$O/log | tee $O/avinet.json | cut -c1-120
timeout 600 python bench.py --mode infer --batch 64 --no-cpu-baseline --steps 10 2>>$O/log | tee $O/infer_b64.json | cut -c1-120
timeout 600 python bench.py --mode infer --batch 1 --graph --no-cpu-baseline --steps 200 --warmup 5 2>>$O/log | tee $O/infer_b1_graph.json | cut -c1-120
timeout 600 python bench.py --clip 64 --height 256 --width 448 --no-sweep --no-cpu-baseline --steps 3 --warmup 1 2>>$O/log | tee $O/cfg5.json | cut -c1-120
timeout 600 python -m vinet_amd.generate_result --synthetic_frames 191 2>>$O/log | tail -1
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
