set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "halo" > $O/pytest_ht.log 2>&1; tail -15 $O/pytest_ht.log
timeout 900 python tools/conv_ab.py --ht --batch 64 --rounds 5 --iters 3 > $O/ht_ab.txt 2>&1; cat $O/ht_ab.txt
timeout 900 python tools/conv_ab.py --ht --batch 64 --rounds 5 --iters 3 --stats > $O/ht_ab_stats.txt 2>&1; cat $O/ht_ab_stats.txt
