cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "halo" > $O/pytest_ht.log 2>&1; tail -12 $O/pytest_ht.log
timeout 900 python tools/conv_ab.py --ht --batch 64 --rounds 5 --iters 3 --stats --only "x1x1" > $O/ht_t_ab.txt 2>&1; cat $O/ht_t_ab.txt
timeout 900 python tools/conv_ab.py --ht --batch 64 --rounds 5 --iters 3 --stats --only "3c s" >> $O/ht_t_ab.txt 2>&1; tail -2 $O/ht_t_ab.txt
