for c in 1024 832 480 192 64 32; do echo "== defer only Cin=$c"; VINET_DBG_DEFER_IN_CAPTURE=$c SHOW=0 python tools/dbg_graph.py 2>&1 | grep -E "parameters differ"; done
