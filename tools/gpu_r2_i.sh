cd $GRAFT_REPO_ROOT
for i in 1 2; do
for cfg in "0 144" "-1 144" "-1 192" "-1 256" "0 192"; do
set -- $cfg
echo "prio=$1 cus=$2"; VINET_WGRAD_CUS=$2 timeout 300 python bench.py --no-sweep --no-cpu-baseline --steps 5 --main-priority $1 2>/dev/null | cut -c1-100
done
done
