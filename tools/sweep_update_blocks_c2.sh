for UB in 0 80 120 240 320; do python tools/bench_configs.py c2 --update-blocks $UB --sweeps 500 2>&1 | grep view-updates | sed "s/^/ub=$UB /"; done
