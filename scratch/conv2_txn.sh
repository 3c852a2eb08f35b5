for t in 0 2 3 4 6 9; do echo "MPA_WG_TXN=$t"; MPA_WG_TXN=$t timeout -k 10 100 python scratch/conv2_time.py 2>&1 | grep "mode 2"; done
