for f in "" "1,12" "1,8" "1,6" "1,4" "2,6" "2,4"; do echo "FORCE=$f"; MPA_FWD_FORCE="$f" timeout -k 10 100 python scratch/b32_force.py 32 2>&1 | grep mode; done
