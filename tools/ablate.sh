for d in 0 1 2 4 8 6 7; do echo "SC_DBG=$d"; SC_DBG=$d python tools/time_search.py --n 10000 --angles 2 --reps 2 --prof 1 2>&1 | tail -1; done
