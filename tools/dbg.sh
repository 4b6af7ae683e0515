for d in 0 16; do echo "DBG=$d"; CTGAN_DBG=$d python tools/conv_bench.py 20 2>/dev/null | sed -n '2,4p'; done
