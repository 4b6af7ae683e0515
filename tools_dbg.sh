for d in 0 1 2 3 4 7; do echo "DBG=$d"; CTGAN_DBG=$d python tools_conv_bench.py 10 2>/dev/null | sed -n '2p;6p'; done
