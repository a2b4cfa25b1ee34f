for b in 0 192 128 96 64; do MDS_KNOBS="0=$b" python tools/kbench.py conv_wgrad 2>&1 | grep -E "b1.1|b2.1" | sed "s/^/blocks=$b  /"; done
