python tools/kbench.py conv_wgrad 2>&1 | grep -E "b1.1|b2.1" | sed 's/^/base    /'
for n in 256 1536 1792; do C3_LIB=libmds_c3abl$n.so.bin python tools/kbench.py conv_wgrad 2>&1 | grep -E "b1.1|b2.1" | sed "s/^/abl$n  /"; done
