python tools/kbench.py conv_wgrad 2>&1 | grep -E "b0.0" | sed 's/^/base    /'
for n in 1056 1312; do C3_LIB=libmds_c3abl$n.so.bin python tools/kbench.py conv_wgrad 2>&1 | grep -E "b0.0" | sed "s/^/abl$n  /"; done
