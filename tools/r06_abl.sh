python tools/c3_bench.py 0 2>&1 | grep -v amdgpu.ids | head -2 | sed 's/^/base  /'
for n in 1 2 3 4 7 8; do C3_LIB=libmds_c3abl$n.so.bin python tools/c3_bench.py 0 2>&1 | grep -v amdgpu.ids | head -2 | sed "s/^/abl$n  /"; done
