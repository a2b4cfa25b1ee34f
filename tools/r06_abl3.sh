export C3_ONLY="s2"
python tools/c3_bench.py 0 2>&1 | grep -v amdgpu.ids | sed 's/^/base   /'
for n in 16 32; do C3_LIB=libmds_c3abl$n.so.bin python tools/c3_bench.py 0 2>&1 | grep -v amdgpu.ids | sed "s/^/abl$n  /"; done
