# transform-wave form of the first 3x3 layer: which stage binds?  (experiment builds: make c3abl ABL=n)
export C3_ONLY="pro"
python tools/c3_bench.py 0 2>&1 | grep -v amdgpu.ids | sed 's/^/base   /'
for n in 16 32 2 8 18 24 40; do C3_LIB=libmds_c3abl$n.so.bin python tools/c3_bench.py 0 2>&1 | grep -v amdgpu.ids | sed "s/^/abl$n  /"; done
