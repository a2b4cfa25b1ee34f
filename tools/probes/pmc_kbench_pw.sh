set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
pass() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" -d /tmp/pk_$name -o pmc --output-format csv -- python $R/tools/kbench.py pw_fwd > /dev/null 2>/tmp/pk_$name.err || true; f=$(find /tmp/pk_$name -name "*counter_collection.csv" | head -1); python $R/tools/pmc_kernel.py $f | grep -A12 "wres" ; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT
pass g2 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass f FETCH_SIZE
pass w WRITE_SIZE
