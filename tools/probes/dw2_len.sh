for L in 0 8 16; do echo "L=$L"; MDS_KNOBS="9=$L" python tools/kbench.py dw_fwd dw_bwd 2>/dev/null | grep "s3 \|s4\|s5 "; done
