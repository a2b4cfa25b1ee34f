for L in 0 4 5 7 8 10 20; do echo "L=$L"; MDS_KNOBS="7=$L" python tools/kbench.py dw_fwd dw_bwd 2>/dev/null | grep "3d"; done
