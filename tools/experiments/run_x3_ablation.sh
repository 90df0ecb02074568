for a in 1 2 4 6 7; do echo "X3_ABL=$a"; HSP_LIB=$GRAFT_REPO_ROOT/build_tmp/libhsp_x3abl$a.so python $GRAFT_REPO_ROOT/tools/time_x3_tall.py 2>&1 | grep M16448 | head -2; done
