cd $GRAFT_REPO_ROOT
SPP=256 bash tools/ab_stair.sh ab/libs/lib_cur.so ab/libs/lib_ts8.so
