cd $GRAFT_REPO_ROOT
bash tools/ab_wf.sh ab/libs/lib_cur.so ab/libs/lib_sw4.so
for lib in cur sw4; do echo -n "$lib "; MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_$lib.so python tools/sweep_point.py 2 wavefront 2>/dev/null | tail -1; done
SPP=256 bash tools/ab_stair.sh ab/libs/lib_cur.so ab/libs/lib_sw5h.so
