cd $GRAFT_REPO_ROOT
bash tools/ab_wf.sh ab/libs/lib_cur.so ab/libs/lib_tw6.so ab/libs/lib_tw8.so
for lib in cur tw6 tw8; do echo -n "$lib "; MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_$lib.so python tools/sweep_point.py 2 wavefront 2>/dev/null | tail -1; done
