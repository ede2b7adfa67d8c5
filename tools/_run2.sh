cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rough.py tests/test_gpu_configs.py tests/test_gpu_nlos.py -x -q -m gpu 2>&1 | tail -5
bash tools/ab_wf.sh ab/libs/lib_cur.so ab/libs/lib_xcd.so
SPP=256 bash tools/ab_stair.sh ab/libs/lib_cur.so ab/libs/lib_xcd.so
for n in 2 6 26; do python tools/sweep_point.py $n wavefront 2>/dev/null | tail -1; done
