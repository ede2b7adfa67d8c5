cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rough.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -5
bash tools/ab_wf.sh ab/libs/lib_base.so ab/libs/lib_diet.so ab/libs/lib_diet2.so
SPP=256 bash tools/ab_stair.sh ab/libs/lib_base.so ab/libs/lib_diet.so ab/libs/lib_diet2.so
for n in 2 6; do python tools/sweep_point.py $n wavefront 2>/dev/null | tail -1; done
