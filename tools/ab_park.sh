for lib in mitransient_amd/csrc/libmitransient_amd.so ab/libpark.so; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'ms/step %.2f' % r['ms_per_step'], 'kernel %.2f' % r['roofline']['avg_launch_ms'], 'Mray/s %.0f' % r['value'])
"
done
tools/ws_env.sh "MITRANSIENT_AMD_LIB=$(pwd)/ab/libpark.so" park
tools/ws_env.sh "A=1" base
