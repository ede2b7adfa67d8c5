cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 2 6; do
  timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -d gpurun_out/sp_$n -o sp --output-format csv -- python tools/sweep_point.py $n wavefront > gpurun_out/sp_$n.log 2>&1
  f=$(find gpurun_out/sp_$n -name "*kernel_stats.csv" | head -1); echo "== n=$n"; grep wavefront gpurun_out/sp_$n.log | tail -1; cut -d, -f1-4 $f | head -8
done
