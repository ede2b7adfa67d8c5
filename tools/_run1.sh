cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 1 2; do
  timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -d gpurun_out/sp_$n -o sp --output-format csv -- python tools/sweep_point.py $n wavefront > gpurun_out/sp_$n.log 2>&1
  f=$(find gpurun_out/sp_$n -name "*kernel_stats.csv" | head -1); echo "== n=$n"; grep wavefront gpurun_out/sp_$n.log | tail -1; cut -d, -f1-4 $f | head -8
done
python - <<'PY'
import csv, collections
for n in (1, 2):
    rows = list(csv.DictReader(open(f"gpurun_out/sp_{n}/sp_kernel_trace.csv")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last render: the last 19 kernels before the final develop
    names = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows]
    idx = [i for i, (k, _) in enumerate(names) if k == "k_wf_raygen"][-1]
    print(n, " ".join(f"{k[5:9]}:{ms:.2f}" for k, ms in names[idx:idx + 20] if k.startswith("k_wf")))
PY
