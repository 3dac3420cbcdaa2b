export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
PTC_CONV8=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c8prof --output-format csv -- python $R/bench.py --model spunet --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-fp16-recipe > /dev/null 2>&1
f=$(ls $R/gpurun_out/c8prof/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:40]: print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
rm -rf $R/gpurun_out/c8prof
