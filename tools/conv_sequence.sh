# the gather-convolution launches of ONE steady-state SpUNet step in issue order (kernel, grid, duration): which launches are follow-ups,
# which are whole convolutions on conv3.  bash tools/conv_sequence.sh [model]
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; M=${1:-spunet}
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/cseq --output-format csv -- python $R/bench.py --model $M --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-fp16-recipe > /dev/null 2>&1
f=$(ls $R/gpurun_out/cseq/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
conv = [r for r in rows if re.search(r"conv[0-9]_kernel", r["Kernel_Name"])]
# the last step: the last third of the launches
conv = conv[len(conv) * 2 // 3:]
for r in conv:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{name[:62]:62s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} wgs  {us:8.1f} us")
PY
rm -rf $R/gpurun_out/cseq
