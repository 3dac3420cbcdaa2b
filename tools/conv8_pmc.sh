export PTC_CONV8=1 C8_ABL=16 TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
C8_ABL=16 timeout 300 python $R/tools/conv8_time.py 128 96 2>&1 | tail -1
for cs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  n=$(echo $cs | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $cs -d $O/c8pmc_$n --output-format csv -- python $R/tools/conv8_time.py 128 96 > /dev/null 2>&1
done
python - <<'PY'
import glob, csv, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out"
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
for f in glob.glob(O+"/c8pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv8_kernel" not in k or "Lb0" in k: continue
        tot[k[:60]][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES","SQ_INSTS_VALU","SQ_WAIT_INST_LDS"): cnt[(k[:60],r["Counter_Name"])]+=1
for k,d in tot.items():
    print(k)
    for c,v in sorted(d.items()):
        n=max(cnt.get((k,"SQ_WAVE_CYCLES"),1),1)
        print(f"   {c:28s} {v:16.0f}")
    print("   launches", {c:n for (kk,c),n in cnt.items() if kk==k})
PY
rm -rf $O/c8pmc_*
