#!/bin/bash
# One GPU-box session: new-kernel tests first, the whole -m gpu suite, op microbench, bench.py in
# A/B configurations, rocprofv3 kernel stats + PMC passes for the roofline kernel.
# Usage (through gpurun): bash tools/gpu_session.sh <tag>
TAG=${1:-s}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $O/${TAG}_env.log 2>&1
nproc >> $O/${TAG}_env.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "linear or layer_norm" > $O/${TAG}_new.log 2>&1
echo "new-tests rc=$?" >> $O/${TAG}_env.log
timeout 1500 python -m pytest tests -q -m gpu > $O/${TAG}_all.log 2>&1
echo "all-tests rc=$?" >> $O/${TAG}_env.log
timeout 900 python tools/bench_ops.py > $O/${TAG}_ops.log 2>&1
echo "bench_ops rc=$?" >> $O/${TAG}_env.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.log 2>&1
echo "bench rc=$?" >> $O/${TAG}_env.log
PTC_FUSE_GATHER=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_nofuse.log 2>&1
PTC_OWN_LINEAR=0 PTC_OWN_NORM=0 PTC_FUSE_GATHER=0 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_lib.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${TAG}_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/${TAG}_roof_stats -- python $GRAFT_REPO_ROOT/tools/roofline_kernel.py > $GRAFT_REPO_ROOT/$O/${TAG}_roof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/${TAG}_roof_fetch -- python $GRAFT_REPO_ROOT/tools/roofline_kernel.py > $GRAFT_REPO_ROOT/$O/${TAG}_roof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$O/${TAG}_roof_write -- python $GRAFT_REPO_ROOT/tools/roofline_kernel.py > $GRAFT_REPO_ROOT/$O/${TAG}_roof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/${TAG}_roof_sq -- python $GRAFT_REPO_ROOT/tools/roofline_kernel.py > $GRAFT_REPO_ROOT/$O/${TAG}_roof_sq.log 2>&1
cd $GRAFT_REPO_ROOT
# keep only the small summaries (the traces are large)
find $O/${TAG}_prof $O/${TAG}_roof_stats -name "*kernel_trace.csv" -delete 2>/dev/null
python tools/pmc_summary.py --stats $O/${TAG}_roof_stats --pmc $O/${TAG}_roof_fetch $O/${TAG}_roof_write $O/${TAG}_roof_sq \
  --kernels attn_fwd_kernel,attn_bwd_dq_kernel,attn_bwd_dkv_kernel --out $O/${TAG}_roof_pmc.json > $O/${TAG}_roof_pmc.log 2>&1
find $O/${TAG}_roof_fetch $O/${TAG}_roof_write $O/${TAG}_roof_sq -name "*.csv" -size +2M -delete 2>/dev/null
tail -3 $O/${TAG}_new.log $O/${TAG}_all.log; tail -2 $O/${TAG}_bench.log $O/${TAG}_bench_nofuse.log $O/${TAG}_bench_lib.log; cat $O/${TAG}_env.log
