#!/bin/bash
# One GPU-box session.  Usage (through gpurun): bash tools/gpu_session.sh <tag> <sections...>
#   tests      whole -m gpu suite            ops     tools/bench_ops.py
#   bench      bench.py (default switches)   ab:<ENV=V,...>  bench.py with switches (no cpu baseline)
#   prof       rocprofv3 kernel stats of ONE steady-state bench.py step (two runs, tools/steady_state_stats.py); profsmall: the 2 x 20000 step
#   contention tools/host_contention.py: host CPU time per step of 1 vs 8 concurrent processes
#   hostlead   tools/host_lead.py: host enqueue time per step vs completed time (executor on / off, and a 2 x 20000-voxel step)
#   roof       rocprofv3 stats + PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) of the roofline kernel
TAG=${1:-s}; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $O/${TAG}_env.log 2>&1
echo "cores $(nproc)" >> $O/${TAG}_env.log
for sec in "$@"; do
  cd $R
  case $sec in
    tests) timeout 1500 python -m pytest tests -q -m gpu > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $O/${TAG}_env.log; tail -4 $O/${TAG}_tests.log;;
    ops) timeout 900 python tools/bench_ops.py > $O/${TAG}_ops.log 2>&1; echo "ops rc=$?" >> $O/${TAG}_env.log;;
    ops:*) timeout 900 python tools/bench_ops.py --only ${sec#ops:} > $O/${TAG}_ops.log 2>&1; echo "ops rc=$?" >> $O/${TAG}_env.log; cat $O/${TAG}_ops.log | cut -c1-420;;
    test:*) timeout 1500 python -m pytest tests -q -m gpu -k "${sec#test:}" > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $O/${TAG}_env.log; tail -15 $O/${TAG}_tests.log;;
    probe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_gfx950 tools/probe_gfx950.hip > $O/${TAG}_probe.log 2>&1 && timeout 300 /tmp/probe_gfx950 >> $O/${TAG}_probe.log 2>&1; tail -20 $O/${TAG}_probe.log | cut -c1-200;;
    pgather) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_gather tools/probe_gather.hip > $O/${TAG}_probe_gather.log 2>&1 && timeout 300 /tmp/probe_gather >> $O/${TAG}_probe_gather.log 2>&1; cat $O/${TAG}_probe_gather.log | cut -c1-250;;
    opsq) timeout 900 python tools/bench_ops.py --quick > $O/${TAG}_ops.log 2>&1; echo "ops rc=$?" >> $O/${TAG}_env.log;;
    w7v:*) PTC_LIB_VARIANT=${sec#w7v:} timeout 600 python tools/wgrad7_time.py > $O/${TAG}_wgrad7_time_${sec#w7v:}.txt 2>&1; cat $O/${TAG}_wgrad7_time_${sec#w7v:}.txt;;
    c7t) timeout 600 python tools/conv7_time.py --all > $O/${TAG}_conv7_time.txt 2>&1; cat $O/${TAG}_conv7_time.txt;;
    attnv:*) PTC_LIB_VARIANT=${sec#attnv:} timeout 300 python tools/bench_ops.py --only attn 2>&1 | grep "attention n_seq" | cut -c1-200 > $O/${TAG}_attn_${sec#attnv:}.txt; echo "variant ${sec#attnv:}"; cat $O/${TAG}_attn_${sec#attnv:}.txt;;
    attnpad) for pad in 0 20000; do echo "PTC_AT_BWD_PAD_LDS=$pad"; PTC_AT_BWD_PAD_LDS=$pad timeout 300 python tools/bench_ops.py --only attn 2>&1 | grep "attention n_seq= 800"; done > $O/${TAG}_attn_pad.txt 2>&1; cat $O/${TAG}_attn_pad.txt;;
    attnprof) cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ap -- python $R/tools/bench_ops.py --only attn > $O/${TAG}_ap.log 2>&1
          cd $R; TOP=12 python tools/prof_top.py $O/${TAG}_ap 1 $O/${TAG}_attn_kernel_stats.csv > $O/${TAG}_ap_top.log 2>&1; rm -rf $O/${TAG}_ap; head -8 $O/${TAG}_attn_kernel_stats.csv | cut -c1-150;;
    w7) timeout 600 python tools/wgrad7_time.py > $O/${TAG}_wgrad7_time.txt 2>&1; cat $O/${TAG}_wgrad7_time.txt;;
    w7prof) cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_w7prof -- python $R/tools/wgrad7_time.py > $O/${TAG}_w7prof.log 2>&1
          cd $R; TOP=12 python tools/prof_top.py $O/${TAG}_w7prof 1 $O/${TAG}_wgrad7_kernel_stats.csv > $O/${TAG}_w7prof_top.log 2>&1; rm -rf $O/${TAG}_w7prof; head -14 $O/${TAG}_wgrad7_kernel_stats.csv | cut -c1-150;;
    bench) timeout 900 python bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.log 2>&1; echo "bench rc=$?" >> $O/${TAG}_env.log; tail -1 $O/${TAG}_bench.log | cut -c1-330;;
    hostlead) timeout 600 python tools/host_lead.py > $O/${TAG}_host_lead.txt 2>&1; PTC_EXEC_BLOCK=0 timeout 600 python tools/host_lead.py >> $O/${TAG}_host_lead.txt 2>&1; timeout 600 python tools/host_lead.py --scenes 2 --points 20000 >> $O/${TAG}_host_lead.txt 2>&1; cat $O/${TAG}_host_lead.txt;;
    contention) timeout 900 python tools/host_contention.py > $O/${TAG}_host_contention.txt 2>&1; cat $O/${TAG}_host_contention.txt;;
    smoke) timeout 600 python __graft_entry__.py --smoke > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${TAG}_env.log; tail -2 $O/${TAG}_smoke.log;;
    trace) timeout 600 python tools/trace_step.py > $O/${TAG}_trace_step.txt 2>$O/${TAG}_trace_step.err; echo "trace rc=$?" >> $O/${TAG}_env.log; head -5 $O/${TAG}_trace_step.txt;;
    copies) timeout 600 python tools/trace_copies.py > $O/${TAG}_trace_copies.txt 2>$O/${TAG}_trace_copies.err; echo "copies rc=$?" >> $O/${TAG}_env.log; head -30 $O/${TAG}_trace_copies.txt;;
    spunet) timeout 900 python bench.py --model spunet --steps 6 --warmup 2 > $O/${TAG}_spunet.log 2>&1; echo "spunet rc=$?" >> $O/${TAG}_env.log; tail -1 $O/${TAG}_spunet.log | cut -c1-400;;
    profm2) cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_profm2 -- python $R/bench.py --model ptv3m2-sonata --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-fp16-recipe > $O/${TAG}_profm2.log 2>&1
          cd $R; TOP=40 python tools/prof_top.py $O/${TAG}_profm2 5 $O/${TAG}_m2_kernel_stats.csv > $O/${TAG}_profm2_top.log 2>&1; rm -rf $O/${TAG}_profm2; head -40 $O/${TAG}_m2_kernel_stats.csv | cut -c1-150; tail -1 $O/${TAG}_m2_kernel_stats.csv;;
    profspunet) cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_profsp -- python $R/bench.py --model spunet --steps 3 --warmup 2 > $O/${TAG}_profsp.log 2>&1
          cd $R; TOP=30 python tools/prof_top.py $O/${TAG}_profsp 5 $O/${TAG}_spunet_kernel_stats.csv > $O/${TAG}_profsp_top.log 2>&1; rm -rf $O/${TAG}_profsp;;
    ab:*) envs=$(echo "${sec#ab:}" | tr ',' ' '); name=$(echo "${sec#ab:}" | tr -c 'A-Za-z0-9=\n' '_');
          env $envs timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_${name}.log 2>&1; echo "$sec: $(tail -1 $O/${TAG}_bench_${name}.log | cut -c1-200)";;
    abo:*) envs=$(echo "${sec#abo:}" | tr ',' ' '); name=$(echo "${sec#abo:}" | tr -c 'A-Za-z0-9=\n' '_');
          env $envs timeout 600 python bench.py --model ptv3-outdoor --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-fp16-recipe > $O/${TAG}_outdoor_${name}.log 2>&1; echo "$sec: $(tail -1 $O/${TAG}_outdoor_${name}.log | cut -c1-260)";;
    abs:*) envs=$(echo "${sec#abs:}" | tr ',' ' '); name=$(echo "${sec#abs:}" | tr -c 'A-Za-z0-9=\n' '_');
          env $envs timeout 600 python bench.py --model spunet --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_spunet_${name}.log 2>&1; echo "$sec: $(tail -1 $O/${TAG}_spunet_${name}.log | cut -c1-260)";;
    prof) cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_prof.log 2>&1
          timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof7 -- python $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_prof7.log 2>&1
          cd $R; TOP=30 python tools/steady_state_stats.py $O/${TAG}_prof 3 $O/${TAG}_prof7 7 $O/${TAG}_kernel_stats.csv > $O/${TAG}_prof_top.log 2>&1
          python tools/kernel_neighbours.py $O/${TAG}_prof > $O/${TAG}_fill_copy_neighbours.txt 2>&1; rm -rf $O/${TAG}_prof $O/${TAG}_prof7; tail -1 $O/${TAG}_kernel_stats.csv;;
    timeline) cd /tmp; timeout 900 rocprofv3 --kernel-trace -d $O/${TAG}_tl -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-fp16-recipe > $O/${TAG}_tl.log 2>&1
          cd $R; PTC_TIMELINE_SEQ=$O/${TAG}_step_sequence.txt python tools/step_timeline.py $O/${TAG}_tl > $O/${TAG}_step_timeline.txt 2>&1; rm -rf $O/${TAG}_tl; head -12 $O/${TAG}_step_timeline.txt;;
    timelinev:*) v=${sec#timelinev:}; cd /tmp; PTC_LIB_VARIANT=$v timeout 900 rocprofv3 --kernel-trace -d $O/${TAG}_tl_$v -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-fp16-recipe > $O/${TAG}_tl_$v.log 2>&1
          cd $R; PTC_TIMELINE_SEQ=$O/${TAG}_step_sequence_$v.txt python tools/step_timeline.py $O/${TAG}_tl_$v > $O/${TAG}_step_timeline_$v.txt 2>&1; rm -rf $O/${TAG}_tl_$v; head -2 $O/${TAG}_step_timeline_$v.txt
          awk '/gemm3/{n++; t+=$4} /wgrad3/{m++; u+=$4} /wgrad_reduce_multi/{k++; w+=$4} END{print "gemm3 launches", n, "us per step", t, "| wgrad3", m, u, "| wgrad_reduce_multi", k, w}' $O/${TAG}_step_sequence_$v.txt;;
    profoutdoor) cd /tmp; BA="--model ptv3-outdoor --no-cpu-baseline --no-secondary --no-fp16-recipe --warmup 1"
          timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_profo2 -- python $R/bench.py --steps 2 $BA > $O/${TAG}_profo2.log 2>&1
          timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_profo5 -- python $R/bench.py --steps 5 $BA > $O/${TAG}_profo5.log 2>&1
          cd $R; TOP=40 python tools/steady_state_stats.py $O/${TAG}_profo2 2 $O/${TAG}_profo5 5 $O/${TAG}_outdoor_kernel_stats.csv > $O/${TAG}_profo_top.log 2>&1
          rm -rf $O/${TAG}_profo2 $O/${TAG}_profo5; head -25 $O/${TAG}_outdoor_kernel_stats.csv | cut -c1-150; tail -1 $O/${TAG}_outdoor_kernel_stats.csv;;
    traffic) cd /tmp; BA="--no-cpu-baseline --no-secondary --no-fp16-recipe --warmup 1"
          for c in FETCH_SIZE WRITE_SIZE; do for k in 2 5; do
            timeout 600 rocprofv3 --pmc $c -d $O/${TAG}_tr_${c}_$k -- python $R/bench.py --steps $k $BA > $O/${TAG}_tr_${c}_$k.log 2>&1; done; done
          cd $R; TOP=45 python tools/step_traffic.py $O/${TAG}_tr_FETCH_SIZE_2 $O/${TAG}_tr_FETCH_SIZE_5 $O/${TAG}_tr_WRITE_SIZE_2 $O/${TAG}_tr_WRITE_SIZE_5 2 5 $O/${TAG}_step_traffic.txt > /dev/null 2>$O/${TAG}_step_traffic.err
          rm -rf $O/${TAG}_tr_FETCH_SIZE_2 $O/${TAG}_tr_FETCH_SIZE_5 $O/${TAG}_tr_WRITE_SIZE_2 $O/${TAG}_tr_WRITE_SIZE_5; head -16 $O/${TAG}_step_traffic.txt;;
    profsmall) cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_profs -- python $R/bench.py --batch 2 --points 20000 --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_profs.log 2>&1
          timeout 900 rocprofv3 --kernel-trace --stats -d $O/${TAG}_profs7 -- python $R/bench.py --batch 2 --points 20000 --steps 7 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_profs7.log 2>&1
          cd $R; TOP=30 python tools/steady_state_stats.py $O/${TAG}_profs 3 $O/${TAG}_profs7 7 $O/${TAG}_small_kernel_stats.csv > $O/${TAG}_profs_top.log 2>&1
          rm -rf $O/${TAG}_profs $O/${TAG}_profs7; tail -1 $O/${TAG}_small_kernel_stats.csv;;
    roof) cd /tmp
          timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_roof_stats -- python $R/tools/roofline_kernel.py > $O/${TAG}_roof_stats.log 2>&1
          timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_roof_fetch -- python $R/tools/roofline_kernel.py > $O/${TAG}_roof_fetch.log 2>&1
          timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_roof_write -- python $R/tools/roofline_kernel.py > $O/${TAG}_roof_write.log 2>&1
          timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/${TAG}_roof_sq -- python $R/tools/roofline_kernel.py > $O/${TAG}_roof_sq.log 2>&1
          timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $O/${TAG}_roof_sq2 -- python $R/tools/roofline_kernel.py > $O/${TAG}_roof_sq2.log 2>&1
          cd $R; python tools/pmc_summary.py --stats $O/${TAG}_roof_stats --pmc $O/${TAG}_roof_fetch $O/${TAG}_roof_write $O/${TAG}_roof_sq $O/${TAG}_roof_sq2 \
            --kernels attn_fwd_kernel,attn_bwd_dq_kernel,attn_bwd_dkv_kernel,attn_bwd1_kernel --out $O/${TAG}_roof_pmc.json > $O/${TAG}_roof_pmc.log 2>&1
          rm -rf $O/${TAG}_roof_stats $O/${TAG}_roof_fetch $O/${TAG}_roof_write $O/${TAG}_roof_sq $O/${TAG}_roof_sq2; tail -3 $O/${TAG}_roof_sq2.log;;
    gemmpmc) cd /tmp
          for cs in qkv fc2 wgrad; do
            export PTC_GK_CASE=$cs
            timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_gk_${cs}_stats -- python $R/tools/gemm_kernels.py > $O/${TAG}_gk_${cs}_stats.log 2>&1
            timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_gk_${cs}_fetch -- python $R/tools/gemm_kernels.py > $O/${TAG}_gk_${cs}_fetch.log 2>&1
            timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_gk_${cs}_write -- python $R/tools/gemm_kernels.py > $O/${TAG}_gk_${cs}_write.log 2>&1
            timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/${TAG}_gk_${cs}_sq -- python $R/tools/gemm_kernels.py > $O/${TAG}_gk_${cs}_sq.log 2>&1
            timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $O/${TAG}_gk_${cs}_sq2 -- python $R/tools/gemm_kernels.py > $O/${TAG}_gk_${cs}_sq2.log 2>&1
            cd $R; python tools/pmc_summary.py --stats $O/${TAG}_gk_${cs}_stats --pmc $O/${TAG}_gk_${cs}_fetch $O/${TAG}_gk_${cs}_write $O/${TAG}_gk_${cs}_sq $O/${TAG}_gk_${cs}_sq2 \
              --kernels gemm3_kernel,wgrad3_kernel,wgrad_reduce_kernel --out $O/${TAG}_gemm_pmc_${cs}.json > $O/${TAG}_gemm_pmc_${cs}.log 2>&1
            cd /tmp; rm -rf $O/${TAG}_gk_${cs}_stats $O/${TAG}_gk_${cs}_fetch $O/${TAG}_gk_${cs}_write $O/${TAG}_gk_${cs}_sq $O/${TAG}_gk_${cs}_sq2
          done; cd $R; cat $O/${TAG}_gemm_pmc_qkv.json | head -60;;
    gemmt:*) for v in $(echo "${sec#gemmt:}" | tr ',' ' '); do [ "$v" = default ] && export PTC_LIB_VARIANT= || export PTC_LIB_VARIANT=$v
            timeout 300 python tools/gemm_time.py 2>&1 | grep -v amdgpu.ids; done > $O/${TAG}_gemm_time.txt; export PTC_LIB_VARIANT=; cat $O/${TAG}_gemm_time.txt;;
    convpmc) cd /tmp
          rocprofv3 -L > $O/${TAG}_counters_list.txt 2>&1
          for cs in s0 s1; do
            export PTC_CK_CASE=$cs
            timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ck_${cs}_stats -- python $R/tools/conv_kernels.py > $O/${TAG}_ck_${cs}_stats.log 2>&1
            timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_ck_${cs}_fetch -- python $R/tools/conv_kernels.py > $O/${TAG}_ck_${cs}_fetch.log 2>&1
            timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_ck_${cs}_write -- python $R/tools/conv_kernels.py > $O/${TAG}_ck_${cs}_write.log 2>&1
            timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/${TAG}_ck_${cs}_tcc -- python $R/tools/conv_kernels.py > $O/${TAG}_ck_${cs}_tcc.log 2>&1
            timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $O/${TAG}_ck_${cs}_tcp -- python $R/tools/conv_kernels.py > $O/${TAG}_ck_${cs}_tcp.log 2>&1
            timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $O/${TAG}_ck_${cs}_sq -- python $R/tools/conv_kernels.py > $O/${TAG}_ck_${cs}_sq.log 2>&1
            timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $O/${TAG}_ck_${cs}_sq2 -- python $R/tools/conv_kernels.py > $O/${TAG}_ck_${cs}_sq2.log 2>&1
            cd $R; python tools/pmc_summary.py --stats $O/${TAG}_ck_${cs}_stats --pmc $O/${TAG}_ck_${cs}_fetch $O/${TAG}_ck_${cs}_write $O/${TAG}_ck_${cs}_tcc $O/${TAG}_ck_${cs}_tcp $O/${TAG}_ck_${cs}_sq $O/${TAG}_ck_${cs}_sq2 \
              --kernels conv8_kernel,conv7_kernel,wgrad7_kernel,rulebook_blocks_kernel,conv5_kernel,conv3_kernel,conv2_kernel,wgrad2_kernel,wgrad_reduce,linear2_kernel,rulebook_subm_kernel,hash_insert --out $O/${TAG}_conv_pmc_${cs}.json > $O/${TAG}_conv_pmc_${cs}.log 2>&1
            grep CONVKERNELS $O/${TAG}_ck_${cs}_stats.log > $O/${TAG}_conv_info_${cs}.txt
            rm -rf $O/${TAG}_ck_${cs}_stats $O/${TAG}_ck_${cs}_fetch $O/${TAG}_ck_${cs}_write $O/${TAG}_ck_${cs}_tcc $O/${TAG}_ck_${cs}_tcp $O/${TAG}_ck_${cs}_sq $O/${TAG}_ck_${cs}_sq2
            cd /tmp
          done; unset PTC_CK_CASE; tail -3 $O/${TAG}_ck_s1_tcp.log;;
    linpmc) cd /tmp
          for sh in 32,128 64,128 64,256; do
            export PTC_LK_SHAPE=$sh; tg=${TAG}_lk_${sh/,/_}
            timeout 300 rocprofv3 --kernel-trace --stats -d $O/${tg}_stats -- python $R/tools/linear_kernels.py > $O/${tg}_stats.log 2>&1
            timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${tg}_fetch -- python $R/tools/linear_kernels.py > $O/${tg}_fetch.log 2>&1
            timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/${tg}_write -- python $R/tools/linear_kernels.py > $O/${tg}_write.log 2>&1
            timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $O/${tg}_tcp -- python $R/tools/linear_kernels.py > $O/${tg}_tcp.log 2>&1
            timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/${tg}_tcc -- python $R/tools/linear_kernels.py > $O/${tg}_tcc.log 2>&1
            timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $O/${tg}_sq -- python $R/tools/linear_kernels.py > $O/${tg}_sq.log 2>&1
            timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $O/${tg}_sq2 -- python $R/tools/linear_kernels.py > $O/${tg}_sq2.log 2>&1
            cd $R; python tools/pmc_summary.py --stats $O/${tg}_stats --pmc $O/${tg}_fetch $O/${tg}_write $O/${tg}_tcp $O/${tg}_tcc $O/${tg}_sq $O/${tg}_sq2 \
              --kernels linear2_kernel --out $O/${TAG}_linear_pmc_${sh/,/_}.json > $O/${tg}_pmc.log 2>&1
            rm -rf $O/${tg}_stats $O/${tg}_fetch $O/${tg}_write $O/${tg}_tcp $O/${tg}_tcc $O/${tg}_sq $O/${tg}_sq2
            cd /tmp
          done; unset PTC_LK_SHAPE; ls $O | grep linear_pmc;;
    *) echo "unknown section $sec";;
  esac
done
cat $O/${TAG}_env.log
