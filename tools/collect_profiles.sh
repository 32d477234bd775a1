#!/bin/bash
# usage (GPU box, repo root): tools/collect_profiles.sh <tag>   -- every profile / bench artefact of a round into gpurun_out/<tag>_*
# (copy what is to be judged into profiles/ afterwards)
T=$1
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
# 1. the judged bench line (N = 1) and the same under rocprofv3 (per-kernel stats must agree with the in-bench HIP events)
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
# (1000 steps: the layer kernel's first ~30 launches of a process run up to 15 % long -- clock and cache warm-up -- and the CSV's average is over every launch)
tools/prof_bench.sh ${T} --steps 1000 --warmup 20 > gpurun_out/${T}_profiled_run_summary.txt 2>&1
cp gpurun_out/prof_${T}_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv
DB=$(find gpurun_out/prof_${T} -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB -4 > gpurun_out/${T}_step_timeline.txt
# 2. steps kept in flight
tools/prof_pipe.sh ${T} cfg2_mnist_CH_M256 40 2 > /dev/null 2>&1
DB=$(find gpurun_out/pipe_${T} -name '*.db' | head -1)
python tools/rocpd_timeline.py $DB -4 1700 > gpurun_out/${T}_two_in_flight_timeline.txt
tail -1 gpurun_out/pipe_${T}.log >> gpurun_out/${T}_two_in_flight_timeline.txt
# 3. phases inside the one-launch layer kernel
python tools/fused_trace.py cfg2_mnist_CH_M256 > gpurun_out/${T}_fused_phase_trace.txt 2>&1
# 4. counters, one set per pass, every BASELINE configuration (head-only ones included)
for c in cfg2_mnist_CH_M256 cfg2_mnist_H_M256 cfg1_mnist_H_M32 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384 cfg5_mnist_H_M1024 cfg5_mnist_CH_M1024; do
  tools/pmc_bench.sh ${T}f_$c "FETCH_SIZE" --steps 2 --warmup 1 --config $c > gpurun_out/${T}_pmc_fetch_$c.txt 2>&1
  tools/pmc_bench.sh ${T}w_$c "WRITE_SIZE" --steps 2 --warmup 1 --config $c > gpurun_out/${T}_pmc_write_$c.txt 2>&1
  tools/pmc_bench.sh ${T}s_$c "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 2 --warmup 1 --config $c > gpurun_out/${T}_pmc_sq_$c.txt 2>&1
done
# the materialised K_uf sweep of the headline configuration (its step uses the one-launch layer): traffic of bench.py's roofline_kuf
DCGP_NO_FUSED_LAYER=1 tools/pmc_bench.sh ${T}f_cfg2_mnist_CH_M256_unfused "FETCH_SIZE" --steps 2 --warmup 1 --config cfg2_mnist_CH_M256 > /dev/null 2>&1
DCGP_NO_FUSED_LAYER=1 tools/pmc_bench.sh ${T}w_cfg2_mnist_CH_M256_unfused "WRITE_SIZE" --steps 2 --warmup 1 --config cfg2_mnist_CH_M256 > /dev/null 2>&1
python tools/pmc_summary.py ${T} > gpurun_out/${T}_pmc_summary.txt 2>&1
python tools/pmc_traffic.py ${T} > gpurun_out/${T}_pmc_traffic.json 2> gpurun_out/${T}_pmc_traffic.err
# (cfg3 / cfg4 step on the fused / GEMM routes: their materialised first-layer sweeps for the K_uf traffic rows)
DCGP_NO_FUSED_LAYER=1 tools/pmc_bench.sh ${T}f_cfg3_mnist_3layer_M256_unfused "FETCH_SIZE" --steps 2 --warmup 1 --config cfg3_mnist_3layer_M256 > /dev/null 2>&1
DCGP_NO_FUSED_LAYER=1 tools/pmc_bench.sh ${T}w_cfg3_mnist_3layer_M256_unfused "WRITE_SIZE" --steps 2 --warmup 1 --config cfg3_mnist_3layer_M256 > /dev/null 2>&1
python tools/pmc_traffic.py ${T} > gpurun_out/${T}_pmc_traffic.json 2> gpurun_out/${T}_pmc_traffic.err
# 5. the other BASELINE configurations
for c in cfg1_mnist_H_M32 cfg2_mnist_H_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384 cfg5_mnist_H_M1024 cfg5_mnist_CH_M1024; do
  timeout 300 python bench.py --config $c --steps 30 --no-cpu-baseline --no-grad-leg --no-extra-legs > gpurun_out/${T}_bench_$c.json 2> gpurun_out/${T}_bench_$c.err
done
# 6. the training step (forward + reverse pass), tiled and with the exact layer-0 de-duplication
tools/prof_grad.sh ${T}g cfg2_mnist_CH_M256 20 > gpurun_out/${T}_grad_step_summary.txt 2>&1
DCGP_DEDUP=1 tools/prof_grad.sh ${T}gd cfg2_mnist_CH_M256 20 > gpurun_out/${T}_grad_step_dedup_summary.txt 2>&1
# 7. two ranks on this one GPU: self-launched and under the driver's launcher (RCCL refuses two ranks on one device -> host join)
timeout 300 python bench.py --gpus 2 --steps 50 --no-cpu-baseline > gpurun_out/${T}_bench_2ranks_selflaunch.json 2> gpurun_out/${T}_bench_2ranks_selflaunch.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --no-cpu-baseline > gpurun_out/${T}_bench_2ranks_torchrun.json 2> gpurun_out/${T}_bench_2ranks_torchrun.err
# 8. the patch sweeps alone on the chip (every configuration), their per-workgroup traces, the store-bandwidth ceiling, real-data learning
python tools/sweep_times.py > gpurun_out/${T}_sweep_times.txt 2>&1
for c in "cfg2_mnist_CH_M256 kuf" "cfg4_cifar_3layer_M384 kuf" "cfg2_mnist_H_M256 head_sweep" "cfg2_mnist_CH_M256 head_sweep"; do python tools/sweep_trace.py $c; done > gpurun_out/${T}_sweep_trace.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o /tmp/store_bw 2>/dev/null && /tmp/store_bw > gpurun_out/${T}_store_bw.txt 2>&1
python tools/digits_train.py 1500 > gpurun_out/${T}_digits_learning.txt 2>&1
# 9. a rank's shard of the headline batch (4 of 32 images): one step synchronous and steps kept in flight
tools/prof_pipe.sh ${T}s4 cfg2_mnist_CH_M256 60 2 4 > /dev/null 2>&1
DB=$(find gpurun_out/pipe_${T}s4 -name '*.db' | head -1); python tools/rocpd_timeline.py $DB -6 450 > gpurun_out/${T}_shard4_two_in_flight_timeline.txt; tail -1 gpurun_out/pipe_${T}s4.log >> gpurun_out/${T}_shard4_two_in_flight_timeline.txt
tools/prof_pipe.sh ${T}s4s cfg2_mnist_CH_M256 60 1 4 > /dev/null 2>&1
DB=$(find gpurun_out/pipe_${T}s4s -name '*.db' | head -1); python tools/rocpd_timeline.py $DB -6 > gpurun_out/${T}_shard4_step_timeline.txt; tail -1 gpurun_out/pipe_${T}s4s.log >> gpurun_out/${T}_shard4_step_timeline.txt
# 10. one step of the other configurations and of the training step, kernel by kernel (of THIS code: collected last)
for c in cfg1_mnist_H_M32 cfg2_mnist_H_M256 cfg3_mnist_3layer_M256 cfg4_cifar_3layer_M384; do
  bash tools/step_timeline.sh ${T}tl_$c $c > /dev/null 2>&1; cp gpurun_out/${T}tl_${c}_timeline.txt gpurun_out/${T}_${c}_step_timeline.txt
done
bash tools/grad_step_timeline.sh ${T}gtl > /dev/null 2>&1; cp gpurun_out/${T}gtl_timeline.txt gpurun_out/${T}_grad_step_timeline.txt
bash tools/grad_step_timeline.sh ${T}gtld DCGP_DEDUP=1 > /dev/null 2>&1; cp gpurun_out/${T}gtld_timeline.txt gpurun_out/${T}_grad_step_dedup_timeline.txt
# 11. the layer kernel under its launch options, and whether the part is power-bound under it
python tools/fused_ab.py > gpurun_out/${T}_fused_launch_options.txt 2>&1
python tools/fused_power.py > gpurun_out/${T}_power_and_clock_final.txt 2>&1
# rocprofv3 databases are large: keep the summaries only
for d in gpurun_out/prof_${T}* gpurun_out/pipe_${T}* gpurun_out/pmc_${T}*; do [ -d "$d" ] && rm -rf "$d"; done
echo collected
