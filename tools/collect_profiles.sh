#!/bin/bash
# Run ON THE GPU BOX (via gpurun): full GPU test suite, bench lines, rocprofv3 kernel stats of the bench command, separate PMC
# passes for the dominant kernels, and the per-path benches.  Outputs under gpurun_out/${VLSA_ROUND:-r06}/ (copied into profiles/ afterwards).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${VLSA_ROUND:-r06}; mkdir -p $O
(VLSA_GRAD_ERRORS_OUT=$O/grad_errors.txt timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-extra --streams 1 > $O/bench_profiled_streams1.json 2>/dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
pmc() { # tag, counters..., -- command
  tag=$1; shift; ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -- "$@" > /dev/null 2>&1
}
pmc fetch FETCH_SIZE -- python tools/run_batch.py 32 50000
pmc write WRITE_SIZE -- python tools/run_batch.py 32 50000
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -- python tools/run_batch.py 32 50000
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -- python tools/run_batch.py 32 50000
pmc fetch64 FETCH_SIZE -- python tools/run_batch.py 64 50000
pmc write64 WRITE_SIZE -- python tools/run_batch.py 64 50000
pmc lds64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -- python tools/run_batch.py 64 50000
pmc gs_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -- python tools/run_gated.py 50000
pmc gs_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_WAVES GRBM_GUI_ACTIVE -- python tools/run_gated.py 50000
pmc gs_mem FETCH_SIZE -- python tools/run_gated.py 50000
python - <<PY
import csv, glob, collections, json
def collect(tags, pat):
    out = {}
    for tag in tags:
        fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
        if not fs: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[8:] or v            # skip warm-up launches
            out[k] = sum(v) / len(v)
    return out
json.dump(collect(("fetch", "write", "sq", "lds"), "partial_dma_batch"), open("$O/pmc_batch_kernel.json", "w"), indent=1)
json.dump(collect(("fetch64", "write64", "lds64"), "partial_dma_batch"), open("$O/pmc_batch_kernel_b64.json", "w"), indent=1)
json.dump(collect(("gs_sq", "gs_lds", "gs_mem"), "_scores"), open("$O/pmc_gated_scores.json", "w"), indent=1)   # (k_scores_tile_p at this size)
PY
# PMC passes for the fp32 streaming kernel
pmc f32_fetch FETCH_SIZE -- python tools/run_batch.py 32 50000 0 f32
pmc f32_write WRITE_SIZE -- python tools/run_batch.py 32 50000 0 f32
pmc f32_sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python tools/run_batch.py 32 50000 0 f32
python - <<PY
import csv, glob, collections, json
out = {}
for tag in ("f32_fetch", "f32_write", "f32_sq"):
    fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "partial_f32_batch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[8:] or v
        out[k] = sum(v) / len(v)
json.dump(out, open("$O/pmc_batch_kernel_f32.json", "w"), indent=1)
PY
rm -rf $O/pmc_f32_fetch $O/pmc_f32_write $O/pmc_f32_sq
# PMC passes for the two persistent backward kernels (bf16 / fp32 batches): traffic vs the algorithmic bytes, matrix-pipe share
pmc bw_fetch FETCH_SIZE -- python tools/prof_train.py 50000 20
pmc bw_write WRITE_SIZE -- python tools/prof_train.py 50000 20
pmc bw_sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python tools/prof_train.py 50000 20
pmc bwf_fetch FETCH_SIZE -- python tools/prof_train.py 50000 20 fp32
pmc bwf_sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/prof_train.py 50000 20 fp32
python - <<PY
import csv, glob, collections, json
def collect(tags, pat):
    out = {}
    for tag in tags:
        fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
        if not fs: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[4:] or v
            out[k] = sum(v) / len(v)
    return out
json.dump({"k_vlfan_backward_dma_batch (bf16, 32 x 50k bags)": collect(("bw_fetch", "bw_write", "bw_sq"), "backward_dma_batch"),
           "k_vlfan_backward_f32_batch (fp32, 32 x 50k bags)": collect(("bwf_fetch", "bwf_sq"), "backward_f32_batch")},
          open("$O/pmc_backward_kernels.json", "w"), indent=1)
PY
rm -rf $O/pmc_bw_fetch $O/pmc_bw_write $O/pmc_bw_sq $O/pmc_bwf_fetch $O/pmc_bwf_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $O/text -- python tools/bench_text.py > /dev/null 2>&1
cp $(find $O/text -name "*kernel_stats.csv" | head -1) $O/text_kernel_stats.csv
python tools/bench_text.py --cpu > $O/bench_text.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python tools/prof_train.py > /dev/null 2>&1
cp $(find $O/train -name "*kernel_stats.csv" | head -1) $O/train_batch_kernel_stats.csv
python tools/bench_attn.py > $O/bench_attn.txt 2>&1
python tools/bench_attn.py 50000 fp32 2>&1 | grep want_attn >> $O/bench_attn.txt
python tools/bench_step.py > $O/bench_step.txt 2>&1
# round 6: the bench.py train_step leg alone (BASELINE configs[4]): hipGraph replay (default) and eager, and the per-step kernel list
(VLSA_BENCH_TRAIN_MODE=graph python tools/bench_train_step.py both 30) > $O/bench_train_step_graph.txt 2>&1
(VLSA_BENCH_TRAIN_MODE=eager python tools/bench_train_step.py both 30) > $O/bench_train_step_eager.txt 2>&1
VLSA_BENCH_TRAIN_MODE=graph rocprofv3 --kernel-trace --output-format csv -d $O/prof_step -- python tools/bench_train_step.py tcga 60 > /dev/null 2>&1
python tools/step_kernels.py $O/prof_step > $O/step_kernels.txt 2>&1; rm -rf $O/prof_step
# round 6: one slide per call -- wall / host per call, the GPU chain from a trace, shader-cycle stamps of the streaming kernel
python tools/prof_single_slide.py 2>&1 | grep "N=" > $O/single_slide.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace_ss -- python tools/prof_single_slide.py 50k > /dev/null 2>&1
python tools/trace_single_slide.py $O/trace_ss >> $O/single_slide.txt 2>&1; rm -rf $O/trace_ss
python tools/kbench_tail.py 2>&1 | tail -3 >> $O/single_slide.txt
[ -f vlsa_amd/_lib/variants/libvlsa_dmatiming.so ] && python tools/dma_stamps.py 2>&1 | grep -v amdgpu > $O/dma_stamps.txt
python tools/bench_epoch.py 2>&1 | grep -v amdgpu > $O/bench_epoch.txt
python tools/bench_train.py > $O/bench_train.txt 2>&1
python tools/kbench_batch_f32.py 2>&1 | grep "N=" > $O/kbench_batch_f32.txt
python tools/kbench_wide.py 2>&1 | grep "N=" > $O/kbench_wide.txt
python tools/kbench_wide.py 50000 20000 2>&1 | grep "bfloat" > $O/kbench_wide_50k.txt
python tools/kbench_gated.py > $O/kbench_gated.txt 2>&1
# round 5: the persistent LDS-DMA score kernel next to k_gated_scores on this box, its timing-only ablations and cycle stamps (variant
# libraries: `python tools/gt_ablate.py build; python tools/gt_stamps.py build` in the CPU container), its counters
# (the kernel-choice A/B through VLSA_GS_TILE needs a -DVLSA_EXPERIMENT build since round 6: profiles/r05_gt_ab.txt is the record)
[ -f vlsa_amd/_lib/variants/libvlsa_gtstamp.so ] && python tools/gt_stamps.py 393216 gated 2>&1 | grep -v amdgpu > $O/gt_stamps.txt
bash tools/pmc_tile.sh 393216 gated > /dev/null 2>&1
[ -f tools/probes/libmfma_issue.so ] && python tools/probes/mfma_issue.py 2>&1 | grep -v amdgpu > $O/mfma_issue.txt
python tools/kbench_featproj.py 2>&1 | grep "N=" > $O/kbench_featproj.txt
python tools/bench_deepmil.py > $O/bench_deepmil.txt 2>&1
(python tools/bench_deepmil.py | grep bfloat16; echo "VLSA_GS_NO_FUSED_POOL=1:"; VLSA_GS_NO_FUSED_POOL=1 python tools/bench_deepmil.py | grep bfloat16) > $O/bench_deepmil_fused.txt 2>&1
python tools/kbench_pool.py 2>&1 | grep -v amdgpu > $O/kbench_pool.txt
python tools/bench_module.py > $O/bench_module.txt 2>&1
python tools/bench_paths.py > $O/bench_paths.txt 2>&1
python tools/bench_zeroshot.py > $O/bench_zeroshot.txt 2>&1
VLSA_BENCH_FORCE_SHARDED=1 python bench.py --no-cpu-baseline > $O/bench_sharded_1rank.json 2> $O/bench_sh1.err
# the N > 1 code path started the way the driver starts it (python bench.py --gpus 2: self-launch), two ranks sharing this GPU over gloo
VLSA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
VLSA_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 10 --warmup 3 > $O/bench_4ranks_one_gpu.json 2> $O/bench_4ranks.err
# ... and with 8 ranks sharing the GPU (short clock ramp: every gloo exchange of 8 processes on one device takes ~0.1-1 s): world = 8
# shard bounds, the 8-record fold and the gathered verification walk through; timings meaningless
VLSA_BENCH_RAMP=1 VLSA_BENCH_WATCHDOG=300 VLSA_BENCH_BACKEND=gloo timeout 500 python bench.py --gpus 8 --steps 2 --warmup 1 > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks.err
# round 4: persistent text-tower forward vs launch-per-stage (+ its in-kernel stamps), the whole-row score kernel vs the default
python tools/bench_text_trainable.py 2>&1 | tail -1 > $O/bench_text_trainable.txt
pmc tt_fetch FETCH_SIZE -- python tools/run_text.py
pmc tt_write WRITE_SIZE -- python tools/run_text.py
mkdir -p $O/pmc_tt && cp -r $O/pmc_tt_fetch $O/pmc_tt_write $O/pmc_tt/ 2>/dev/null
python tools/run_text.py summarise $O/pmc_tt > $O/pmc_text_tower.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/text_train -- python tools/bench_text_trainable.py > /dev/null 2>&1
cp $(find $O/text_train -name "*kernel_stats.csv" | head -1) $O/text_train_kernel_stats.csv 2>/dev/null
# ---- round 3: backward kernels of the N-sized layers, attention-weights traffic, text tower with the shared prefix
python tools/kbench_mlp_bwd.py > $O/kbench_mlp_bwd.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mb -- python tools/run_mlp_bwd.py 50000 > /dev/null 2>&1
cp $(find $O/mb -name "*kernel_stats.csv" | head -1) $O/mlp_bwd_kernel_stats.csv
pmc mb_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -- python tools/run_mlp_bwd_one.py 50000 gated
pmc mb_b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -- python tools/run_mlp_bwd_one.py 50000 gated
pmc mb_c FETCH_SIZE -- python tools/run_mlp_bwd_one.py 50000 gated
pmc mb_d WRITE_SIZE -- python tools/run_mlp_bwd_one.py 50000 gated
pmc at_f FETCH_SIZE -- python tools/run_batch_attn.py
pmc at_w WRITE_SIZE -- python tools/run_batch_attn.py
python - <<PY
import csv, glob, collections, json
def collect(tags, pats):
    out = {}
    for tag in tags:
        fs = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
        if not fs: continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fs[0])):
            for p in pats:
                if p in r["Kernel_Name"]:
                    acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for p, cs in acc.items():
            for k, v in cs.items():
                v = v[3:] or v
                out.setdefault(p, {})[k] = sum(v) / len(v)
    return out
json.dump(collect(("mb_a", "mb_b", "mb_c", "mb_d"), ["k_mlp_backward"]), open("$O/pmc_mlp_backward_gated.json", "w"), indent=1)
json.dump(collect(("at_f", "at_w"), ["k_vlfan_partial_dma_batch<true>", "k_attn_normalise_batch", "k_vlfan_merge_pool_batch", "k_vlfan_merge_pool_small"]),
          open("$O/pmc_batch_attn_traffic.json", "w"), indent=1)
PY
rm -rf $O/mb $O/pmc_mb_a $O/pmc_mb_b $O/pmc_mb_c $O/pmc_mb_d $O/pmc_at_f $O/pmc_at_w
# what the machine gives next to the product: read-ceiling probe (built here), the product on cache-resident rows
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/hbm_read_probe.hip -o tools/probes/libhbm_read_probe.so 2>/dev/null
python tools/hbm_read_probe.py 2>&1 | grep -v amdgpu > $O/hbm_read_probe.txt
python tools/kbench_resident.py 2>&1 | grep -v amdgpu > $O/kbench_resident.txt
# the score kernel: both modules' counters at 393 216 patches (steady state), the ungated module's shapes side by side
bash tools/pmc_gated.sh 393216 > /dev/null 2>&1
rm -rf $O/pmc_gs_gated_sq $O/pmc_gs_gated_lds $O/pmc_gs_gated_mem $O/pmc_gs_gated_wait $O/pmc_gs_ungated_sq $O/pmc_gs_ungated_lds $O/pmc_gs_ungated_mem $O/pmc_gs_ungated_wait
rm -rf $O/train $O/stats $O/text $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds $O/pmc_fetch64 $O/pmc_write64 $O/pmc_lds64 $O/pmc_gs_sq $O/pmc_gs_lds $O/pmc_gs_mem
# ---- round 5: the launch chains behind the streaming kernels (one slide per call, 256 slide-sized bags per launch), the single-slide tail
# alone, the machine's price of the score-line write stream, the GPU timeline of the handler's evaluation loop
python tools/prof_tails.py 2>&1 | grep -v amdgpu > $O/bench_tails.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tails -- python tools/prof_tails.py single wide > /dev/null 2>&1
cp $(find $O/tails -name "*kernel_stats.csv" | head -1) $O/tails_kernel_stats.csv; rm -rf $O/tails
python tools/kbench_tail.py 2>&1 | grep -v amdgpu > $O/kbench_tail.txt
python tools/hbm_rw_probe.py 2>&1 | grep -v amdgpu > $O/hbm_rw_probe.txt
python tools/prof_eval_loop.py 2>&1 | grep "us per" > $O/eval_loop_timeline.txt
rocprofv3 --kernel-trace --output-format csv -d $O/evl -- python tools/prof_eval_loop.py > /dev/null 2>&1
python tools/prof_eval_loop.py gaps $(find $O/evl -name "*kernel_trace.csv" | head -1) >> $O/eval_loop_timeline.txt; rm -rf $O/evl
python tools/prof_hit.py 2798 2>&1 | grep -v amdgpu | head -24 > $O/prof_hit.txt
cat $O/pytest_gpu.txt; cut -c1-400 $O/bench.json; cut -c1-200 $O/bench_driver_args.json; cat $O/pmc_gated_scores.json | head -30
