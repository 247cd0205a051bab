set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/pytest_gpu.txt
python tools/kbench_gated.py > $O/kbench_gated.txt 2>&1
bash tools/pmc_gated.sh 393216 > /dev/null 2>&1
rm -rf $O/pmc_gs_gated_sq $O/pmc_gs_gated_lds $O/pmc_gs_gated_mem $O/pmc_gs_gated_wait $O/pmc_gs_ungated_sq $O/pmc_gs_ungated_lds $O/pmc_gs_ungated_mem $O/pmc_gs_ungated_wait
(echo '# attention-score kernel: default shapes vs the 8-wave shapes of round 2 (VLSA_GS_HG2=0: ungated module, VLSA_GS_G4=0: gated module), same box, alternating processes, us per bag'; python tools/kbench_gated_ab.py - VLSA_GS_HG2=0,VLSA_GS_G4=0 2>&1 | grep 'gated=') > $O/kbench_gated_ab.txt
for n in 40000 50000 70000 100000; do for sp in 0 1; do VLSA_GS_SPLIT=$sp python tools/gs_rows.py $n 2>/dev/null | sed "s/$/ split=$sp/"; done; done > $O/gs_split.txt
python tools/bench_deepmil.py > $O/bench_deepmil.txt 2>&1
python tools/kbench_mlp_bwd.py > $O/kbench_mlp_bwd.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
cat $O/pytest_gpu.txt; tail -2 $O/smoke.txt; cut -c1-200 $O/bench.json
