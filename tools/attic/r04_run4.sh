cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_text_tower.py -q -m gpu -x 2>&1 | tail -5) > $O/pytest_tt.txt
tail -3 $O/pytest_tt.txt
for m in 1 0; do VLSA_TT_PERSIST=$m timeout 300 python tools/bench_text.py 2>&1 | grep -v amdgpu | sed "s/^/PERSIST=$m: /"; done > $O/bench_text_persist.txt; cat $O/bench_text_persist.txt
timeout 300 python tools/tt_persist_stamps.py 2>&1 | grep -v amdgpu > $O/tt_persist_stamps.txt; cat $O/tt_persist_stamps.txt
