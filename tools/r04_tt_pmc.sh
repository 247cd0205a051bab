# scratch runner: HBM-side traffic of the text tower per pass (forward; with "bwd": forward + backward)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
ARG=""; TAG=""; if [ "${1:-}" = "bwd" ]; then ARG="--bwd"; TAG="_bwd"; fi
rm -rf /tmp/pmc_tt
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tt/f -- python tools/run_text.py $ARG > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tt/w -- python tools/run_text.py $ARG > /dev/null 2>&1
python tools/run_text.py summarise /tmp/pmc_tt | tee $O/pmc_text_tower$TAG.json
