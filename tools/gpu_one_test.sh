cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 500 python -m pytest tests -m gpu -q --timeout=200 -p no:cacheprovider -k "comp_forms or single_rank_over_rccl or never_returns or (gpus_switch and 1-env0)" 2>&1 | tail -40) > gpurun_out/one_test.log 2>&1
tail -15 gpurun_out/one_test.log | cut -c1-300
