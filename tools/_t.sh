cd $GRAFT_REPO_ROOT
( time timeout 1200 python bench.py > gpurun_out/r6_default_bench.json 2> gpurun_out/r6_default_bench.err ) 2> gpurun_out/r6_default_bench.time
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6_smoke.txt 2>&1
