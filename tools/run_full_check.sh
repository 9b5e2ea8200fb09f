cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.txt 2>&1 || { tail -30 gpurun_out/t_all.txt; exit 1; }
timeout -k 10 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || { tail -20 gpurun_out/bench_full.err; exit 1; }
tail -c 3000 gpurun_out/bench_full.json
