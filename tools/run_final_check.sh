cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1 || { tail -20 gpurun_out/smoke.txt; exit 1; }
tail -1 gpurun_out/smoke.txt
bash tools/run_full_check.sh
