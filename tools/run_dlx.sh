cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dlx4 -o dlx -- python $R/tools/dlexact.py 1e9 2e6 864.6 > $R/gpurun_out/dlx_prof4.log 2>&1; grep "^n=" $R/gpurun_out/dlx_prof4.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_dlx_FETCH -o p -- python $R/tools/dlexact.py 1e9 2e6 864.6 > $R/gpurun_out/pmc_dlx_FETCH.log 2>&1; grep "^n=" $R/gpurun_out/pmc_dlx_FETCH.log | cut -c1-120
