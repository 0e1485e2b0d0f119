cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r02 > gpurun_out/prof_r02.log 2>&1 || { tail gpurun_out/prof_r02.log; exit 1; }
tail -3 gpurun_out/prof_r02.log
timeout -k 10 900 python -u tools/diagnose_outliers.py 25 2000 > gpurun_out/r02_outlier_diagnosis.txt 2> gpurun_out/diag.err || { tail gpurun_out/diag.err; exit 1; }
tail -5 gpurun_out/r02_outlier_diagnosis.txt
