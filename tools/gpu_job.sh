set -u
bash tools/prof.sh round6 > gpurun_out/prof_run.log 2>&1
tail -5 gpurun_out/prof_run.log
du -sh gpurun_out/prof
