cd $GRAFT_REPO_ROOT
( for c in cfg2 cfg4 f64 speech mm mm64 fbank nemo nemo_norm w512 nm64 nm40; do MELSPEC_LIB_OLDER=1 AB_REPS=3 python tools/ab_run.py --case $c r04 now 2>&1 | tail -2; done ) > gpurun_out/r05_vs_r04_final.txt 2>&1
tools/round_profiles.sh r05 > gpurun_out/round_r05.log 2>&1
tail -5 gpurun_out/round_r05.log; cat gpurun_out/r05_vs_r04_final.txt
