mkdir -p gpurun_out/calib; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/calib/fetch -o c -- $R/scripts/ubench/_bin/fetch_calib > $R/gpurun_out/calib/fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/calib/write -o c -- $R/scripts/ubench/_bin/fetch_calib > $R/gpurun_out/calib/write.log 2>&1
cd $R; python scripts/calibrate_counters.py gpurun_out/calib
