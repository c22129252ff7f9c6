cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1
wc -l $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt
