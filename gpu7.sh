cd $GRAFT_REPO_ROOT
AQ_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 2>&1 | grep -v Warning | tail -3 | cut -c1-600
