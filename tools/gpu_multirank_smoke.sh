# exercise bench.py's N>1 code path on a 1-GPU box: 2 ranks, gloo, both on device 0, small graph
export GSPX_ALL_RANKS_DEVICE0=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 2 --warmup 1 --vertices 200000 --backend gloo
