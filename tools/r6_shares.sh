# round-6 (GPU box): the strong-scaling shares of BASELINE configs[3] on one GPU, under torch.distributed.run (world-1 RCCL all-reduce in the timed region)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_strong; mkdir -p $O
for n in 10000000 5000000 2500000 1250000; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --requests $n --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 > $O/strong_$n.json 2> $O/strong_$n.err
  python -c "
import json
d=json.load(open('$O/strong_$n.json')); print('share $n', round(d['ms_per_step'],4), d['traffic_modes']['tuned_benign']['kernels_ms_per_step'])"
done
