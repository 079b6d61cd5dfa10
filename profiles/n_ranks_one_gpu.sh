set -x
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 3 --size 2048 > gpurun_out/r04_n2_rows.json 2> gpurun_out/r04_n2_rows.err; echo rc=$?
tail -3 gpurun_out/r04_n2_rows.err; cat gpurun_out/r04_n2_rows.json
for pol in auto bands spread; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 3 --dist-backend gloo --workload fanin --policy $pol --steps 10 --warmup 3 --size 1024 > gpurun_out/r04_n3_fanin_$pol.json 2> gpurun_out/r04_n3_fanin_$pol.err; echo rc=$?
tail -3 gpurun_out/r04_n3_fanin_$pol.err; cat gpurun_out/r04_n3_fanin_$pol.json
done
