# the north star's per-rank shard at N = 1, 2, 4, 8 (1 048 576 / N particles x 1 000 landmarks) as one rank's frame on one card, on the
# multi-GPU code path (one-rank RCCL communicator): what a rank computes per frame; what N ranks add is the wire
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
for n in 1048576 524288 262144 131072; do
  python3 bench.py $Q --particles $n --landmarks 1000 --steps 40 --force-collectives > gpurun_out/shard_fc_$n.json 2> gpurun_out/shard_fc_$n.err || echo fail fc $n
  python3 bench.py $Q --particles $n --landmarks 1000 --steps 40 > gpurun_out/shard_1_$n.json 2> gpurun_out/shard_1_$n.err || echo fail $n
done
echo done
