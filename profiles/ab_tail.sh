# the classes' update inside the launch of the weights (default) against a launch of its own (SLAM_COV_MERGE=0), and the free list
# kernel without fences: dense default, 32 observed (split pages)
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
for rep in 1 2; do
  for m in 1 0; do
    SLAM_COV_MERGE=$m python3 bench.py $Q > gpurun_out/tail_dense_m${m}_$rep.json 2> gpurun_out/tail_dense_m${m}_$rep.err || echo fail dense $m
    SLAM_COV_MERGE=$m python3 bench.py $Q --observed 32 > gpurun_out/tail_o32_m${m}_$rep.json 2> gpurun_out/tail_o32_m${m}_$rep.err || echo fail o32 $m
  done
done
SLAM_COV_MERGE=1 python3 bench.py $Q --scaling strong --particles-total 1048576 --landmarks 1000 --steps 30 > gpurun_out/tail_ns_m1.json 2> gpurun_out/tail_ns_m1.err || echo fail ns
SLAM_COV_MERGE=0 python3 bench.py $Q --scaling strong --particles-total 1048576 --landmarks 1000 --steps 30 > gpurun_out/tail_ns_m0.json 2> gpurun_out/tail_ns_m0.err || echo fail ns
echo done
