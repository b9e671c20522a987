# the free list of a paged frame inside the scorer's launch (default) against a launch of its own (SLAM_FREE_LIST_RIDER=0)
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
for rep in 1 2; do
  for m in 1 0; do
    SLAM_FREE_LIST_RIDER=$m python3 bench.py $Q --observed 32 > gpurun_out/rider_o32_m${m}_$rep.json 2> gpurun_out/rider_o32_m${m}_$rep.err || echo fail o32 $m
    SLAM_FREE_LIST_RIDER=$m python3 bench.py $Q --landmarks 5000 --observed 32 --steps 40 > gpurun_out/rider_5k_m${m}_$rep.json 2> gpurun_out/rider_5k_m${m}_$rep.err || echo fail 5k $m
  done
done
for m in 1 0; do
SLAM_FREE_LIST_RIDER=$m python3 bench.py $Q --scaling strong --particles-total 1048576 --landmarks 1000 --observed 32 --steps 30 > gpurun_out/rider_ns_m$m.json 2> gpurun_out/rider_ns_m$m.err || echo fail ns
SLAM_FREE_LIST_RIDER=$m python3 bench.py $Q --observed 32 --paged > gpurun_out/rider_o32pages_m$m.json 2> gpurun_out/rider_o32pages_m$m.err || echo fail pages
done
echo done
