# split against split pages as the share of observed landmarks grows (where should SLAM_MAP_AUTO change over?)
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
for K in 64 128 192 256 384 500; do
  for lay in split split_pages; do
    python3 bench.py $Q --observed $K --map-layout $lay > gpurun_out/th_${K}_$lay.json 2> gpurun_out/th_${K}_$lay.err || echo fail $K $lay
  done
done
for K in 250 500 1000 2000; do
  for lay in split split_pages; do
    python3 bench.py $Q --landmarks 5000 --observed $K --map-layout $lay --steps 40 > gpurun_out/th5k_${K}_$lay.json 2> gpurun_out/th5k_${K}_$lay.err || echo fail 5k $K $lay
  done
done
echo done
