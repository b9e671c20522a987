# split against split pages as the share of observed landmarks grows (where should SLAM_MAP_AUTO change over?)
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
for K in 128 160 192 224 256; do
  for lay in split split_pages; do
    python3 bench.py $Q --observed $K --map-layout $lay > gpurun_out/th_${K}_$lay.json 2> gpurun_out/th_${K}_$lay.err || echo fail $K $lay
  done
done
for K in 1000 1500 2000 2500; do
  for lay in split split_pages; do
    python3 bench.py $Q --landmarks 5000 --observed $K --map-layout $lay --steps 40 > gpurun_out/th5k_${K}_$lay.json 2> gpurun_out/th5k_${K}_$lay.err || echo fail 5k $K $lay
  done
done
for K in 250 333 500; do
  for lay in split split_pages; do
    python3 bench.py $Q --scaling strong --particles-total 1048576 --landmarks 1000 --observed $K --map-layout $lay --steps 30 > gpurun_out/thns_${K}_$lay.json 2> gpurun_out/thns_${K}_$lay.err || echo fail ns $K $lay
  done
done
echo done
