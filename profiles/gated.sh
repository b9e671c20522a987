# ESS-gated sessions (AUTO keeps them on rows): what the other layouts would give
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
for ess in 0.3 0.1; do
  for lay in rows split_pages pages; do
    python3 bench.py $Q --observed 32 --ess $ess --map-layout $lay --steps 120 > gpurun_out/gated_o32_${ess}_$lay.json 2> gpurun_out/gated_o32_${ess}_$lay.err || echo fail o32 $ess $lay
  done
  for lay in rows split; do
    python3 bench.py $Q --ess $ess --map-layout $lay --steps 120 > gpurun_out/gated_dense_${ess}_$lay.json 2> gpurun_out/gated_dense_${ess}_$lay.err || echo fail dense $ess $lay
  done
done
echo done
