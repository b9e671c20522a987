Q="--no-cpu-baseline --no-extra-legs"
for lay in pages split_pages auto; do
  python3 bench.py $Q --observed 32 --map-layout $lay > gpurun_out/sp_o32_$lay.json 2> gpurun_out/sp_o32_$lay.err || echo fail $lay
  python3 bench.py $Q --landmarks 5000 --observed 32 --map-layout $lay --steps 40 > gpurun_out/sp_5k_$lay.json 2> gpurun_out/sp_5k_$lay.err || echo fail 5k $lay
  python3 bench.py $Q --scaling strong --particles-total 1048576 --landmarks 1000 --observed 32 --map-layout $lay --steps 30 > gpurun_out/sp_ns_$lay.json 2> gpurun_out/sp_ns_$lay.err || echo fail ns $lay
done
python3 bench.py $Q --observed 128 --map-layout split_pages > gpurun_out/sp_o128_split_pages.json 2>gpurun_out/sp_o128.err || echo fail
echo done
