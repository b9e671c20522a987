# dense default, north star, config 5 share: one run each (compare with the previous build's lines on the same box by running this
# script from both trees in one call: profiles/ab.py)
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
T=${1:-dense}
for rep in 1 2 3; do
  python3 bench.py $Q > gpurun_out/${T}_d_$rep.json 2> gpurun_out/${T}_d_$rep.err || echo fail dense
done
python3 bench.py $Q --scaling strong --particles-total 1048576 --landmarks 1000 --steps 30 > gpurun_out/${T}_ns.json 2> gpurun_out/${T}_ns.err || echo fail ns
python3 bench.py $Q --particles 524288 --landmarks 5000 --steps 20 > gpurun_out/${T}_c5.json 2> gpurun_out/${T}_c5.err || echo fail c5
python3 bench.py $Q --observed 128 > gpurun_out/${T}_o128.json 2> gpurun_out/${T}_o128.err || echo fail o128
echo done
