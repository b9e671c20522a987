# what a 20-step timed region costs against 200 steps: brackets or the region's ends?
Q="--no-cpu-baseline --no-extra-legs --no-sweep"
for rep in 1 2 3; do
  for v in "s20:--steps 20 --warmup 5" "s20_noev:--steps 20 --warmup 5 --events none" "s20_e10:--steps 20 --warmup 5 --event-every 10" "s20_w20:--steps 20 --warmup 20" "s200:--steps 200 --warmup 20" "s200_noev:--steps 200 --warmup 20 --events none"; do
    name=${v%%:*}; flags=${v#*:}
    python3 bench.py $Q $flags > gpurun_out/short_${name}_$rep.json 2>/dev/null || echo fail $name
  done
done
echo done
