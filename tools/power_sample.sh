#!/bin/bash
# Power / clock of the GPU while ONE kernel class runs back to back (the Gram matvec at the C4 shape, then the gradient GEMM):
# evidence for "power-limited" in DESIGN.md section 3.2.  rocm-smi is sampled every 0.5 s next to a ~12 s loop of launches.
# usage (on the GPU box): tools/power_sample.sh [outdir]
out=${1:-gpurun_out/power}
mkdir -p "$out"
rocm-smi --showpower --showclocks --showmaxpower > "$out/idle.txt" 2>&1
for what in matvec grad; do
  python tools/power_loop.py $what 12 > "$out/loop_$what.log" 2>&1 &
  pid=$!
  sleep 4   # import + warm-up
  : > "$out/samples_$what.txt"
  for i in $(seq 1 14); do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" >> "$out/samples_$what.txt"
    echo "--" >> "$out/samples_$what.txt"
    sleep 0.5
  done
  wait $pid
done
grep -h "Power" "$out"/samples_matvec.txt | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END {print "matvec: power samples (W) min/median/max", a[1], a[int((NR+1)/2)], a[NR]}'
grep -h "Power" "$out"/samples_grad.txt | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END {print "grad:   power samples (W) min/median/max", a[1], a[int((NR+1)/2)], a[NR]}'
grep -h -i "max" "$out/idle.txt" | head -3
tail -2 "$out"/loop_matvec.log "$out"/loop_grad.log
