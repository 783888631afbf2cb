#!/bin/bash
# round 5, VERDICT item 6: what differs between boxes.  One call per box: the 10-step headline line (two sub-batch streams), the same batch on ONE stream (every kernel runs
# alone: its per-kernel times are exact, the two-stream ones are smeared by the neighbour stream), a device-to-device copy rate (HBM clock) and the multiply-add peak probe
# (inside both bench lines).  TAG=name of the output files.
set -o pipefail
O=$PWD/gpurun_out/r05v; mkdir -p $O; T=${TAG:-box}
python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline --no-rlc > $O/$T.json 2> $O/$T.err || { tail -5 $O/$T.err; exit 1; }
BN254_STREAMS=1 python bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-rlc > $O/${T}_one_stream.json 2> $O/${T}_one_stream.err || { tail -5 $O/${T}_one_stream.err; exit 1; }
python - > $O/$T.copy_rate <<'PY'
import torch, json
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(json.dumps({"d2d_copy_1GiB_ms": ms, "GB_per_s_read_plus_write": 2 * (1 << 30) / ms / 1e6}))
PY
cat $O/$T.copy_rate
# clocks and power WHILE the batch runs (after the measurements above, so that sampling does not disturb them): sysfs every 50 ms during a 20-step run
CARD=$(ls -d /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -1)
if [ -n "$CARD" ]; then
  D=$(dirname $CARD)
  ( while true; do
      s=$(grep '\*' $D/pp_dpm_sclk 2>/dev/null | tr -d '\n'); m=$(grep '\*' $D/pp_dpm_mclk 2>/dev/null | tr -d '\n'); f=$(grep '\*' $D/pp_dpm_fclk 2>/dev/null | tr -d '\n')
      p=$(cat $D/hwmon/hwmon*/power1_average 2>/dev/null | head -1); t=$(cat $D/hwmon/hwmon*/temp1_input 2>/dev/null | head -1)
      echo "$(date +%s.%N) sclk[$s] mclk[$m] fclk[$f] power_uW[$p] temp_mC[$t]"
      sleep 0.05
    done ) > $O/$T.clock_samples 2>/dev/null &
  SAMPLER=$!
  python bench.py --steps 20 --warmup 3 --no-configs --no-cpu-baseline --no-rlc > $O/${T}_sampled.json 2> $O/${T}_sampled.err
  kill $SAMPLER 2>/dev/null
  wait $SAMPLER 2>/dev/null
  wc -l $O/$T.clock_samples
  sort -k2 $O/$T.clock_samples | awk '{print $2, $3, $4}' | uniq -c | sort -rn | head -8
fi
