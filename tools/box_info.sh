#!/bin/bash
# What the GPU box exposes about clocks / power / partitions (sysfs, rocm-smi, amd-smi) -- read-only.
for d in /sys/class/drm/card*/device; do
  [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ] || continue
  echo "== $d -> $(readlink -f $d)"
  ls $d | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk current_compute_partition current_memory_partition \
           power_dpm_force_performance_level gpu_busy_percent mem_busy_percent available_compute_partition; do
    [ -r $d/$f ] && { echo "-- $f"; cat $d/$f; }
  done
  for h in $d/hwmon/hwmon*; do
    echo "-- $h: $(ls $h | tr '\n' ' ')"
    for f in power1_average power1_input power1_cap temp1_input temp2_input temp3_input freq1_input freq2_input; do
      [ -r $h/$f ] && echo "$f $(cat $h/$f)"
    done
  done
done
echo "== rocm-smi"; timeout 60 rocm-smi --showclocks --showpower --showtemp --showperflevel --showcomputepartition --showmemorypartition 2>&1 | head -80
echo "== amd-smi"; timeout 60 amd-smi metric --clock --power --temperature 2>&1 | head -150
echo "== rocminfo (agents)"; timeout 60 rocminfo 2>&1 | grep -E "Marketing Name|Compute Unit|Max Clock|Name:.*gfx" | head -20
