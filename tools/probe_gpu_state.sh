#!/bin/bash
# One-off probe (round 3): which clock / power readouts exist on the GPU box, and what they cost.
for c in /sys/class/drm/card*/device; do
  echo "== $c"; ls $c | tr '\n' ' ' | head -c 1500; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk current_link_speed gpu_busy_percent; do echo "-- $f"; cat $c/$f 2>&1 | head -12; done
  for h in $c/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input freq2_input temp1_input; do echo -n "$f: "; cat $h/$f 2>&1; done; done
done
echo "== rocm-smi json"; time rocm-smi --showclocks --showpower --showtemp --json 2>&1 | head -c 3000; echo
echo "== amd-smi metric"; time amd-smi metric --clock --power --json 2>&1 | head -c 3000; echo
python - <<'PY'
import time
t0=time.time()
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("amdsmi handles", len(hs), "init s", time.time()-t0)
    t0=time.time()
    for fn in ("amdsmi_get_gpu_metrics_info","amdsmi_get_power_info"):
        try:
            r = getattr(amdsmi, fn)(hs[0]); print(fn, {k: r[k] for k in list(r)[:40]} if isinstance(r, dict) else r)
        except Exception as e: print(fn, "ERR", e)
    try:
        print(amdsmi.amdsmi_get_clock_info(hs[0], amdsmi.AmdSmiClkType.GFX)); print(amdsmi.amdsmi_get_clock_info(hs[0], amdsmi.AmdSmiClkType.MEM))
    except Exception as e: print("clk ERR", e)
    print("query s", time.time()-t0)
except Exception as e:
    print("amdsmi ERR", repr(e))
PY
