mkdir -p gpurun_out/probe
{
echo "== hwmon"; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $d; ls $d; for f in $d/power1_average $d/power1_input $d/freq1_input $d/freq2_input $d/power1_cap $d/temp1_input $d/name; do [ -e $f ] && echo "$f = $(cat $f 2>&1)"; done; done
echo "== device files"; for c in /sys/class/drm/card*/device; do echo $c; ls $c | tr '\n' ' '; echo; for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent mem_busy_percent current_link_speed; do [ -e $c/$f ] && { echo "-- $f"; cat $c/$f; }; done; done
echo "== amdsmi"; python -c "import amdsmi; print(amdsmi.__file__)" 2>&1 | tail -1
ls /opt/rocm/share/amd_smi 2>&1 | head; ls /opt/rocm/lib | grep -i smi
echo "== rocm-smi timing"; time rocm-smi --showpower --showclocks --json
echo "== gpu_metrics size"; for c in /sys/class/drm/card*/device; do ls -la $c/gpu_metrics 2>&1; done
nproc; free -g | head -2
hipcc --version | head -3
} > gpurun_out/probe/probe.txt 2>&1
tail -100 gpurun_out/probe/probe.txt
