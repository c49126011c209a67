# Round 4: the VALU issue-interval micro-benchmark (tools/ubench/valu_rate.hip) and the clock it ran at.
# 1. plain run: in-wave s_memtime / s_memrealtime stamps and HIP events
# 2. the same binary under rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace: busy cycles / duration = clock per launch
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04valu
mkdir -p $O
cd tools/ubench
[ -x valu_rate ] || hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate 2>/dev/null
./valu_rate > $O/valu_rate.txt 2>&1
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/gui -o v -- $R/tools/ubench/valu_rate > $O/valu_rate_under_pmc.txt 2> $O/gui.err
cd $R
python tools/valu_clock.py $O/gui > $O/valu_clock.txt 2>&1
rm -f $O/gui/*/v_agent_info.csv
du -sh $O
