#!/bin/bash
# the verification kernels at one wave per SIMD (the allocator's choice: 256 VGPRs + 2 / 11 / 24
# AGPRs) against two (amdgpu_waves_per_eu(2, 2): 256 registers, small spills) - and the same switch on the signers,
# the Pedersen composition and the bulk hash kernel.  Variant libraries are built in the container and travel with
# the snapshot:
#   first pass (product = round 5's registers):  make VARIANT=v2 EXTRA="-DSP_VERIFY_WAVES=2 -DSP_AIR_ECDSA_WAVES=2"
#   second pass (product = 2 waves, adopted):    make VARIANT=w1 EXTRA="-DSP_VERIFY_WAVES=0 -DSP_AIR_ECDSA_WAVES=0"
#                                                make VARIANT=w3 EXTRA="-DSP_VERIFY_WAVES=3 -DSP_SIGN_WAVES=3 -DSP_AIR_WAVES=3 -DSP_ACC_WAVES=4"  Output: gpurun_out/r06_verify_occupancy.txt
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_verify_occupancy.txt
mkdir -p gpurun_out
: > $O
echo "# verification kernels, occupancy A/B ($(date -u +%Y-%m-%dT%H:%M:%SZ)); best of 3 event-timed loops per line" >> $O
for v in "" ${VARIANTS:-w1 w3}; do
  if [ -z "$v" ]; then lib=stark-perpetual_amd/lib/libstarkperp.so; name="product"; else lib=stark-perpetual_amd/csrc/build_$v/libstarkperp_$v.so; name="$v"; fi
  [ -f "$lib" ] || { echo "missing $lib" >> $O; continue; }
  STARKPERP_LIB=$PWD/$lib python tools/verify_occupancy_ab.py "$name" >> $O 2>> gpurun_out/r06_verify_occupancy.err
done
# the same A/B interleaved once more (order effects: clocks, power state)
for v in ${VARIANTS:-w1 w3} ""; do
  if [ -z "$v" ]; then lib=stark-perpetual_amd/lib/libstarkperp.so; name="product, second pass"; else lib=stark-perpetual_amd/csrc/build_$v/libstarkperp_$v.so; name="$v, second pass"; fi
  [ -f "$lib" ] || continue
  STARKPERP_LIB=$PWD/$lib python tools/verify_occupancy_ab.py "$name" >> $O 2>> gpurun_out/r06_verify_occupancy.err
done
cat $O
