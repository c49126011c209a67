# Round 5 soak of the signer after the ADVICE r4 changes (chunks of <= 2^20 items, scrubbed scratch) and of the masked
# walk: 2 621 440 items = two full chunks + a partial one; every setting must print the same sha256(r, s).
cd "$(dirname "$0")/.."
N=2621440
for e in "A=default" "STARKPERP_SIGN_COMPACT_MIN=0" "STARKPERP_SIGN_CHUNK=65536" "STARKPERP_SIGN_MASKED=1" "STARKPERP_SIGN_MASKED=1 STARKPERP_SIGN_COMPACT_MIN=0"; do
  echo "== $e"; env $e python tools/soak_sign.py $N 2>&1 | grep -v amdgpu.ids
done
