#!/bin/bash
# Prints one md5 per object of pointcloudmatters_amd/csrc: the digest of its .hip_fatbin section, i.e. of the gfx950 code object alone.
# Use: run before and after a change that is meant to touch HOST code only (argument checks, launch tables); equal digests prove that
# the device code -- what the GPU parity tests exercised -- is byte-identical.  (Round 4: the status-contract sweep of
# tests/test_capi.py was made while the GPU pool was closed; profiles/r04_device_digest.txt holds the digests of that build.)
cd "$(dirname "$0")/../pointcloudmatters_amd/csrc" || exit 1
for o in *.o; do
  /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$o" /tmp/pcm_fb.$$ 2>/dev/null
  echo "$(md5sum < /tmp/pcm_fb.$$ | cut -c1-32) $o"
done
rm -f /tmp/pcm_fb.$$
