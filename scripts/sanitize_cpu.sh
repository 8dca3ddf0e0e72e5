#!/bin/bash
# (host) The C code that runs on the CPU -- the oracle (oracle/lbdrn_oracle.c, oracle/plane_codec.c) and the OpenJPEG shim
# (csrc/jp2_shim.c) -- rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer (first finding aborts) and put under the CPU tests
# that drive them: oracle vs the reference's fixtures, the host logic, the JPEG 2000 payload incl. damaged streams.  The regular
# builds are restored afterwards.  GPU sanitizers are not available on this pool; this is the CPU half.
#     bash scripts/sanitize_cpu.sh
set -u
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
SAN="-O1 -g -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined"
cp oracle/_build/liblbdrn_oracle.so /tmp/lbdrn_oracle_regular.so
cp lbdrn-msic_amd/liblbdrn_jp2.so /tmp/lbdrn_jp2_regular.so 2>/dev/null
restore() {
  cp /tmp/lbdrn_oracle_regular.so oracle/_build/liblbdrn_oracle.so; touch oracle/_build/liblbdrn_oracle.so
  [ -f /tmp/lbdrn_jp2_regular.so ] && { cp /tmp/lbdrn_jp2_regular.so lbdrn-msic_amd/liblbdrn_jp2.so; touch lbdrn-msic_amd/liblbdrn_jp2.so; }
}
trap restore EXIT
gcc $SAN -std=gnu11 -ffp-contract=off -fno-fast-math -Wall -o oracle/_build/liblbdrn_oracle.so oracle/lbdrn_oracle.c oracle/plane_codec.c -lm || exit 1
INC=$(ls -d /opt/conda/include/openjpeg-* 2>/dev/null | tail -1); LIB=$(ls /opt/conda/lib/libopenjp2.so* 2>/dev/null | head -1)
if [ -n "$INC" ] && [ -n "$LIB" ]; then
  gcc $SAN -Wall -I"$INC" -Iinclude -o lbdrn-msic_amd/liblbdrn_jp2.so lbdrn-msic_amd/csrc/jp2_shim.c "$LIB" -Wl,-rpath,/opt/conda/lib || exit 1
fi
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_golden.py tests/test_host_logic.py tests/test_jp2_payload.py -x -q -m "not gpu"
