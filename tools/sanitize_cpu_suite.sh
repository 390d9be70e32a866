#!/bin/bash
# Runs the CPU test suite with the oracle, the product's HIP-free host logic (tests/native glue) and the synthetic eNB built with
# AddressSanitizer + UndefinedBehaviorSanitizer (gcc).  Prints every sanitizer report; the normal builds are restored afterwards.
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
T=$(mktemp -d)
F="-O1 -g -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer"
make -s -C oracle >/dev/null 2>&1; make -s -C tools/txgen >/dev/null; make -s -C tests/native >/dev/null
cp oracle/_build/liblsn_oracle.so tools/txgen/_build/libtxgen.so tests/native/_build/liblsn_hosttest.so "$T"/
(cd oracle && gcc -std=gnu11 -fno-fast-math $F -w -o _build/liblsn_oracle.so o_*.c -lm)
(cd tools/txgen && g++ -std=c++17 $F -o _build/libtxgen.so txgen.cc)
H=ltesniffer_amd/csrc/host
g++ -std=c++17 $F -o tests/native/_build/liblsn_hosttest.so tests/native/lsn_hosttest.cc $H/lsn_lte.cc $H/lsn_rrc.cc $H/lsn_search.cc
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -q -m "not gpu" -p no:cacheprovider -s 2>&1 | grep -E "runtime error|AddressSanitizer|passed|failed" | sort | uniq -c || true
cp "$T"/liblsn_oracle.so oracle/_build/; cp "$T"/libtxgen.so tools/txgen/_build/; cp "$T"/liblsn_hosttest.so tests/native/_build/
touch oracle/_build/liblsn_oracle.so tools/txgen/_build/libtxgen.so tests/native/_build/liblsn_hosttest.so
