#!/bin/bash
# Host sanitizers, CPU build only: the oracle and every host-side C file of
# the product (tree utilities, evaluation driver, SPR search) built with
# -fsanitize=address,undefined, then the CPU test-suite on that build.
#   tests/host_sanitizers.sh [pytest args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/oracle/_build_asan
make -s -C $ROOT/oracle OUT=$OUT \
  CFLAGS="-O1 -g -march=x86-64-v3 -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -std=gnu99 -fPIC -Wall -Wextra -Wno-unused-parameter -fopenmp -fvisibility=default"
export PLLHIP_ORACLE_LIB=$OUT/libpll_oracle.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export OMP_NUM_THREADS=2
cd $ROOT
python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
