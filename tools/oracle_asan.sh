#!/bin/bash
# The oracle's golden suite (every reference fixture under tests/golden) with the C restatement built under AddressSanitizer and
# UndefinedBehaviorSanitizer -- SURVEY.md section 5's sanitizer hook.  CPU only; writes nothing outside the repo.
#     tools/oracle_asan.sh [pytest args]
case "${1:-}" in -h|--help) sed -n '2,4p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
set -e
cd "$(dirname "$0")/.."
make -C oracle asan >/dev/null
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export ORACLE_LIB_PATH="$PWD/oracle/libunwarp_oracle_asan.so"
exec python -m pytest tests/test_oracle_golden.py -q -p no:cacheprovider "$@"
