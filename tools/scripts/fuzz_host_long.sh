#!/bin/bash
# The long budget of tests/test_sanitizers.py: >= 10^5 mutation trials of the host parser under ASan + UBSan, 8 processes with different seeds.
# usage: tools/scripts/fuzz_host_long.sh [trials per process = 15000] [seconds per process = 1200]
set -e
cd "$(dirname "$0")/../.."
make -s -C jpegxl-rs_amd -j8 asan
CORPUS=$(mktemp -d)
python - "$CORPUS" <<'PY'
import sys, os
sys.path.insert(0, "tests")
import test_sanitizers as T
print("corpus files:", T.make_corpus(sys.argv[1]))
PY
export ASAN_OPTIONS=abort_on_error=1:detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
pids=()
for seed in 11 12 13 14 15 16 17 18; do
  jpegxl-rs_amd/build_asan/fuzz_host "$CORPUS" "${1:-15000}" "${2:-1200}" $seed > "$CORPUS/out_$seed.json" 2> "$CORPUS/err_$seed.txt" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
cat "$CORPUS"/out_*.json
if [ $rc -ne 0 ]; then tail -50 "$CORPUS"/err_*.txt; echo "SANITIZER REPORT (see above)"; exit 1; fi
echo "no sanitizer report"
