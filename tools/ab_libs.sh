#!/bin/bash
# same-box comparison of several builds of the library on the bench's timed loop:  bash tools/ab_libs.sh rounds lib1.so lib2.so ...
N=$1; shift
cd "$(dirname "$0")/.."
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for ((i = 0; i < N; i++)); do
  line="tree $(run X=1)"
  for lib in "$@"; do line="$line   $(basename $lib) $(run LOFT_HIP_LIB=$lib)"; done
  echo "$line"
done
