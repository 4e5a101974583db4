#!/bin/bash
# starter.sh <stencil.stc> [tuning.py options] -- one whole tuning session in the current directory, the role of the
# reference's benchmarks/<stencil>/starter.sh:1-11:
#   cu/ bin/ prof/ working directories with the support header beside the emitted sources,
#   the search (tuning.py: generate -> hipcc -> HIP-event duration per configuration, best-so-far in tuning_out/duration.log),
#   the rocprofv3 counters of the best configurations (tuning.py --profile-top: compile_run.sh + getGpuMetrics.py ->
#   gpuMetrics.csv, one row per winner, like the reference's per-configuration ncu scrape),
#   and the session's wall time appended to tuning-time.log.
set -u
tuner_dir=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
if [ $# -lt 1 ]; then
    echo "usage: starter.sh <stencil.stc> [tuning.py options]" >&2
    exit 64
fi
for d in cu bin prof; do mkdir -p "$d"; done
install -m 644 "$tuner_dir/../csrc/support/common.hpp" cu/common.hpp
SECONDS=0
python3 "$tuner_dir/tuning.py" "$@" --out tuning_out --profile-top "${DRS_PROFILE_TOP:-3}"
status=$?
printf '%s\nrunning time: %ds (tuning.py exit %d)\n' "$(date '+%F %T')" "$SECONDS" "$status" >> tuning-time.log
exit $status
