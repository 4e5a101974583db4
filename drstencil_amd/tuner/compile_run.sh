#!/bin/bash
# compile_run.sh <config-name> [<kernel-regex>]
# Counterpart of the reference's benchmarks/<stencil>/compile_run.sh:1-5 (nvcc -arch=sm_80 +
# `ncu --kernel-id ::dr_<name>:10 --csv --set full`): compile cu/<name>.hip for gfx950 and profile the
# emitted program with rocprofv3.  Three runs, because gfx950 cannot count FETCH_SIZE and WRITE_SIZE
# in one pass and counters must not be mixed with other trace domains:
#   prof/<name>/trace  kernel durations   (--kernel-trace --stats)
#   prof/<name>/fetch  FETCH_SIZE         (--pmc, own run)
#   prof/<name>/write  WRITE_SIZE         (--pmc, own run)
# The program's own stdout (timing + [Test] lines) lands in prof/<name>.log.
set -e
ARCH=gfx950
name=$1
regex=${2:-dr_}
here=$(cd "$(dirname "$0")" && pwd)
support=${DRS_SUPPORT:-$here/../csrc/support}
mkdir -p bin prof
src=cu/${name}.hip
[ -f "$src" ] || src=cu/${name}.cu      # the reference's tuner writes cu/<name>.cu; hipcc compiles it as HIP
# the compiler's per-kernel resource report (registers, scratch, spills) goes to prof/<name>.resources.txt
CFLAGS="-O3 --offload-arch=${ARCH} -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -I${support} -I cu"
# --cc-opt flags are part of a kernel's identity (the bench headline uses -fno-slp-vectorize): the generator prints them in the
# banner's "// build:" line between -ffp-contract=off and -Rpass-analysis, and the profiled binary is built with them like the
# runtime's plugin (capi.cpp)
ccopts=$(sed -n 's/^\/\/ build: .*-ffp-contract=off\(.*\) -Rpass-analysis=kernel-resource-usage.*/\1/p' "${src}" | head -1)
CFLAGS="${CFLAGS}${ccopts}"
hipcc -x hip ${src} ${CFLAGS} -o bin/${name} 2> prof/${name}.resources.txt || { cat prof/${name}.resources.txt >&2; exit 1; }
# same build policy as the runtime (capi.cpp): a dr_ kernel that needs AGPRs is rebuilt without the scheduler's high-register-
# pressure reschedule stage (the compiler miscompiled two such kernels in the round-1 fuzz)
if awk '/Function Name: dr_/{f=1} /Function Name: gold_/{f=0} f && /AGPRs: [1-9]/{found=1} END{exit !found}' prof/${name}.resources.txt; then
  hipcc -x hip ${src} ${CFLAGS} -mllvm -amdgpu-disable-unclustered-high-rp-reschedule -o bin/${name} 2> prof/${name}.resources.txt || { cat prof/${name}.resources.txt >&2; exit 1; }
fi
# ... and the runtime's refusals: a dr_ kernel that spills to scratch or spills scalar registers is not measured
# (DRS_ALLOW_SCRATCH=1 overrides, as in capi.cpp)
if [ "${DRS_ALLOW_SCRATCH:-0}" != "1" ] && awk '/Function Name: dr/{f=1} /Function Name: gold_/{f=0} f && (/ScratchSize \[bytes\/lane\]: [1-9]/ || /SGPRs Spill: [1-9]/){found=1} END{exit !found}' prof/${name}.resources.txt; then
  echo "compile_run.sh: dr_ kernel of ${name} spills (scratch or scalar registers): refused like the runtime does" >&2
  exit 3
fi
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d prof/${name}/trace -- bin/${name} > prof/${name}.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "${regex}" --output-format csv -d prof/${name}/fetch -- bin/${name} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "${regex}" --output-format csv -d prof/${name}/write -- bin/${name} > /dev/null 2>&1
