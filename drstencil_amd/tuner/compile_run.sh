#!/bin/bash
# compile_run.sh <config-name> [<kernel-regex>]
# Counterpart of the reference's benchmarks/<stencil>/compile_run.sh:1-5 (nvcc -arch=sm_80 +
# `ncu --kernel-id ::dr_<name>:10 --csv --set full`): compile cu/<name>.hip for gfx950 and profile the
# emitted program with rocprofv3.  One run per counter group, because gfx950 cannot count FETCH_SIZE and WRITE_SIZE
# in one pass and counters must not be mixed with other trace domains:
#   prof/<name>/trace  kernel durations   (--kernel-trace --stats)
#   prof/<name>/fetch  FETCH_SIZE         (--pmc, own run)
#   prof/<name>/write  WRITE_SIZE         (--pmc, own run)
#   prof/<name>/tcc    L2 hits / misses; prof/<name>/sq, sq2: wave cycles, waiting, LDS bank conflicts, instruction mix (VALU, VMEM, LDS,
#   SALU, branches); prof/<name>/grbm: busy cycles -> the clock the chip held under this kernel  (round 4: the "why" beside each
#   winner; the reference's `ncu --set full` row has these families too, getGpuMetrics.py:9).  DRS_PROFILE_PASSES selects the groups
#   (default: all of them; "fetch write" = the round-3 set).
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
declare -A PMC=(
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE"
  [tcc]="TCC_HIT_sum TCC_MISS_sum"
  [sq]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  [sq2]="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  [grbm]="GRBM_GUI_ACTIVE GRBM_COUNT"
)
for pass in ${DRS_PROFILE_PASSES:-fetch write tcc sq sq2 grbm}; do
  rm -rf prof/${name}/${pass}
  # a group this rocprofv3 / device cannot collect leaves no directory: getGpuMetrics.py writes empty cells for it
  rocprofv3 --kernel-trace --pmc ${PMC[$pass]} --kernel-include-regex "${regex}" --output-format csv -d prof/${name}/${pass} -- bin/${name} > prof/${name}/${pass}.log 2>&1 || { echo "compile_run.sh: counter pass ${pass} failed for ${name}" >&2; rm -rf prof/${name}/${pass}; }
done
