#!/bin/bash
# try_gpus_host.sh -- `drstencil --gpus N`: generate, compile and run the emitted N-GPU host on a one-GPU box: (a) rank 1 of 4 rehearsed alone (self-neighbour
# exchange through RCCL), (b) the forking form, which must notice that ranks 1..3 have no GPU and leave at once with exit code 1, (c) a 2D y-slab rehearsal
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-gpurun_out/gpus_host}
mkdir -p $W
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$R/include -I$R/drstencil_amd/csrc/support -L$R/drstencil_amd -ldrstencil_amd -Wl,-rpath,$R/drstencil_amd"
$R/bin/drstencil --3d --dtype fp32 --step 2 --sn 16 --gpus 4 --check -o $W/g4.hip $R/tests/stc/t3_star.stc || exit 1
$CC -o $W/g4 $W/g4.hip || exit 1
$R/bin/drstencil --dtype fp64 --gpus 3 --check -o $W/g3_2d.hip $R/tests/stc/t2_star.stc || exit 1
$CC -o $W/g3_2d $W/g3_2d.hip || exit 1
echo "== (a) DRS_SLAB_REHEARSE=1/4 g4"; DRS_SLAB_REHEARSE=1/4 timeout -k 5 120 $W/g4; echo "rc=$?"
echo "== (b) g4 on one GPU (must fail cleanly)"; timeout -k 5 120 $W/g4; echo "rc=$?"
echo "== (b2) DRS_SLAB_WORLD=1 g4: a one-rank world through the same code, --check against the gold kernel"; DRS_SLAB_WORLD=1 timeout -k 5 120 $W/g4; echo "rc=$?"
echo "== (c2) DRS_SLAB_WORLD=1 g3_2d"; DRS_SLAB_WORLD=1 timeout -k 5 120 $W/g3_2d; echo "rc=$?"
echo "== (c) DRS_SLAB_REHEARSE=1/3 g3_2d"; DRS_SLAB_REHEARSE=1/3 timeout -k 5 120 $W/g3_2d; echo "rc=$?"
# round 4: the failure paths.  (d) a rank that dies makes rank 0 end the others at once (SIGCHLD), exit code 1; (e) a rank that hangs is ended by the
# watchdog (DRS_SLAB_TIMEOUT seconds), exit code 124; (f) no spec files are left behind in TMPDIR by any of the runs above
echo "== (d) g4 with rank 2 dying early"; t0=$(date +%s); DRS_SLAB_TEST_FAIL_RANK=2 timeout -k 5 60 $W/g4; echo "rc=$? after $(( $(date +%s) - t0 )) s"
echo "== (e) g4 with rank 1 hanging, DRS_SLAB_TIMEOUT=3"; t0=$(date +%s); DRS_SLAB_TEST_HANG_RANK=1 DRS_SLAB_TIMEOUT=3 timeout -k 5 60 $W/g4; echo "rc=$? after $(( $(date +%s) - t0 )) s"
echo "== (f) spec files left in ${TMPDIR:-/tmp}: $(ls -d ${TMPDIR:-/tmp}/drs_t3_star_* ${TMPDIR:-/tmp}/drs_t2_star_* 2>/dev/null | wc -l)"
echo "== (g) second run of the rehearsal: the kernel cache must hit (stable kernel names)"; before=$(ls $R/drstencil_amd/_kcache | wc -l); DRS_SLAB_REHEARSE=1/4 DRS_NO_COMPILE=1 timeout -k 5 120 $W/g4 > /dev/null; echo "rc=$? cache entries $before -> $(ls $R/drstencil_amd/_kcache | wc -l)"
