#!/bin/bash
# The profile set of a round, on the GPU box:  bash tools/profile_round.sh TAG
# rocprofv3 kernel stats and the bench line of the same run for the GLZ and MJPEG workloads,
# FETCH_SIZE / WRITE_SIZE passes of both (PMC counters in runs of their own), and the SQ
# instruction counters.  Everything lands under gpurun_out/prof_TAG/.
set -u
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG
mkdir -p $O
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --also 0 --verify 0"
for w in glz1080 mjpeg2160; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 bench.py --workload $w $ARGS > $O/${w}_line.log 2>&1 || exit 1
  echo "stats $w done"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_$w -- python3 bench.py --workload $w $ARGS > $O/pmc_${c}_$w.log 2>&1 || exit 1
    echo "pmc $c $w done"
  done
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq_$w -- python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --also 0 --verify 0 > $O/pmc_sq_$w.log 2>&1 || exit 1
  echo "pmc sq $w done"
done
# the kernel for an under-filled GPU: sessions64 (32 images per step, a CU each)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sessions64 -- python3 bench.py --workload sessions64 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --also 0 --verify 0 > $O/sessions64_line.log 2>&1 || exit 1
echo "stats sessions64 done"
find $O -name "*.csv" | head -60
