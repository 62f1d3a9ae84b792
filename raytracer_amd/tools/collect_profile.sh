#!/bin/bash
# Collects the rocprofv3 passes that profiles/summarize.py condenses: kernel trace + stats, the SQ passes the
# vector-ALU model of bench.py rests on (sq, sq3: the instruction counts per class), clock (GRBM), fabric traffic (FETCH_SIZE, WRITE_SIZE, one pass each), cache
# counters, and the FETCH_SIZE calibration (tools/fetch_calib: random 64-byte and 128-byte records and a stream
# over a 25 MiB and a 1 GiB table).  Every pass is a run of its own.  Usage, on the GPU box from the repo root:
#   bash raytracer_amd/tools/collect_profile.sh gpurun_out/prof_xyz
set -u
D=${1:-gpurun_out/prof}
export TMPDIR=/tmp
mkdir -p "$D"
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-frame-calls"
pass() {  # name, then the rocprofv3 options
	local name=$1; shift
	timeout -s KILL 200 rocprofv3 "$@" --output-format csv -d "$D/$name" -- $B > "$D/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$D/$name.log"; exit 1; }
}
pass kt --kernel-trace --stats
pass sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP32
pass sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA
pass grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch --pmc FETCH_SIZE
pass write --pmc WRITE_SIZE
pass cache --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
C=./raytracer_amd/tools/fetch_calib
if [ -x $C ]; then
	for M in 25 1024; do
		timeout -s KILL 60 $C $M 8 > "$D/calib_$M.log" 2>&1 || { echo "calibration run failed"; exit 1; }
		timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/calib_fetch_$M" -- $C $M 8 > "$D/calib_fetch_$M.log" 2>&1 || { echo "calibration FETCH_SIZE pass failed"; exit 1; }
		timeout -s KILL 60 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$D/calib_req_$M" -- $C $M 8 > "$D/calib_req_$M.log" 2>&1 || { echo "calibration request pass failed"; exit 1; }
	done
fi
# issue rates of the instructions the traversal kernel uses, and which of the SQ_INSTS_VALU_* counters each of them lands in
R=./raytracer_amd/tools/valu_rate
if [ -x $R ]; then
	timeout -s KILL 120 $R > "$D/rates.log" 2>&1 || { echo "valu_rate failed"; exit 1; }
	timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP32 --output-format csv -d "$D/rates_classes" -- $R > "$D/rates_classes.log" 2>&1 || { echo "valu_rate counter pass failed"; exit 1; }
fi
echo "profile passes written to $D"
