#!/bin/bash
# Counter passes for ONE workload of tools/probe.py: separate rocprofv3 --pmc runs (the blocks have few
# slots per pass; FETCH_SIZE and WRITE_SIZE cannot share one), kernel-trace only -- never combined with
# --sys-trace / hip / hsa tracing.  Usage: [ALGO=4] [PASSES="sq1 sq4"] [TUNE="--tune median_algo=1"] tools/pmc_passes.sh <workload> <outdir-tag> [launches]
set -e
WL=$1; TAG=$2; N=${3:-6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${TAG}
mkdir -p $OUT
run() {  # name, counters...
  local name=$1; shift
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $name "; then return; fi   # PASSES="sq1 sq2": a subset
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o run -- python3 tools/probe.py --workload $WL --launches $N $TUNE ${ALGO:+--algo $ALGO} > $OUT/$name.log 2>&1
  echo "pass $name done"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
run sq3 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq4 GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS
run tcp1 TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_GATE_EN1
run tcp2 TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_WRITE TCP_TOTAL_READ
run tcp3 TCP_TCC_ATOMIC_WITH_RET_REQ TCP_TCC_ATOMIC_WITHOUT_RET_REQ TCP_TCP_TA_DATA_STALL_CYCLES TCP_TOTAL_ACCESSES
run tcc1 TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_REQ
run tcc2 TCC_WRITE TCC_READ TCC_ATOMIC TCC_CYCLE
run tcc3 TCC_HIT TCC_MISS TCC_TAG_STALL TCC_EA0_RDREQ
run tcc4 TCC_WRITEBACK TCC_NORMAL_WRITEBACK TCC_TOO_MANY_EA_WRREQS_STALL TCC_BUSY
run fetch FETCH_SIZE
run write WRITE_SIZE
