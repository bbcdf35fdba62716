#!/bin/bash
# Hardware-counter passes over a few frames of the bench workload (run on the GPU box through gpurun).
#   tools/pmc.sh <outdir> [first-group [last-group]]
# One rocprofv3 run per counter group (the SQ block has 8 slots, TCC 4), counters only -- never combined with the
# trace domains (MI355X_MICROARCH.md "rocprofv3 PMC slots").  tools/pmc_report.py sums the CSVs per kernel.
set -e
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
export TMPDIR=/tmp
groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
  "SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS_F32"
  "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TOTAL_ACCESSES TCP_PENDING_STALL_CYCLES"
  "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY"
  "TCP_GATE_EN1 TCP_GATE_EN2 TCP_TCR_TCP_STALL_CYCLES TCP_RFIFO_STALL_CYCLES"   # (TA_* counters abort rocprofv3 7.2 on this pool)
  "TCC_HIT TCC_MISS TCC_REQ TCC_EA_RDREQ"
  "TCC_EA_RDREQ_32B TCC_EA_WRREQ TCC_EA_WRREQ_64B TCC_READ"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
  "FETCH_SIZE"      # KB; on gfx950 x2 for wide coalesced reads (MI355X_MICROARCH.md "HBM")
  "WRITE_SIZE"      # KB
)
first=${2:-1}; last=${3:-${#groups[@]}}
i=0
for g in "${groups[@]}"; do
  i=$((i + 1))
  if [ $i -lt $first ] || [ $i -gt $last ]; then continue; fi
  rocprofv3 --pmc $g -d "$out/g$i" -o run --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > "$out/g$i.log" 2>&1 || echo "group $i failed (see $out/g$i.log)"
  echo "group $i done"
done
