#!/bin/bash
# Counter passes over tools/one_view.py (single stream, GS_VIEWS views of the bench scene), one rocprofv3 --pmc
# run per pass (at most 8 SQ_, 4 TCC_ (FETCH_SIZE takes 3, WRITE_SIZE 2), 2 GRBM_ counters each), then the
# kernel trace with --stats.  Run on the GPU box from the repo root:  bash tools/pmc_collect.sh gpurun_out/pmc_r03   (GS_LISTS=exact|culled, GS_N / GS_W / GS_H / GS_LS0 / GS_LS1 select the workload)
# Fold with: python tools/pmc_summary.py profiles/r03_pmc_summary.json <out>/p1 <out>/p2 <out>/p3 <out>/p4
set -e
OUT=${1:-gpurun_out/pmc}
export GS_VIEWS=${GS_VIEWS:-8}
mkdir -p "$OUT"
cd /tmp 2>/dev/null && cd - >/dev/null
export TMPDIR=/tmp
pass() {
    n=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT/p$n" --output-format csv -- python3 tools/one_view.py > "$OUT/p$n.log" 2>&1
    echo "pass $n done"
}
pass 1 FETCH_SIZE TCC_EA0_ATOMIC_sum SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass 2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE
pass 3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 tools/one_view.py > "$OUT/trace.log" 2>&1
echo "trace done"
