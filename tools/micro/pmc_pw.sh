#!/bin/bash
# SQ counters of one pw_bench configuration.  Usage (GPU box, repo root):
#   bash tools/micro/pmc_pw.sh <tag> <cfg> <layer filter>   ->  gpurun_out/<tag>_pmc_sq_set{1..4}.csv, table on stdout
set -e
TAG=$1; CFG=$2; FILT=$3
export TMPDIR=/tmp
export PW_CFG=$CFG
cd tools/micro
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf ../../gpurun_out/pmc_pw_${TAG}_$i
  rocprofv3 --pmc $SET --output-format csv -d ../../gpurun_out/pmc_pw_${TAG}_$i -- ./pw_bench 256 "$FILT" > /dev/null 2> ../../gpurun_out/pmc_pw_${TAG}_$i.err || echo "set $i failed"
  f=$(find ../../gpurun_out/pmc_pw_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" ../../gpurun_out/${TAG}_pmc_sq_set$i.csv
  rm -rf ../../gpurun_out/pmc_pw_${TAG}_$i
done
cd ../..
python3 tools/pmc_sq_table.py $TAG | grep -E "kernel \||conv_pw|conv_igemm" | cut -c1-260
python3 - <<PY
import csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open("gpurun_out/${TAG}_pmc_sq_set4.csv")):
    k = r["Kernel_Name"][:70]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); n[k] += 1
for k, v in d.items():
    if "conv_pw" in k or "conv_igemm" in k:
        g = v["GRBM_GUI_ACTIVE"]
        print(k[:60], n[k], "launches; MFMA busy %.3f of SIMD-cycles; LDS idx active %.3f of CU-cycles; MFMA insts/launch %.0f" % (
            v["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8 * 256 * 4) if g else 0, v["SQ_LDS_IDX_ACTIVE"] / (g / 8 * 256) if g else 0, v["SQ_INSTS_MFMA"] / n[k]))
PY
