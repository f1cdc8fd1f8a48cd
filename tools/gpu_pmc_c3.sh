#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in c3 c2; do
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_a_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_a_$w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmc_b_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_b_$w.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
for w in ("c3","c2"):
  for tag in ("a","b"):
    for path in glob.glob("$OUT/pmc_%s_%s/**/*counter_collection.csv"%(tag,w), recursive=True):
        d=collections.defaultdict(list)
        for row in csv.DictReader(open(path)):
            if 'permute_kernel' in row['Kernel_Name']:
                d[row['Counter_Name']].append(float(row['Counter_Value']))
        print(w, tag, {k: sorted(v)[len(v)//2] for k,v in d.items()})
PY
