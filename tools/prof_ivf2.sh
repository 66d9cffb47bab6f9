#!/bin/bash
# development aid: per-kernel times of the IVF chain on bench-like data (matrix-core path)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ivf2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ivf2 -o m -- python3 $R/tools/ivf_bench2.py "$@" > $R/gpurun_out/ivf2_mfma.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_ivf2/m_kernel_stats.csv")))
skip=("kpp_","kmeans_","seq_sqsum","scatter_rows","pool_row_norms","max_f32","copyBuffer")
for r in rows:
    if any(s in r['Name'] for s in skip): continue
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9),'us  min', ('%.1f'%(float(r['MinNs'])/1e3)).rjust(8))
PY
