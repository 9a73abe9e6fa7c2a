cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python tools/time_workloads.py cartpole > gpurun_out/gaps.log 2>&1
f=$(find gpurun_out/gaps -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take a window in the middle of the longest run of alternating pipeline/combine kernels
seq=[(r['Kernel_Name'][:60],int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows]
import collections
out=collections.defaultdict(list)
for i in range(1,len(seq)):
    a,b=seq[i-1],seq[i]
    if 'rolloutPipelineKernel' in a[0] and 'combineKernel' in b[0]:
        out['rollout_dur'].append(a[2]-a[1]); out['gap_rollout_to_combine'].append(b[1]-a[2]); out['combine_dur'].append(b[2]-b[1])
    if 'combineKernel' in a[0] and 'rolloutPipelineKernel' in b[0]:
        out['gap_combine_to_rollout'].append(b[1]-a[2])
for k,v in out.items():
    v=sorted(v); print(k, 'n=%d median=%.2f us p10=%.2f p90=%.2f'%(len(v), v[len(v)//2]/1e3, v[len(v)//10]/1e3, v[9*len(v)//10]/1e3))
PY
