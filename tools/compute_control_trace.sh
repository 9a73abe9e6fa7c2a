# kernel trace of mppi_compute_control (Cartpole K=16384, T=100) calls from an idle stream: where the call's ~36-38 us go on the
# device — every kernel of a call (rollout, then whatever the hand-over launches: merge + control phase or the merging control
# phase, trajectory phase, input copy), start relative to the rollout's, duration, gap to the previous kernel's end; what is left
# is the host side (launch-to-start latency of the first kernel + the flag's way back).  MPPI_AMD_NO_MERGE_CONTROL=1 for the
# two-launch form.  Writes gpurun_out/compute_control_trace.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cc_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cc_trace -- python tools/compute_control_latency.py idle > gpurun_out/cc_trace.log 2>&1
f=$(find gpurun_out/cc_trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, json, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    return re.sub(r'^void\s+', '', n).split('<')[0].split('(')[0].replace('mppi::kernels::', '')
seq = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# a call = a rollout launch and every kernel up to the next rollout launch (whatever form the hand-over has: with or without
# ingest launch, merge as a launch of its own or inside the control phase, trajectory phase on the side stream)
calls, cur = [], None
for k in seq:
    if k[0].startswith('rolloutPipelineKernel'):
        if cur:
            calls.append(cur)
        cur = [k]
    elif cur:
        cur.append(k)
calls = calls[len(calls) // 2:]  # the idle-stream loop is the last one the tool runs
shape = collections.Counter(tuple(k[0] for k in c) for c in calls).most_common(1)[0][0]
calls = [c for c in calls if tuple(k[0] for k in c) == shape]
def med(v):
    v = sorted(v)
    return round(v[len(v) // 2] / 1e3, 2)
out = collections.OrderedDict()
out['calls'] = len(calls)
out['kernels_of_a_call'] = list(shape)
for i, name in enumerate(shape):
    out['%d_%s' % (i, name)] = {'start_after_rollout_start_us': med([c[i][1] - c[0][1] for c in calls]),
                                'duration_us': med([c[i][2] - c[i][1] for c in calls]),
                                'gap_after_previous_end_us': med([c[i][1] - c[i - 1][2] for c in calls]) if i else None}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/compute_control_trace.json', 'w'), indent=1)
PY
tail -3 gpurun_out/cc_trace.log
