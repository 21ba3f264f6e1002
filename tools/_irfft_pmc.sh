mkdir -p gpurun_out/r06 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r06/irfft_$c -- python3 tools/sweep.py --kinds irfft32 --only-n 65536 --out gpurun_out/r06/sw_h.json > gpurun_out/r06/irfft_$c.log 2>&1
f=$(find gpurun_out/r06/irfft_$c -name "*counter_collection.csv" | head -1)
python3 - "$f" $c <<'P'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'kofft' in r['Kernel_Name']: d[r['Kernel_Name'][:100]].append(float(r['Counter_Value']))
for k,v in d.items(): print(sys.argv[2], k, len(v), sum(v)/len(v)*1024*(2 if sys.argv[2]=='FETCH_SIZE' else 1)/1e6, 'MB per launch')
P
done
